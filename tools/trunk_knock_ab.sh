#!/bin/bash
# Timing probes of the 15 x 10 trunk's GEMM core (POL_TRUNK_KNOCK bits, csrc/inv_policy.hip): every variant library in its own process,
# two alternations on one box (tools/trunk_ab.py / tools/bwd_ab.py do the alternating).  Build first (CPU):
#   for k in 1 2 16 32 3 18 50 51; do python tools/build_variant.py tknock$k -DPOL_TRUNK_KNOCK=$k; done
# usage (GPU): bash tools/trunk_knock_ab.sh > gpurun_out/trunk_knock.txt
cd "$(dirname "$0")/.."
L=inversus_reinforcement_learning_amd/libinversus_hip.so
timeout -k 10 900 python tools/trunk_ab.py $L tools/exp/lib_tknock1.so tools/exp/lib_tknock2.so tools/exp/lib_tknock16.so tools/exp/lib_tknock32.so tools/exp/lib_tknock3.so tools/exp/lib_tknock18.so tools/exp/lib_tknock50.so tools/exp/lib_tknock51.so || exit 1
echo "== backward (training forward ms, backward ms)"
timeout -k 10 600 python tools/bwd_ab.py $L tools/exp/lib_tknock1.so tools/exp/lib_tknock2.so tools/exp/lib_tknock16.so tools/exp/lib_tknock18.so
