#!/bin/bash
# Timing probes of the fc1 main loop (tools/exp/lib_knock{1,2,3}.so: -DFC1_KNOCK = 1 no LDS fragment reads after the first K-tile,
# 2 no DMA requests, 3 neither; results are wrong on purpose) against the shipped library, each in its own process, alternated twice.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in "" knock1 knock2 knock4; do
    if [ -z "$v" ]; then lib=inversus_reinforcement_learning_amd/libinversus_hip.so; else lib=tools/exp/lib_$v.so; fi
    INVERSUS_HIP_LIB=$PWD/$lib python tools/fc1_bench.py 16384 4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${v:-shipped}', {k: d[k]['ours_ms_median'] for k in ('fwd','dfeat','dw')})"
  done
done
