#!/bin/bash
# Round-2 measurement suite (run on the GPU box through gpurun): the bench line, rocprofv3 kernel stats of the same command,
# PMC traffic passes for the dominant kernel, SQ counters of the step kernels and of the fused policy trunk, kernel stats of the
# fused PPO update.  Outputs under gpurun_out/r02_m/ (copied into profiles/ by hand).  Counters are collected in their own
# runs with --kernel-trace only.
set -o pipefail
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r02_m; mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/kt.err; echo "kt rc=$?"
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
pmc() {  # name, target script, counters...
  local name=$1 target=$2; shift 2
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$name -- python3 $target > /dev/null 2> $O/pmc_$name.err; echo "pmc $name rc=$?"
}
pmc w tools/prof_step.py WRITE_SIZE
pmc f tools/prof_step.py FETCH_SIZE
python3 tools/pmc_summary.py k_step $O/pmc_w/*/*counter_collection.csv $O/pmc_f/*/*counter_collection.csv > $O/pmc_traffic_k_step4.txt 2>&1
pmc s1 tools/prof_step.py SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES
python3 tools/pmc_summary.py k_step $O/pmc_s1/*/*counter_collection.csv > $O/pmc_sq_k_step4_easy.txt 2>&1
export DIFF=hard; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmc_s2 -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_s2.err; unset DIFF
python3 tools/pmc_summary.py k_step $O/pmc_s2/*/*counter_collection.csv > $O/pmc_sq_k_step4_hard.txt 2>&1
export INVERSUS_STEP_KERNEL=wave DIFF=hard; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmc_s3 -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_s3.err; unset INVERSUS_STEP_KERNEL DIFF
python3 tools/pmc_summary.py k_step $O/pmc_s3/*/*counter_collection.csv > $O/pmc_sq_k_step_wave_hard.txt 2>&1
export REPS=12; pmc p1 tools/prof_policy.py SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pmc p2 tools/prof_policy.py SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS
unset REPS; python3 tools/pmc_summary.py k_policy_trunk $O/pmc_p1/*/*counter_collection.csv $O/pmc_p2/*/*counter_collection.csv > $O/policy_trunk_pmc.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_pol -- python3 tools/prof_policy.py > /dev/null 2> $O/kt_pol.err
find $O/kt_pol -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/policy_trunk_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_upd -- python3 tools/agent_update_profile.py --no-prof > $O/update_under_rocprof.json 2> $O/kt_upd.err
find $O/kt_upd -name "*kernel_stats.csv" | head -1 | xargs -I{} head -40 {} > $O/update_kernel_stats_top40.csv
rm -rf $O/kt $O/kt_pol $O/kt_upd $O/pmc_w $O/pmc_f $O/pmc_s1 $O/pmc_s2 $O/pmc_s3 $O/pmc_p1 $O/pmc_p2
for f in pmc_traffic_k_step4 pmc_sq_k_step4_easy pmc_sq_k_step4_hard pmc_sq_k_step_wave_hard policy_trunk_pmc; do echo "== $f"; cat $O/$f.txt; done
head -4 $O/bench_kernel_stats.csv; head -3 $O/policy_trunk_kernel_stats.csv; head -12 $O/update_kernel_stats_top40.csv | cut -c1-160; cat $O/update_under_rocprof.json | tail -1
