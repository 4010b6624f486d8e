#!/bin/bash
# Round-3 measurement suite (run on the GPU box through gpurun): rocprofv3 --kernel-trace --stats and, in separate passes,
# --pmc WRITE_SIZE / --pmc FETCH_SIZE for EVERY benched configuration of the step kernel (tools/prof_step_cfg.py; the program
# follows `--` directly), then the bench line and its kernel stats.  Outputs under gpurun_out/r03_m/ (copied into profiles/).
set -o pipefail
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r03_m; mkdir -p $O
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID:" | tee $O/gpu_unique_id.txt
CFGS=${CFGS:-"easy hard selfplay_f32 selfplay_packed hard_packed c5 c5_packed easy_streaming hard_streaming rollout"}
for c in $CFGS; do
  export CFG=$c LAUNCHES=200; [ $c = rollout ] && export LAUNCHES=8; [ ${c:0:2} = c5 ] && export LAUNCHES=60
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -- python3 tools/prof_step_cfg.py > /dev/null 2> $O/kt_$c.err; echo "kt $c rc=$?"
  f=$(find $O/kt_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|k_step" $f > $O/kstats_$c.csv
  export LAUNCHES=40; [ $c = rollout ] && export LAUNCHES=3
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw_$c -- python3 tools/prof_step_cfg.py > /dev/null 2> $O/pw_$c.err; echo "pmc-w $c rc=$?"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf_$c -- python3 tools/prof_step_cfg.py > /dev/null 2> $O/pf_$c.err; echo "pmc-f $c rc=$?"
  python3 tools/pmc_summary.py k_step $O/pw_$c/*/*counter_collection.csv $O/pf_$c/*/*counter_collection.csv > $O/pmc_$c.txt 2>&1
  rm -rf $O/kt_$c $O/pw_$c $O/pf_$c
  echo "== $c"; cat $O/kstats_$c.csv | cut -c1-200; cat $O/pmc_$c.txt
done
unset CFG LAUNCHES
if [ -z "$NO_BENCH" ]; then
  python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline --no-extras --no-streaming > $O/bench_under_rocprof.json 2> $O/kt.err; echo "kt bench rc=$?"
  find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; rm -rf $O/kt
  head -5 $O/bench_kernel_stats.csv | cut -c1-220; python3 -c "import json;d=json.load(open('$O/bench.json'));print(json.dumps({k:d[k] for k in ('value','ms_per_step','roofline')}, indent=1))"
fi
if [ -z "$NO_UPDATE" ]; then
  python3 tools/agent_update_profile.py > $O/update_torch_profile.txt 2>&1; grep "update ms" $O/update_torch_profile.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_upd -- python3 tools/agent_update_profile.py --no-prof > $O/update_under_rocprof.json 2> $O/kt_upd.err; echo "kt update rc=$?"
  find $O/kt_upd -name "*kernel_stats.csv" | head -1 | xargs -I{} head -40 {} > $O/update_kernel_stats_top40.csv; rm -rf $O/kt_upd
  python3 tools/gae_kernel_time.py > $O/gae_kernel_time.json 2>/dev/null; cat $O/gae_kernel_time.json | tr -d '\n'; echo
  python3 tools/two_stream_steps.py > $O/two_stream_steps.json 2>/dev/null; cat $O/two_stream_steps.json | tr -d '\n'; echo
fi
