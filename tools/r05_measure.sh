#!/bin/bash
# Round-5 measurement suite (run on the GPU box through gpurun; outputs under gpurun_out/<tag>_m/, copied into profiles/ by hand):
#   bench line (default run and the driver's short form), rocprofv3 --kernel-trace --stats of the headline loop, WRITE_SIZE / FETCH_SIZE
#   passes of the streaming and cache-resident step (separate --pmc runs, program directly behind `--`), the fused update and the
#   self-play collector under rocprofv3 WITH --marker-trace (the roctx phases: profiles/r05_roctx.md), the fc1 products against the
#   library, the C5 update.  Every profiled program sits directly behind `--`, each under its own timeout.
set -o pipefail
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r05_end}; O=gpurun_out/${TAG}_m; mkdir -p $O
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID:" | head -1 | tee $O/gpu_unique_id.txt
python3 bench.py > $O/bench_easy_with_extras.json 2> $O/bench.err; echo "bench rc=$?"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style_20_steps.json 2> $O/bench20.err; echo "bench (driver style) rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline --no-extras --no-cache-resident > $O/bench_under_rocprof.json 2> $O/kt.err; echo "kt bench rc=$?"
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} head -8 {} > $O/bench_kernel_stats_headline_only.csv; rm -rf $O/kt
for c in easy_streaming easy; do
  export CFG=$c LAUNCHES=40
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw_$c -- python3 tools/prof_step_cfg.py > /dev/null 2> $O/pw_$c.err
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf_$c -- python3 tools/prof_step_cfg.py > /dev/null 2> $O/pf_$c.err
  python3 tools/pmc_summary.py k_step $O/pw_$c/*/*counter_collection.csv $O/pf_$c/*/*counter_collection.csv > $O/pmc_$c.txt 2>&1
  rm -rf $O/pw_$c $O/pf_$c; echo "== pmc $c"; cat $O/pmc_$c.txt
done
unset CFG LAUNCHES
python3 tools/agent_update_profile.py --no-prof > $O/update_speed.json 2>/dev/null; cat $O/update_speed.json
timeout -k 10 400 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $O/kt_upd -- python3 tools/agent_update_profile.py --no-prof > $O/update_under_rocprof.json 2> $O/kt_upd.err; echo "kt update rc=$?"
find $O/kt_upd -name "*kernel_stats.csv" | head -1 | xargs -I{} head -40 {} > $O/update_kernel_stats_top40.csv
find $O/kt_upd -name "*marker_api_stats.csv" | head -1 | xargs -I{} cp {} $O/update_marker_api_stats.csv; rm -rf $O/kt_upd
timeout -k 10 400 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $O/kt_col -- python3 tools/collector_profile.py --no-prof > $O/collector_under_rocprof.json 2> $O/kt_col.err; echo "kt collector rc=$?"
find $O/kt_col -name "*kernel_stats.csv" | head -1 | xargs -I{} head -24 {} > $O/collector_kernel_stats_top24.csv
find $O/kt_col -name "*marker_api_stats.csv" | head -1 | xargs -I{} cp {} $O/collector_marker_api_stats.csv; rm -rf $O/kt_col
echo "Cijk rows: update $(grep -c Cijk $O/update_kernel_stats_top40.csv), collector $(grep -c Cijk $O/collector_kernel_stats_top24.csv)"
python3 tools/fc1_bench.py 16384 10 > $O/fc1_bench_vs_library.json 2>/dev/null; cat $O/fc1_bench_vs_library.json
python3 tools/c5_update_speed.py --no-torch > $O/c5_update_speed.json 2>/dev/null; tail -1 $O/c5_update_speed.json
python3 -c "
import json; d=json.load(open('$O/bench_easy_with_extras.json')); print(json.dumps({k:d[k] for k in ('value','ms_per_step','value_cache_resident','roofline')}, indent=0)[:1800]); print({k: (v if not isinstance(v, dict) else {a:b for a,b in list(v.items())[:6]}) for k, v in d.get('extras', {}).items()})" 2>&1 | cut -c1-3500
cut -c1-160 $O/update_kernel_stats_top40.csv | head -16; cat $O/update_marker_api_stats.csv $O/collector_marker_api_stats.csv 2>/dev/null; true
