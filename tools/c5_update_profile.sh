#!/bin/bash
# rocprofv3 kernel stats of one C5 (32 x 32) fused PPO iteration at 16 384 envs (program directly after `--`).
set -o pipefail
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/${1:-r04}_c5; mkdir -p $O
python3 tools/c5_update_speed.py ${C5_ARGS:---no-torch} > $O/c5_speed.json 2> $O/c5_speed.err; echo "speed rc=$?"; tail -1 $O/c5_speed.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/c5_update_speed.py --no-torch > $O/c5_under_rocprof.json 2> $O/kt.err; echo "kt rc=$?"
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} head -30 {} > $O/c5_update_kernel_stats_top30.csv; rm -rf $O/kt
cut -c1-150 $O/c5_update_kernel_stats_top30.csv
