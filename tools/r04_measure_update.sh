#!/bin/bash
# rocprofv3 --kernel-trace --stats of the fused update and of the self-play collector (the verdict's check: no Cijk_* GEMM-library
# kernel left in either).  Run on the GPU box through gpurun; outputs under gpurun_out/<tag>_upd/.
set -o pipefail
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r04_v1}; O=gpurun_out/${TAG}_upd; mkdir -p $O
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID:" | head -1 | tee $O/gpu_unique_id.txt
python3 tools/agent_update_profile.py --no-prof > $O/update_speed.json 2>/dev/null; cat $O/update_speed.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_upd -- python3 tools/agent_update_profile.py --no-prof > $O/update_under_rocprof.json 2> $O/kt_upd.err; echo "kt update rc=$?"
find $O/kt_upd -name "*kernel_stats.csv" | head -1 | xargs -I{} head -40 {} > $O/update_kernel_stats_top40.csv; rm -rf $O/kt_upd
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_col -- python3 tools/collector_profile.py --no-prof > $O/collector_under_rocprof.json 2> $O/kt_col.err; echo "kt collector rc=$?"
find $O/kt_col -name "*kernel_stats.csv" | head -1 | xargs -I{} head -24 {} > $O/collector_kernel_stats_top24.csv; rm -rf $O/kt_col
echo "Cijk rows in the update profile: $(grep -c Cijk $O/update_kernel_stats_top40.csv); in the collector profile: $(grep -c Cijk $O/collector_kernel_stats_top24.csv)"
cut -c1-150 $O/update_kernel_stats_top40.csv | head -24; cut -c1-150 $O/collector_kernel_stats_top24.csv | head -10
