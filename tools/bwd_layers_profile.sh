cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bwd_layers_prof; mkdir -p $O
export INVERSUS_BWD_LAYERS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/agent_update_profile.py --no-prof > $O/out.json 2> $O/err.txt
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} head -16 {} | cut -c1-200
rm -rf $O/kt
