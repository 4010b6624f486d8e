#!/bin/bash
# Round-2 measurement suite (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats of the same command,
# PMC traffic passes for the dominant kernel.  Outputs under gpurun_out/r02_m/.
set -o pipefail
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r02_m; mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/kt.err; echo "kt rc=$?"
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_w.err; echo "pmc_w rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 tools/prof_step.py > /dev/null 2> $O/pmc_f.err; echo "pmc_f rc=$?"
python3 tools/pmc_summary.py k_step $O/pmc_w/*/*counter_collection.csv $O/pmc_f/*/*counter_collection.csv > $O/pmc_traffic_k_step.txt 2>&1
cat $O/pmc_traffic_k_step.txt
rm -rf $O/kt $O/pmc_w $O/pmc_f
head -c 1500 $O/bench.json; echo; head -8 $O/bench_kernel_stats.csv
