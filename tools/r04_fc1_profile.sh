#!/bin/bash
# rocprofv3 rows of the first Linear's kernels (csrc/inv_fc1.hip) next to the library GEMMs of the same products, one gpurun call.
# usage (on the GPU box): bash tools/r04_fc1_profile.sh <tag>
set -o pipefail
TAG=${1:-r04_v1}
OUT=gpurun_out/${TAG}_fc1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt -- python3 $R/tools/fc1_bench.py 16384 4 > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/kt.err
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$OUT/pmc_$name -- python3 $R/tools/fc1_bench.py 16384 1 > /dev/null 2> $R/$OUT/pmc_$name.err
done
cd $R
python3 - <<PY
import csv, glob, collections, os
out = "$OUT"
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(out + "/kernel_stats_top.csv", "w") as fh:
        w = csv.writer(fh); w.writerow(["Name", "Calls", "AverageNs", "MinNs", "MaxNs"])
        for r in rows[:14]:
            w.writerow([r["Name"][:110], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fc1" in k or "Cijk" in k:
            agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_summary.txt", "w") as fh:
    for k, d in agg.items():
        fh.write(k + "\n")
        for c, v in sorted(d.items()):
            fh.write(f"   {c:32s} mean {sum(v)/len(v):16.1f}  n {len(v)}\n")
print(open(out + "/kernel_stats_top.csv").read())
print(open(out + "/pmc_summary.txt").read())
PY
