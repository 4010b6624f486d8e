#!/bin/bash
# PMC rows of the fc1 kernels (shipped schedule) -- one gpurun call.  usage: bash tools/r04_fc1_pmc.sh <tag>
set -o pipefail
TAG=${1:-r04_v2}
OUT=gpurun_out/${TAG}_fc1pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$OUT/pmc_$name -- python3 $R/tools/fc1_bench.py 16384 1 > /dev/null 2> $R/$OUT/pmc_$name.err
done
cd $R
python3 - <<PY
import csv, glob, collections
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fc1" in k:
            agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/pmc_GRBM*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fc1" in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out + "/pmc_summary.txt", "w") as fh:
    for k, d in agg.items():
        fh.write(k + (f"   duration under the GRBM pass: mean {sum(dur[k])/len(dur[k]):.0f} ns" if dur.get(k) else "") + "\n")
        for c, v in sorted(d.items()):
            fh.write(f"   {c:32s} mean {sum(v)/len(v):16.1f}  n {len(v)}\n")
print(open(out + "/pmc_summary.txt").read())
PY
