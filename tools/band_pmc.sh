#!/bin/bash
# PMC rows of the banded forward kernel, one counter group per pass (no other trace domains).  usage: bash tools/band_pmc.sh <tag>
set -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/${TAG}_bandpmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE" "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_EA_RD_UNCACHED_32B_sum TCC_BUBBLE_sum" \
           "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$OUT/pmc_$i -- python3 $R/tools/band_fwd_loop.py 16384 4 > /dev/null 2> $R/$OUT/pmc_$i.err || echo "pass $i ($set) failed" >> $R/$OUT/failed.txt
  echo "pass $i done: $set"
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
        if "banded" in k:
            agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/pmc_*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "banded" in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out + "/pmc_summary.txt", "w") as fh:
    for k, d in agg.items():
        fh.write(k + (f"   duration over all passes: mean {sum(dur[k])/len(dur[k]):.0f} ns" if dur.get(k) else "") + "\n")
        for c, v in sorted(d.items()):
            fh.write(f"   {c:32s} mean {sum(v)/len(v):18.1f}  n {len(v)}\n")
print(open(out + "/pmc_summary.txt").read())
PY
