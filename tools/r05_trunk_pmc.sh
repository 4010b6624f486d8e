#!/bin/bash
# PMC rows of the 15x10 trunk forward (k_policy_trunk) for two builds of the library, one counter group per pass (no other trace
# domains: gpurun refuses --pmc together with the hip / marker traces).  usage: bash tools/r05_trunk_pmc.sh <tag> <libA.so> <libB.so>
# The verdict's table (r4, next 2): SQ_INSTS_LDS, SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT, TCP read requests, before / after.
set -o pipefail
TAG=${1:-r05}; shift
R=$GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG}_trunkpmc
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for LIB in "$@"; do
  name=$(basename $LIB .so)
  export INVERSUS_HIP_LIB=$R/$LIB
  i=0
  for set in "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
             "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    REPS=24 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$OUT/${name}_$i -- python3 $R/tools/prof_policy.py > /dev/null 2> $R/$OUT/${name}_$i.err || echo "$name pass $i ($set) failed" >> $R/$OUT/failed.txt
    echo "$name pass $i done: $set"
  done
done
cd $R
python3 - <<PY
import csv, glob, collections, os
out = "$OUT"
with open(out + "/pmc_summary.txt", "w") as fh:
    for lib in sorted({os.path.basename(d).rsplit("_", 1)[0] for d in glob.glob(out + "/*_[0-9]")}):
        agg, dur = collections.defaultdict(list), []
        for f in glob.glob(f"{out}/{lib}_*/**/*counter_collection.csv", recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "k_policy_trunk" in r["Kernel_Name"]]
            per = collections.defaultdict(list)
            for r in rows:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for c, v in per.items():
                agg[c] += v[len(v) // 4:]                      # skip the first quarter (clock ramp)
        for f in glob.glob(f"{out}/{lib}_*/**/*kernel_trace.csv", recursive=True):
            v = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_policy_trunk" in r["Kernel_Name"]]
            dur += v[len(v) // 4:]
        fh.write(f"{lib}: k_policy_trunk, per launch of 16 384 samples; duration under the counters: mean {sum(dur) / max(1, len(dur)):.0f} ns (n {len(dur)})\n")
        for c, v in sorted(agg.items()):
            fh.write(f"   {c:32s} mean {sum(v) / len(v):18.1f}  n {len(v)}\n")
print(open(out + "/pmc_summary.txt").read())
PY
