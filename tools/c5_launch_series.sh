#!/bin/bash
# Per-launch durations of the 32x32 f32 step (805 MB of stores per launch) in dispatch order over a LONG run (default 6 000 launches,
# ~0.8 s): is the spread between runs (119 vs 142 us averages) the box, or where on the clock ramp after an idle period a short run
# sits?  rocprofv3 --kernel-trace, program directly after `--`.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r03_c5series; mkdir -p $O
CFG=c5 LAUNCHES=${LAUNCHES:-6000} rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/prof_step_cfg.py > /dev/null 2> $O/kt.err
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, statistics as st
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_step" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
t0 = int(rows[0]["Start_Timestamp"])
print("launches", len(d), " run length %.2f s" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e9))
blk = max(1, len(d) // 24)
print("mean us per block of %d launches (time since the first launch, ms: mean):" % blk)
print("  " + "  ".join("%d:%.0f" % ((int(rows[i]["Start_Timestamp"]) - t0) / 1e6, st.mean(d[i:i + blk])) for i in range(0, len(d) - blk + 1, blk)))
print("first 60: mean %.1f sd %.1f | last 600: mean %.1f sd %.1f min %.1f max %.1f" % (st.mean(d[:60]), st.pstdev(d[:60]), st.mean(d[-600:]), st.pstdev(d[-600:]), min(d[-600:]), max(d[-600:])))
PY
rm -rf $O/kt
echo "--- rocm-smi right after the run"
rocm-smi --showclocks --showpower --showuniqueid 2>/dev/null | grep -E "Unique ID:|fclk|mclk|sclk|Power \(W\)"
