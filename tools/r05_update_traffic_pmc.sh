#!/bin/bash
# Memory-side traffic of ONE fused PPO minibatch (16 384 samples, 15x10) from the PMC counters, kernel by kernel: separate
# `rocprofv3 --kernel-trace --pmc WRITE_SIZE` and `--pmc FETCH_SIZE` passes over tools/agent_update_profile.py (the program directly
# behind `--`, no other trace domains), summarised per kernel as KiB x 1024 per launch x launches per minibatch.  FETCH_SIZE is doubled
# (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B
# stores and f32 atomics.  Both count fabric requests, Infinity-Cache hits included.  usage: bash tools/r05_update_traffic_pmc.sh [tag]
set -o pipefail
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r05}; O=gpurun_out/${TAG}_updpmc; mkdir -p $O
timeout -k 10 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 tools/agent_update_profile.py --no-prof > $O/pw.out 2> $O/pw.err; echo "WRITE_SIZE pass rc=$?"
timeout -k 10 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 tools/agent_update_profile.py --no-prof > $O/pf.out 2> $O/pf.err; echo "FETCH_SIZE pass rc=$?"
python3 - <<PY > $O/update_traffic_per_minibatch.txt
import csv, glob, collections
acc = {"WRITE_SIZE": collections.defaultdict(list), "FETCH_SIZE": collections.defaultdict(list)}
for f in glob.glob("$O/p[wf]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in acc:
            acc[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
MB = 24                                                      # minibatches in the script: (1 warm-up + 5) updates x 4
names = sorted(set(acc["WRITE_SIZE"]) | set(acc["FETCH_SIZE"]))
rows, tw, tf = [], 0.0, 0.0
for k in names:
    w, f = acc["WRITE_SIZE"].get(k, []), acc["FETCH_SIZE"].get(k, [])
    calls = max(len(w), len(f))
    wb = sum(w) * 1024 / MB; fb = 2 * sum(f) * 1024 / MB     # bytes per minibatch (collection kernels included: they are listed, see below)
    rows.append((wb + fb, k, calls, wb, fb))
print("kernel | launches (24 minibatches + 6 rollouts of 4 steps) | written MB / minibatch | fetched MB / minibatch (FETCH_SIZE x 2) | sum")
upd = 0.0
for s, k, calls, wb, fb in sorted(rows, reverse=True)[:40]:
    print(f"{k[:110]:110s} | {calls:5d} | {wb/1e6:9.1f} | {fb/1e6:9.1f} | {s/1e6:9.1f}")
tot = sum(r[0] for r in rows)
coll = sum(r[0] for r in rows if ("false, false" in r[1] or "false, true" in r[1] or "k_step4" in r[1]))
print(f"all kernels: {tot/1e9:.2f} GB per minibatch-equivalent; of that the rollout's inference forwards / env steps: {coll/1e9:.2f} GB")
PY
cat $O/update_traffic_per_minibatch.txt | cut -c1-220
rm -rf $O/pw $O/pf
