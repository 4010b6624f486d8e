#!/usr/bin/env python3
"""BASELINE config C1 end to end at the REFERENCE's hyper-parameters (verdict r4, next-round item 3).

    python tools/c1_reference_curve.py [--total_steps 200000] [--seeds 0 1 2] [--out gpurun_out/c1]

Runs `python -m inversus_reinforcement_learning_amd.training --mode vs_dummy --num_envs 4 --total_steps 200000
--opponent_difficulty easy` with the package defaults (= /root/reference/inversus_rl/ppo_agent.py:16-28: lr 1e-4, batch 512, 4 epochs,
clip 0.2, entropy 0.02, value 0.1; steps_per_env = max(2048 // 4, 128) = 512 -> 2 048 samples per update like the reference's log rows)
once per (seed, path): path "f32" = the torch f32 module for act and update (reference numerics), path "fused" = --fused (bf16 HIP
trunk / heads kernels for act and update).  GAE layout "reference_flat" (what the reference's run used, its interleaved chain) unless
--gae_layout says otherwise.  Each run is a child process; its training_log.csv (the reference's 8 columns) is copied to
<out>/c1_<path>_<layout>_seed<k>_training_log.csv and a summary table is printed / written as <out>/c1_summary.json:
first 3 rows and the mean of the last 10 rows of win_rate / avg_ep_len / entropy, next to the reference-held numbers
(/root/reference/runs/inversus_random_base_v1/training_log.csv:2-4, :90-99 -- quoted as constants below; the file is not copied).
Parity is STATISTICAL here (global-RNG dummy, sampling, shuffles: SURVEY 8a R14/R17): the check is the band, not the bits."""
import argparse
import csv
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# reference-held curve, /root/reference/runs/inversus_random_base_v1/training_log.csv (rows 2-4 and the mean / range of rows 90-99)
REFERENCE = {"first_rows": [{"step": 2048, "win_rate": 0.0, "avg_ep_len": 500.0, "entropy": 2.5411},
                            {"step": 4096, "win_rate": 0.4545, "avg_ep_len": 289.36, "entropy": 2.5463},
                            {"step": 6144, "win_rate": 0.5294, "avg_ep_len": 325.65, "entropy": 2.5358}],
             "last10": {"win_rate": (0.33, 0.37, 0.349), "avg_ep_len": (380.08, 399.18, 388.24), "entropy": (2.3304, 2.4164, 2.3897),
                        "avg_reward": (2.6369, 3.1596, 2.8742), "episodes_at_200k": 548}}


def read_rows(path):
    with open(path) as fh:
        return [{k: float(v) for k, v in row.items()} for row in csv.DictReader(fh)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--total_steps", type=int, default=200_000)
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2])
    ap.add_argument("--paths", nargs="+", default=["f32", "fused"])
    ap.add_argument("--gae_layout", default="reference_flat")
    ap.add_argument("--graph", action="store_true", help="fused path: --graph (one HIP graph per rollout); tag 'fusedgraph'")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "c1"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    summary = {"reference": REFERENCE, "runs": []}
    for path in args.paths:
        for seed in args.seeds:
            tag = f"c1_{path}{'graph' if (args.graph and path == 'fused') else ''}_{args.gae_layout}_seed{seed}"
            log_dir = os.path.join(args.out, "run_" + tag)
            cmd = [sys.executable, "-m", "inversus_reinforcement_learning_amd.training", "--mode", "vs_dummy", "--num_envs", "4",
                   "--total_steps", str(args.total_steps), "--opponent_difficulty", "easy", "--seed", str(seed), "--log_dir", log_dir,
                   "--gae_layout", args.gae_layout] + (["--fused"] if path == "fused" else []) + (["--graph"] if (args.graph and path == "fused") else [])
            t0 = time.time()
            p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            secs = time.time() - t0
            print(f"[{tag}] rc={p.returncode} {secs:.0f} s", flush=True)
            if p.returncode != 0:
                print(p.stdout[-3000:], flush=True)
                summary["runs"].append({"tag": tag, "rc": p.returncode})
                continue
            dst = os.path.join(args.out, tag + "_training_log.csv")
            shutil.copy(os.path.join(log_dir, "training_log.csv"), dst)
            shutil.rmtree(log_dir, ignore_errors=True)                     # the 41 MB checkpoints are not evidence
            rows = read_rows(dst)
            last = rows[-10:]

            def stat(k):
                v = [r[k] for r in last]
                return round(min(v), 4), round(max(v), 4), round(sum(v) / len(v), 4)
            rec = {"tag": tag, "path": path, "seed": seed, "gae_layout": args.gae_layout, "seconds": round(secs, 1), "rows": len(rows),
                   "env_steps_per_s": round(args.total_steps / secs), "episodes": int(rows[-1]["episode"]),
                   "first_rows": [{k: round(r[k], 4) for k in ("step", "win_rate", "avg_ep_len", "entropy")} for r in rows[:3]],
                   "last10": {k: stat(k) for k in ("win_rate", "avg_ep_len", "entropy", "avg_reward")}}
            summary["runs"].append(rec)
            print(json.dumps(rec), flush=True)
    with open(os.path.join(args.out, f"c1_summary_{args.gae_layout}{'_graph' if args.graph else ''}.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print("| path | seed | win rate (last 10 rows: min / max / mean) | avg_ep_len | entropy | episodes | s |")
    print("|---|---|---|---|---|---|---|")
    r = REFERENCE["last10"]
    print(f"| reference (runs/inversus_random_base_v1) | ? | {r['win_rate']} | {r['avg_ep_len']} | {r['entropy']} | {r['episodes_at_200k']} | - |")
    for rec in summary["runs"]:
        if "last10" in rec:
            l = rec["last10"]
            print(f"| {rec['path']} ({rec['gae_layout']}) | {rec['seed']} | {l['win_rate']} | {l['avg_ep_len']} | {l['entropy']} | {rec['episodes']} | {rec['seconds']} |")


if __name__ == "__main__":
    main()
