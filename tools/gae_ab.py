#!/usr/bin/env python3
"""A/B of library builds on the per-env GAE kernel: python tools/gae_ab.py lib1.so ... (HIP events around 50 launches)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, ROOT)
    from inversus_reinforcement_learning_amd.device_env import gae
    res = {}
    for T, n in ((128, 16384), (128, 131072)):
        torch.manual_seed(0)
        r = torch.randn(T, n, device="cuda"); v = torch.randn(T, n, device="cuda"); d = (torch.rand(T, n, device="cuda") < 0.03).to(torch.uint8)
        a, _ = gae(r, v, d)
        out = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): gae(r, v, d)
            e1.record(); torch.cuda.synchronize(); out.append(e0.elapsed_time(e1) / 50 * 1e3)
        res[f"T={T} N={n} us"] = round(min(out), 2)
        res[f"T={T} N={n} sum"] = float(a.double().abs().sum())
    print(json.dumps(res)); sys.exit(0)
for lib in sys.argv[1:]:
    env = dict(os.environ, INVERSUS_HIP_LIB=os.path.join(ROOT, lib))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
    print(lib, r.stdout.strip() or r.stderr[-500:], flush=True)
