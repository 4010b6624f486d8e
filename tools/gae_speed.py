#!/usr/bin/env python3
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import gae
from tools.ablate import timeit
res = {}
for T, n in ((128, 16384), (128, 131072), (512, 4096)):
    r = torch.randn(T, n, device="cuda"); v = torch.randn(T, n, device="cuda"); d = torch.rand(T, n, device="cuda") < 0.03
    for layout in ("per_env", "reference_flat"):
        us = timeit(lambda: gae(r, v, d, layout=layout), reps=30, warm=3)   # includes the wrapper's allocations; kernel time: rocprof
        res[f"GAE {layout} T={T} N={n}"] = {"us": round(us, 2), "GB/s": round(17 * T * n / us / 1e3, 1)}
print(json.dumps(res, indent=1))
