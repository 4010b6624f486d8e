#!/usr/bin/env python3
"""Kernel time of the per-env GAE scan without the Python wrapper (one pre-marshalled ctypes call per launch; HIP events over
200 back-to-back launches, median of 5) -- the wrapper's tensor allocations cost more host time than the kernel runs."""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd import _lib as L
lib = L.load()
res = {}
cfgs = ((128, 16384), (128, 131072), (512, 4096), (128, 4096))
if os.environ.get("GAE_CFG"):                                          # e.g. GAE_CFG=128,16384: one configuration (for a rocprofv3 row of its own)
    cfgs = (tuple(int(x) for x in os.environ["GAE_CFG"].split(",")),)
for T, n in cfgs:
    r = torch.randn(T, n, device="cuda"); v = torch.randn(T, n, device="cuda"); d = (torch.rand(T, n, device="cuda") < 0.03).to(torch.uint8)
    adv, ret = torch.empty_like(r), torch.empty_like(r)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (C.c_void_p(r.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(d.data_ptr()), T, n, float(np.float32(0.99)),
            float(np.float32(0.99 * 0.95)), None, L.INV_GAE_PER_ENV, C.c_void_p(adv.data_ptr()), C.c_void_p(ret.data_ptr()), st)
    for _ in range(20):
        lib.inv_gae(*args)
    out = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            lib.inv_gae(*args)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 200 * 1e3)
    us = sorted(out)[2]
    res[f"per_env T={T} N={n}"] = {"us": round(us, 2), "GBps": round(17 * T * n / us / 1e3, 1), "frac_of_8TBps": round(17 * T * n / us / 1e3 / 8000, 3)}
print(json.dumps(res, indent=1))
