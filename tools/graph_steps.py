#!/usr/bin/env python3
"""Headline loop (vs_dummy easy, 16 384 envs, f32 observations into 32 distinct slots) as stream launches vs ONE HIP graph of K step
launches: us per vector step, wall clock over `reps` regions of K steps each.  Usage: python tools/graph_steps.py [K ...]"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch, StepResult
n, SLOTS = 16384, 32
Ks = [int(x) for x in sys.argv[1:]] or [20, 100, 1000]
b = DeviceEnvBatch(n, 15, 10, 16, opponent="dummy", difficulty="easy", opp_seed=1)
b.seed(np.arange(n, dtype=np.int64)); b.reset()
so = b._alloc_out((SLOTS,), True, False, False, None)
slots = [StepResult(**{k: (getattr(so, k)[j] if getattr(so, k) is not None else None) for k in StepResult.__slots__}) for j in range(SLOTS)]
stp = [b.make_stepper(slots[j], None, auto_reset=True) for j in range(SLOTS)]
a1 = torch.stack([b.uniform_actions(1234, t) for t in range(max(Ks))])
base = a1.data_ptr()
for t in range(64):
    stp[t % SLOTS](base + (t % len(a1)) * n)
torch.cuda.synchronize()
out = {}
for K in Ks:
    def eager():
        for k in range(K):
            stp[k % SLOTS](base + k * n)
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        # the steppers were bound to the stream that was current when they were made: re-make them on the capture stream
        cst = [b.make_stepper(slots[j], None, auto_reset=True) for j in range(SLOTS)]
        for k in range(K):
            cst[k % SLOTS](base + k * n)
    res = {}
    for name, fn in (("stream launches", eager), ("one graph", g.replay), ("stream launches again", eager), ("one graph again", g.replay)):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e6)
        res[name] = round(sorted(ts)[3], 3)
    b.check()
    out[f"K={K}"] = res
print(json.dumps(out))
