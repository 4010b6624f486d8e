#!/usr/bin/env python3
"""Kernel-time decomposition of the step path on the GPU (HIP events, median of repeats).
Not part of the product: a measurement helper for profiles/."""
import os, sys, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch

def timeit(fn, reps=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts)//2] * 1e3  # us

def main():
    n = int(os.environ.get("N", 16384)); W = int(os.environ.get("W", 15)); H = int(os.environ.get("H", 10))
    res = {}
    for diff in ("easy", "hard"):
        b = DeviceEnvBatch(n, W, H, opponent="dummy", difficulty=diff, opp_seed=1)
        b.seed(np.arange(n, dtype=np.int64)); b.reset()
        a1 = b.uniform_actions(1, 0); a2 = torch.zeros_like(a1)
        full = b._alloc_out((), True, False, False, None)
        both = b._alloc_out((), True, True, False, None)
        noobs = b._alloc_out((), False, False, False, None)
        for t in range(100):   # get into a steady state with bullets flying
            b.step(b.uniform_actions(1, t), None, auto_reset=True, out=full)
        res[f"{diff}:step+obs(dummy)"] = timeit(lambda: b.step(a1, None, auto_reset=True, out=full))
        res[f"{diff}:step+obs(a2 given)"] = timeit(lambda: b.step(a1, a2, auto_reset=True, out=full))
        res[f"{diff}:step+obs+obs_p2(a2 given)"] = timeit(lambda: b.step(a1, a2, auto_reset=True, out=both))
        res[f"{diff}:step no obs(dummy)"] = timeit(lambda: b.step(a1, None, auto_reset=True, out=noobs))
        res[f"{diff}:step no obs(a2 given)"] = timeit(lambda: b.step(a1, a2, auto_reset=True, out=noobs))
        o = (full.obs, full.extra)
        res[f"{diff}:render only"] = timeit(lambda: b.render(0, out=o))
    x = torch.empty(n * 12 * W * H, dtype=torch.float32, device="cuda")
    res["torch fill_ of obs buffer"] = timeit(lambda: x.fill_(1.0))
    y = torch.empty_like(x)
    res["torch copy_ obs->obs (r+w)"] = timeit(lambda: y.copy_(x))
    big = torch.empty(256 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda")
    res["torch fill_ 256MiB (us)"] = timeit(lambda: big.fill_(1.0), reps=50)
    print(json.dumps({"n": n, "W": W, "H": H, "us": res}, indent=1))

if __name__ == "__main__":
    main()
