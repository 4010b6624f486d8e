#!/usr/bin/env python3
"""A/B of the two step kernels (wave-per-env vs four envs per wave) and the two observation store policies (plain / auto =
write-through where line-aligned) on the configurations bench.py reports:
us per vector step at N = 16384 (HIP events over 256 back-to-back launches, median of 5)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
n, K = 16384, 16

def timed(fn, reps=256, warm=64):
    for _ in range(warm): fn()
    out = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(out)[2]

def cfg(kind, W, H, opponent, difficulty, obs_p2, bits, T=1):
    kind, store = kind.split("/")
    b = DeviceEnvBatch(n, W, H, K, opponent=opponent, difficulty=difficulty, opp_seed=1, step_kernel=kind, obs_store=store)
    b.seed(np.arange(n, dtype=np.int64)); b.reset()
    acts = torch.stack([b.uniform_actions(5, t) for t in range(64)])
    a2 = torch.stack([b.uniform_actions(6, t) for t in range(64)]) if opponent == "selfplay" else None
    if T < 0:                                                        # single launches, but into |T| DISTINCT output buffers (a rollout's footprint)
        T = -T
        from inversus_reinforcement_learning_amd.device_env import StepResult
        o = b._alloc_out((T,), True, obs_p2, False, None, bits)
        outs = [StepResult(**{k2: (getattr(o, k2)[j] if getattr(o, k2) is not None else None) for k2 in StepResult.__slots__}) for j in range(T)]
        st = [b.make_stepper(outs[j], None if a2 is None else a2[j], auto_reset=True) for j in range(T)]
        k = [0]
        def f():
            j = k[0] % T; k[0] += 1; st[j](acts[j].data_ptr())
        us = timed(f, reps=8 * T, warm=T)
    elif T > 1:
        o = b._alloc_out((T,), True, obs_p2, False, None, bits)
        us = timed(lambda: b.rollout(acts[:T], None if a2 is None else a2[:T], out=o), reps=16, warm=4) / T
    else:
        o = b._alloc_out((), True, obs_p2, False, None, bits)
        st = [b.make_stepper(o, None if a2 is None else a2[j], auto_reset=True) for j in range(64)]
        k = [0]
        def f():
            j = k[0] % 64; k[0] += 1; st[j](acts[j].data_ptr())
        us = timed(f)
    b.check()
    by = bench.algorithmic_bytes_per_env_step(W, H, K, selfplay=obs_p2) * n
    if bits:
        by -= (2 if obs_p2 else 1) * (12 * W * H * 4 - 4 * ((12 * W * H + 31) // 32)) * n
    return {"us": round(us, 2), "GBps": round(by / us / 1e3, 1)}

only = sys.argv[1:]
rows = [("vs_dummy easy 15x10 f32", 15, 10, "dummy", "easy", False, False, 1),
        ("vs_dummy hard 15x10 f32", 15, 10, "dummy", "hard", False, False, 1),
        ("selfplay 15x10 f32 P1+P2", 15, 10, "selfplay", "easy", True, False, 1),
        ("selfplay 15x10 PACKED P1+P2", 15, 10, "selfplay", "easy", True, True, 1),
        ("vs_dummy hard 15x10 PACKED", 15, 10, "dummy", "hard", False, True, 1),
        ("vs_dummy hard 32x32 f32", 32, 32, "dummy", "hard", False, False, 1),
        ("rollout T=32 easy 15x10 f32", 15, 10, "dummy", "easy", False, False, 32),
        ("32 single launches into 32 DISTINCT buffers, easy 15x10 f32", 15, 10, "dummy", "easy", False, False, -32),
        ("rollout T=32 hard 15x10 PACKED", 15, 10, "dummy", "hard", False, True, 32),
        ("32 DISTINCT buffers, hard 15x10 f32", 15, 10, "dummy", "hard", False, False, -32),
        ("16 DISTINCT buffers, selfplay 15x10 f32 P1+P2", 15, 10, "selfplay", "easy", True, False, -16),
        ("4 DISTINCT buffers, hard 32x32 f32", 32, 32, "dummy", "hard", False, False, -4),
        ("rollout T=4 hard 32x32 f32", 32, 32, "dummy", "hard", False, False, 4),
        ("rollout T=8 hard 32x32 PACKED", 32, 32, "dummy", "hard", False, True, 8)]
res = {}
for name, *c in rows:
    if only and not any(o_ in name for o_ in only):
        continue
    kinds = ("wave/plain", "row4/plain") if c[5] else ("wave/plain", "wave/auto", "row4/plain", "row4/auto")   # packed: no f32 stores
    res[name] = {k: cfg(k, *c) for k in kinds}
    torch.cuda.empty_cache()
    print(json.dumps({name: res[name]}), flush=True)
