#!/usr/bin/env python3
"""Profiling target for rocprofv3 (round 3): LAUNCHES launches of ONE benched step configuration, chosen by CFG:
  easy | hard | selfplay_f32 | selfplay_packed | hard_packed | c5 (32x32 f32, hard) | c5_packed |
  easy_streaming | hard_streaming (32 distinct output buffers) | rollout (inv_rollout T=32, easy f32; LAUNCHES = rollouts)
Env: KERNEL=auto|wave|row4, STORE=auto|plain."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch, StepResult

cfg = os.environ.get("CFG", "easy")
n = int(os.environ.get("N", 16384))
launches = int(os.environ.get("LAUNCHES", 200))
W, H = (32, 32) if cfg.startswith("c5") else (15, 10)
selfplay = cfg.startswith("selfplay")
diff = "hard" if cfg.startswith(("hard", "c5")) else "easy"
bits = cfg.endswith("packed")
T = 32 if cfg.endswith("streaming") else 1
b = DeviceEnvBatch(n, W, H, 16, opponent="selfplay" if selfplay else "dummy", difficulty=diff, opp_seed=1,
                   step_kernel=os.environ.get("KERNEL", "auto"), obs_store=os.environ.get("STORE", "auto"))
b.seed(np.arange(n, dtype=np.int64)); b.reset()
acts = torch.stack([b.uniform_actions(5, t) for t in range(64)])
a2 = torch.stack([b.uniform_actions(6, t) for t in range(64)]) if selfplay else None
if cfg == "rollout":
    o = b._alloc_out((32,), True, False, False, None)
    for _ in range(launches):
        b.rollout(acts[:32], out=o)
else:
    o = b._alloc_out((T,), True, selfplay, False, None, bits)
    outs = [StepResult(**{k: (getattr(o, k)[j] if getattr(o, k) is not None else None) for k in StepResult.__slots__}) for j in range(T)]
    st = [b.make_stepper(outs[j % T], None if a2 is None else a2[j], auto_reset=True) for j in range(64)]
    for t in range(launches):
        st[t % 64](acts[t % 64].data_ptr())
torch.cuda.synchronize()
b.check()
