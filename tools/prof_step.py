#!/usr/bin/env python3
"""Profiling target: a fixed number of k_step launches in one configuration (for rocprofv3 --pmc)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch

n = int(os.environ.get("N", 16384)); W = int(os.environ.get("W", 15)); H = int(os.environ.get("H", 10))
mode = os.environ.get("MODE", "full")          # full | noobs | a2 | both
diff = os.environ.get("DIFF", "easy")
launches = int(os.environ.get("LAUNCHES", 40))
b = DeviceEnvBatch(n, W, H, opponent="dummy", difficulty=diff, opp_seed=1)
b.seed(np.arange(n, dtype=np.int64)); b.reset()
out = b._alloc_out((), mode != "noobs", mode == "both", False, None)
a1 = b.uniform_actions(1234, 0)
a2 = torch.zeros_like(a1) if mode in ("a2", "both") else None
for t in range(launches):
    b.uniform_actions(1234, t, out=a1)
    b.step(a1, a2, auto_reset=True, out=out)
torch.cuda.synchronize()
b.check()
