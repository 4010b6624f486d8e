#!/usr/bin/env python3
"""Banded 32 x 32 trunk at B = 16384: ms per forward for a given number of resident workgroups (probe build -DPOL_GRID_PROBE, env
POL_BLOCKS_X2 = workgroups per two CUs).  Usage: INVERSUS_HIP_LIB=tools/exp/lib_grid.so POL_BLOCKS_X2=2 python tools/banded_grid_probe.py"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
torch.manual_seed(0)
B = 16384
pol = InversusCNNPolicy(12, 32, 32, 4).cuda().eval()
fused = FusedPolicy(pol)
b = DeviceEnvBatch(B, 32, 32, opponent="dummy", difficulty="hard", opp_seed=1)
b.seed(np.arange(B, dtype=np.int64)); b.reset()
for k in range(8): b.step(b.uniform_actions(3, k), None, auto_reset=True, obs=False)
bits, extra = b.render(0, bits=True)
for _ in range(2): f = fused.features(bits)
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): f = fused.features(bits)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4)
print(json.dumps({"POL_BLOCKS_X2": os.environ.get("POL_BLOCKS_X2", "4"), "ms": [round(x, 3) for x in ts], "checksum": float(f.float().abs().sum())}))
