#!/usr/bin/env python3
"""Trunk forward time: inference epilogue / train numerics with saved == NULL / training forward (SAVE), N = 16384 packed obs."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
from inversus_reinforcement_learning_amd.fused_train import FusedTrainPolicy

n = 16384
torch.manual_seed(0)
b = DeviceEnvBatch(n, opponent="selfplay"); b.seed(np.arange(n, dtype=np.int64)); b.reset()
rng = np.random.RandomState(0)
for _ in range(8):
    b.step(rng.randint(0, 13, n).astype(np.uint8), rng.randint(0, 13, n).astype(np.uint8), auto_reset=True, obs=False)
gb, x = b.render(0, bits=True)
pol = InversusCNNPolicy(12, 10, 15, 4).cuda()
fi, ft, fs = FusedPolicy(pol), FusedPolicy(pol, train_numerics=True), FusedTrainPolicy(pol)

def t(fn, reps=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)

out = {}
for rep in range(3):
    out.setdefault("inference", []).append(t(lambda: fi.features(gb)))
    out.setdefault("train_numerics_nosave", []).append(t(lambda: ft.features(gb)))
    with torch.no_grad():
        out.setdefault("train_fwd_save", []).append(t(lambda: fs.features(gb)))
print(json.dumps(out))
