#!/usr/bin/env python3
"""Is the fused trunk's time data-dependent (DVFS under MFMA load) or a short-run artefact?  N = 16384."""
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
n = 16384
torch.manual_seed(0)
pol = InversusCNNPolicy(12, 10, 15, 4).cuda().eval()
f = FusedPolicy(pol)
b = DeviceEnvBatch(n, opponent="selfplay"); b.seed(np.arange(n, dtype=np.int64)); b.reset()
rng = np.random.RandomState(0)
for _ in range(8):
    b.step(rng.randint(0, 13, n).astype(np.uint8), rng.randint(0, 13, n).astype(np.uint8), auto_reset=True, obs=False)
real_bits, _ = b.render(0, bits=True)
real_f32, _ = b.render(0)
rand_f32 = (torch.rand(n, 12, 10, 15, device="cuda") < 0.2).float()
rand_bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, 57), dtype=torch.int32, device="cuda") & torch.randint(-2 ** 31, 2 ** 31 - 1, (n, 57), dtype=torch.int32, device="cuda")
def t(x, reps, warm=3):
    for _ in range(warm): f.features(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f.features(x)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)
out = {}
for name, x in (("real packed", real_bits), ("real f32", real_f32), ("random 20% f32", rand_f32), ("random 25% packed", rand_bits)):
    time.sleep(0.5)
    out[name] = {"5 reps": t(x, 5), "40 reps": t(x, 40), "200 reps": t(x, 200)}
print(json.dumps(out, indent=1))
print("ones fraction: real", float(real_f32.mean()), "random", float(rand_f32.mean()))
