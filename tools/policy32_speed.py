#!/usr/bin/env python3
"""32 x 32 board (BASELINE config C5) with the policy: the banded fused trunk vs the torch module, B = 16384 (4096 for torch f32)."""
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
def t(fn, reps=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return round(e0.elapsed_time(e1) / reps, 3)
torch.manual_seed(0)
B = 16384
pol = InversusCNNPolicy(12, 32, 32, 4).cuda().eval()
fused = FusedPolicy(pol)
b = DeviceEnvBatch(B, 32, 32, opponent="dummy", difficulty="hard", opp_seed=1)
b.seed(np.arange(B, dtype=np.int64)); b.reset()
for k in range(8): b.step(b.uniform_actions(3, k), None, auto_reset=True, obs=False)
bits, extra = b.render(0, bits=True)
res = {"B": B}
res["banded fused trunk ms (packed obs)"] = t(lambda: fused.features(bits))
res["banded fused trunk TFLOP/s (497.7 MFLOP/sample)"] = round(497.7e6 * B / (res["banded fused trunk ms (packed obs)"] * 1e-3) / 1e12, 1)
res["fused forward (trunk + heads) ms"] = t(lambda: fused(bits, extra))
planes, _ = b.render(0)
Bt = 2048
with torch.no_grad():
    res[f"torch f32 forward ms (B={Bt}, scaled x{B // Bt})"] = round(t(lambda: pol(planes[:Bt], extra[:Bt]), 2, 1) * (B // Bt), 1)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        res[f"torch bf16 autocast forward ms (B={Bt}, scaled x{B // Bt})"] = round(t(lambda: pol(planes[:Bt], extra[:Bt]), 2, 1) * (B // Bt), 1)
print(json.dumps(res, indent=1))
