#!/usr/bin/env python3
"""ms per banded 32 x 32 trunk forward at B = 16384 (packed observations), the library INVERSUS_HIP_LIB names (default: the shipped one)."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
torch.manual_seed(0)
B = 16384
f = FusedPolicy(InversusCNNPolicy(12, 32, 32, 4).cuda().eval())
g = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 384), dtype=torch.int32, device="cuda") & torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 384), dtype=torch.int32, device="cuda")
for _ in range(2): f.features(g)
torch.cuda.synchronize()
ts = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f.features(g)
    e1.record(); torch.cuda.synchronize()
    ts.append(round(e0.elapsed_time(e1) / 5, 3))
print("banded forward ms", ts, (sys.argv[1] if len(sys.argv) > 1 else ""), flush=True)
