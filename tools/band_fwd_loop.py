#!/usr/bin/env python3
"""A few launches of the banded 32 x 32 trunk forward at B = 16384 (packed observations) -- the program tools/band_pmc.sh profiles."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
f = FusedPolicy(InversusCNNPolicy(12, 32, 32, 4).cuda().eval())
g = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 384), dtype=torch.int32, device="cuda") & torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 384), dtype=torch.int32, device="cuda")
for _ in range(n):
    f.features(g)
torch.cuda.synchronize()
