#!/usr/bin/env python3
"""Profiling target: the fused policy trunk at B = 16384 (for rocprofv3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
torch.manual_seed(0)
B = int(os.environ.get("B", 16384))
pol = InversusCNNPolicy(12, 10, 15, 4).cuda().eval()
fused = FusedPolicy(pol)
g = (torch.rand(B, 12, 10, 15, device="cuda") < 0.2).float()
# REPS launches back to back: the first few run while the clock ramps (1.3-1.5 ms), the steady state is what a collector sees
for _ in range(int(os.environ.get("REPS", 60))):
    fused.features(g)
torch.cuda.synchronize()
