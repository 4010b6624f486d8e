#!/usr/bin/env python3
"""Per-tensor relative L2 error of the fused-trunk gradients (and of torch bf16 autocast, for scale) vs f32 autograd."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_fused_train import _policy, _obs, rel
from inversus_reinforcement_learning_amd.fused_train import FusedTrainPolicy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pol = _policy(1)
gf, gb, x = _obs(n, 7)
torch.manual_seed(2)
r1, r2 = torch.randn(n, 13, device="cuda"), torch.randn(n, 1, device="cuda")
def grads(fwd):
    pol.zero_grad()
    lo, v = fwd()
    ((lo.float() * r1).sum() + (v.float() * r2).sum()).backward()
    return {k: p.grad.clone() for k, p in pol.named_parameters()}
want = grads(lambda: pol(gf, x))
def ac():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return pol(gf, x)
res = {"fused": {k: round(rel(v, want[k]), 4) for k, v in grads(lambda: FusedTrainPolicy(pol)(gf, x)).items()},
       "torch_bf16_autocast": {k: round(rel(v, want[k]), 4) for k, v in grads(ac).items()}}
print(json.dumps(res, indent=1))
