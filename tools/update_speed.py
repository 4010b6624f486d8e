#!/usr/bin/env python3
"""One PPO minibatch (forward + loss + backward + clip + Adam) at B samples: torch f32, torch bf16 autocast, fused trunk."""
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_train import FusedTrainPolicy, _TrunkTrainFn, _trunk_params
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
torch.manual_seed(0)
pol = InversusCNNPolicy(12, 10, 15, 4).cuda().train()
opt = torch.optim.Adam(pol.parameters(), lr=1e-4)
g = (torch.rand(B, 12, 10, 15, device="cuda") < 0.2).float(); x = torch.rand(B, 4, device="cuda")
gbits = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 57), dtype=torch.int32, device="cuda") & torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 57), dtype=torch.int32, device="cuda")
gbits[:, 56] &= 0xFF
a = torch.randint(0, 13, (B,), device="cuda"); adv = torch.randn(B, device="cuda"); ret = torch.randn(B, device="cuda"); olp = torch.full((B,), -2.56, device="cuda")
ft = FusedTrainPolicy(pol)
def step(fwd):
    def f():
        lo, v = fwd()
        d = torch.distributions.Categorical(logits=lo.float(), validate_args=False); lp = d.log_prob(a); r = torch.exp(lp - olp)
        loss = -torch.min(r * adv, torch.clamp(r, 0.8, 1.2) * adv).mean() + 0.1 * torch.nn.functional.mse_loss(v.float().squeeze(-1), ret) - 0.02 * d.entropy().mean()
        opt.zero_grad(); loss.backward(); torch.nn.utils.clip_grad_norm_(pol.parameters(), 0.5); opt.step()
    return f
def ac():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return pol(g, x)
def t(fn, reps=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / reps * 1e3, 2)
res = {"B": B}
class _P(dict):
    def __setitem__(self, k, v):
        super().__setitem__(k, v); print(json.dumps({k: v}), flush=True)
res = _P(res)
with torch.no_grad():
    res["  train-forward kernel ms"] = t(lambda: ft.features(gbits), 5)
feat = ft.features(gbits); df = torch.randn_like(feat)
res["  trunk fwd + bwd (kernels + parameter-gradient reductions) ms"] = t(lambda: torch.autograd.grad(ft.features(gbits), _trunk_params(pol), df), 3)
res["fused trunk (packed obs) ms"] = t(step(lambda: ft(gbits, x)))
res["fused trunk (f32 obs) ms"] = t(step(lambda: ft(g, x)))
if "--no-torch" not in sys.argv:
    res["torch bf16 autocast ms"] = t(step(ac), 2)
    res["torch f32 ms"] = t(step(lambda: pol(g, x)), 2)
