#!/usr/bin/env python3
"""torch.profiler view of one fused PPO minibatch (B = 16384): top GPU kernels by total time."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_train import FusedTrainPolicy
B = 16384
torch.manual_seed(0)
pol = InversusCNNPolicy(12, 10, 15, 4).cuda().train()
opt = torch.optim.Adam(pol.parameters(), lr=1e-4)
gbits = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 57), dtype=torch.int32, device="cuda") & torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 57), dtype=torch.int32, device="cuda")
gbits[:, 56] &= 0xFF
x = torch.rand(B, 4, device="cuda")
a = torch.randint(0, 13, (B,), device="cuda"); adv = torch.randn(B, device="cuda"); ret = torch.randn(B, device="cuda"); olp = torch.full((B,), -2.56, device="cuda")
ft = FusedTrainPolicy(pol)
def step():
    lo, v = ft(gbits, x)
    d = torch.distributions.Categorical(logits=lo.float()); lp = d.log_prob(a); r = torch.exp(lp - olp)
    loss = -torch.min(r * adv, torch.clamp(r, 0.8, 1.2) * adv).mean() + 0.1 * torch.nn.functional.mse_loss(v.float().squeeze(-1), ret) - 0.02 * d.entropy().mean()
    opt.zero_grad(); loss.backward(); torch.nn.utils.clip_grad_norm_(pol.parameters(), 0.5); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=70))
