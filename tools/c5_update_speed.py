#!/usr/bin/env python3
"""BASELINE config C5 (32 x 32 board) with the policy in the loop: self-play collection and the PPO update through the fused
(banded) path at 16 384 envs, ms per 16 384-sample minibatch; beside it the same minibatch through torch autograd over the module
(f32 like the reference, and bf16 autocast), measured on 2 048 samples and scaled.   python tools/c5_update_speed.py [--no-torch]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch, unpack_obs
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
from inversus_reinforcement_learning_amd.training import RolloutCollector
n, T, H, W = int(os.environ.get("N", 16384)), int(os.environ.get("T", 2)), 32, 32
torch.manual_seed(0)
b = DeviceEnvBatch(n, W, H, opponent="selfplay"); b.seed(np.arange(n, dtype=np.int64)); b.reset()
agent = PPOAgent(InversusCNNPolicy(12, H, W, 4), fused_act=True, fused_update=True, batch_size=n, epochs=1)
col = RolloutCollector(b, agent, FusedPolicy(InversusCNNPolicy(12, H, W, 4).cuda().eval(), actor_only=True))
res = {"envs": n, "rollout_steps": T}
col.collect(T); agent.update(); torch.cuda.synchronize()
cs, us = [], []
for _ in range(3):                                                     # median of 3 iterations (the first collect after an update can
    t0 = time.perf_counter(); col.collect(T); torch.cuda.synchronize(); t1 = time.perf_counter()      # pay for allocator churn)
    agent.update(); torch.cuda.synchronize(); t2 = time.perf_counter()
    cs.append(t1 - t0); us.append(t2 - t1)
c, u = sorted(cs)[1], sorted(us)[1]
res["collect_ms_per_vector_step"] = round(c * 1e3 / T, 2)
res["selfplay_collector_env_steps_per_s"] = round(n * T / c)
res["fused_update_ms_per_minibatch"] = round(u * 1e3 / T, 2)
res["train_env_steps_per_s_1_epoch"] = round(n * T / (c + u))
res["peak_memory_GB"] = round(torch.cuda.max_memory_allocated() / 1e9, 1)
print(json.dumps(res), flush=True)
if "--no-torch" not in sys.argv:
    m = 2048
    col.collect(1)
    grid = unpack_obs(agent.obs_grid_buffer[0][:m], H, W); extra = agent.obs_extra_buffer[0][:m]
    agent.reset_buffers()
    pol = agent.policy
    r1, r2 = torch.randn(m, 13, device="cuda"), torch.randn(m, 1, device="cuda")
    for name, ctx in (("torch_f32_autograd", None), ("torch_bf16_autocast_autograd", torch.bfloat16)):
        def step():
            for p in pol.parameters():
                p.grad = None
            with torch.autocast("cuda", dtype=ctx or torch.bfloat16, enabled=ctx is not None):
                lo, v = pol(grid, extra)
            ((lo.float() * r1).sum() + (v.float() * r2).sum()).backward()
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter(); step(); step(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / 2
        res[name + "_ms_per_16384_samples (measured on 2048, x8)"] = round(ms * 8, 1)
    print(json.dumps(res), flush=True)
