#!/usr/bin/env python3
"""torch-profiler view of a self-play collector (packed observations, fused actors), N = 16384: GPU time per vector step by kernel."""
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
from inversus_reinforcement_learning_amd.training import RolloutCollector
from torch.profiler import profile, ProfilerActivity
n, T = 16384, 8
torch.manual_seed(0)
b = DeviceEnvBatch(n, opponent="selfplay"); b.seed(np.arange(n, dtype=np.int64)); b.reset()
agent = PPOAgent(InversusCNNPolicy(12, 10, 15, 4), fused_act=True)
opp = FusedPolicy(InversusCNNPolicy(12, 10, 15, 4).cuda().eval(), actor_only=True)
col = RolloutCollector(b, agent, opp, packed_obs=True)
for _ in range(2):
    col.collect(T); agent.reset_buffers()
torch.cuda.synchronize(); t0 = time.perf_counter(); col.collect(T); torch.cuda.synchronize()
print(json.dumps({"wall ms per vector step": round((time.perf_counter() - t0) * 1e3 / T, 3)}))
agent.reset_buffers()
if "--no-prof" in sys.argv:                       # under rocprofv3: its kernel trace is the profile
    sys.exit(0)
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    col.collect(T); torch.cuda.synchronize()
ka = [k for k in prof.key_averages() if k.device_time_total > 0 and k.device_type.name != "CPU"]
tot = sum(k.device_time_total for k in ka)
print(json.dumps({"GPU busy ms per vector step": round(tot / 1e3 / T, 3), "launches per vector step": round(sum(k.count for k in ka) / T, 1)}))
for k in sorted(ka, key=lambda k: -k.device_time_total)[:28]:
    print(f"{k.device_time_total / T:9.1f} us/step  x{k.count / T:5.1f}  {k.key[:110]}")
