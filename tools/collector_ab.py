#!/usr/bin/env python3
"""Self-play collector, packed vs f32 observations, alternating (N = 16384, fused actors)."""
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
from inversus_reinforcement_learning_amd.training import RolloutCollector

n = 16384
torch.manual_seed(0)
b = DeviceEnvBatch(n, opponent="selfplay"); b.seed(np.arange(n, dtype=np.int64)); b.reset()
agent = PPOAgent(InversusCNNPolicy(12, 10, 15, 4), fused_act=True)
opp = FusedPolicy(InversusCNNPolicy(12, 10, 15, 4).cuda().eval(), actor_only=len(sys.argv) < 2 or sys.argv[1] != 'full')
res = {"packed": [], "f32": []}
for rep in range(4):
    for packed in (True, False):
        col = RolloutCollector(b, agent, opp, packed_obs=packed)
        col.collect(2); agent.reset_buffers(); torch.cuda.synchronize()
        t0 = time.perf_counter(); col.collect(16); torch.cuda.synchronize()
        res["packed" if packed else "f32"].append(round(16 * n / (time.perf_counter() - t0)))
        agent.reset_buffers()
print(json.dumps(res))
