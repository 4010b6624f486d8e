#!/usr/bin/env python3
"""RolloutCollector(graph=True) at full size (self-play, 16 384 envs, T = 128, packed observations, fused actors): seconds per rollout eager
vs captured + replayed, peak device memory, and the same for the reference-sized quick start (4 envs, T = 512).  One JSON line per case."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
from inversus_reinforcement_learning_amd.training import RolloutCollector

for n, T, mode in ((16384, 128, "selfplay"), (4, 512, "dummy")):
    out = {"envs": n, "T": T, "mode": mode}
    for graph in (False, True):
        torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
        torch.manual_seed(0)
        b = DeviceEnvBatch(n, opponent=mode if mode == "selfplay" else "dummy")
        b.seed(np.arange(n, dtype=np.int64)); b.reset()
        agent = PPOAgent(InversusCNNPolicy(12, 10, 15, 4), fused_act=True)
        opp = FusedPolicy(InversusCNNPolicy(12, 10, 15, 4).cuda().eval(), actor_only=True) if mode == "selfplay" else None
        col = RolloutCollector(b, agent, opp, graph=graph)
        ts = []
        for r in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            col.collect(T); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            agent.reset_buffers()
        key = "graph" if graph else "eager"
        out[key + "_s_per_rollout"] = [round(x, 4) for x in ts]
        out[key + "_env_steps_per_s_last"] = round(n * T / ts[-1])
        out[key + "_peak_GB"] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
        del col, agent, b, opp
    print(json.dumps(out), flush=True)
