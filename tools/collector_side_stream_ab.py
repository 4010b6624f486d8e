#!/usr/bin/env python3
"""Self-play collector at N = 16 384 with the opponent's forward on the main stream vs on a side stream (RolloutCollector(
opponent_stream=True)): two collectors on two env batches with the same seeds, alternated in ONE process, ms per vector step.
Both forwards are this library's kernels (no workgroup of them waits on another), so the two-stream form cannot deadlock
(profiles/r04_incidents.md); this script measures whether it pays."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
from inversus_reinforcement_learning_amd.training import RolloutCollector
n, T, rounds = 16384, 8, 6
cols = {}
for name, side in (("one_stream", False), ("two_streams", True)):
    torch.manual_seed(0)
    b = DeviceEnvBatch(n, opponent="selfplay"); b.seed(np.arange(n, dtype=np.int64)); b.reset()
    agent = PPOAgent(InversusCNNPolicy(12, 10, 15, 4), fused_act=True)
    opp = FusedPolicy(InversusCNNPolicy(12, 10, 15, 4).cuda().eval(), actor_only=True)
    cols[name] = (RolloutCollector(b, agent, opp, packed_obs=True, opponent_stream=side), agent)
res = {k: [] for k in cols}
for r in range(rounds + 1):
    for name, (col, agent) in cols.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        col.collect(T)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3 / T
        agent.reset_buffers()
        if r:                                                      # round 0 = warm-up
            res[name].append(round(dt, 4))
out = {k: {"ms_per_vector_step_median": sorted(v)[len(v) // 2], "min": min(v), "all": v} for k, v in res.items()}
out["env_steps_per_s"] = {k: round(n / (v["ms_per_vector_step_median"] * 1e-3)) for k, v in out.items()}
print(json.dumps(out))
