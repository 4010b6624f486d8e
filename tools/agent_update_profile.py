#!/usr/bin/env python3
"""PPOAgent(fused_act, fused_update).update() at B = 16384 per minibatch: ms per minibatch, kernel launches per minibatch and the
top GPU kernels (torch.profiler).  Usage: python tools/agent_update_profile.py [--no-prof] [--autograd]"""
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
from inversus_reinforcement_learning_amd.training import RolloutCollector
n, T = 16384, 4
torch.manual_seed(0)
b = DeviceEnvBatch(n, opponent="selfplay"); b.seed(np.arange(n, dtype=np.int64)); b.reset()
agent = PPOAgent(InversusCNNPolicy(12, 10, 15, 4), fused_act=True, fused_update=True, batch_size=n, epochs=1)
if "--autograd" in sys.argv:
    agent.direct_grads = False          # loss.backward() over the same kernels instead of FusedTrainPolicy.loss_and_grads
col = RolloutCollector(b, agent, FusedPolicy(InversusCNNPolicy(12, 10, 15, 4).cuda().eval()))
def once():
    col.collect(T); torch.cuda.synchronize(); t0 = time.perf_counter(); agent.update(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / T
once()
ms = sorted(once() for _ in range(5))
print(json.dumps({"update ms per 16384-sample minibatch (median of 5 updates x 4 minibatches)": round(ms[2], 3), "min": round(ms[0], 3)}))
if "--no-prof" not in sys.argv:
    from torch.profiler import profile, ProfilerActivity
    col.collect(T); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        agent.update(); torch.cuda.synchronize()
    ka = prof.key_averages()
    launches = sum(k.count for k in ka)
    total = sum(k.device_time_total for k in ka) if hasattr(ka[0], "device_time_total") else sum(k.cuda_time_total for k in ka)
    print(json.dumps({"kernel launches per minibatch": round(launches / T, 1), "GPU busy ms per minibatch": round(total / 1e3 / T, 3)}))
    print(ka.table(sort_by="cuda_time_total", row_limit=36, max_name_column_width=64))
