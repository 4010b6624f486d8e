#!/usr/bin/env python3
"""Upper bound of what any change to the per-minibatch RE-PACKING of the parameters could buy the fused update (timing only: the
packed operands are cached after their first build, so minibatches 2.. run on stale weights -- wrong numerics on purpose).
ms per 16 384-sample minibatch, shipped vs cached packing, alternated in one process."""
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
from inversus_reinforcement_learning_amd.training import RolloutCollector
from inversus_reinforcement_learning_amd import fused_train as FT, heads as HD
n, T = 16384, 4
torch.manual_seed(0)
b = DeviceEnvBatch(n, opponent="selfplay"); b.seed(np.arange(n, dtype=np.int64)); b.reset()
agent = PPOAgent(InversusCNNPolicy(12, 10, 15, 4), fused_act=True, fused_update=True, batch_size=n, epochs=1)
col = RolloutCollector(b, agent, FusedPolicy(InversusCNNPolicy(12, 10, 15, 4).cuda().eval()))
def once():
    col.collect(T); torch.cuda.synchronize(); t0 = time.perf_counter(); agent.update(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / T
orig = dict(pc=FT.pack_conv, pl=FT.pack_ln, fc1=HD._pack_fc1, fr=HD._pack_fragments)
cache = {}
def cached(name, fn):
    def f(*a, **k):
        key = (name, tuple(getattr(x, "shape", x) if not isinstance(x, (list, tuple)) else len(x) for x in a), tuple(sorted((kk, str(getattr(v, "shape", v))) for kk, v in k.items() if kk != "out")))
        if key not in cache:
            cache[key] = fn(*a, **k)
        return cache[key]
    return f
def set_mode(stale):
    FT.pack_conv = cached("pc", orig["pc"]) if stale else orig["pc"]
    FT.pack_ln = cached("pl", orig["pl"]) if stale else orig["pl"]
    HD._pack_fc1 = cached("fc1", orig["fc1"]) if stale else orig["fc1"]
    HD._pack_fragments = cached("fr", orig["fr"]) if stale else orig["fr"]
once()
res = {"shipped": [], "cached packing (stale weights)": []}
for rep in range(3):
    for stale, name in ((False, "shipped"), (True, "cached packing (stale weights)")):
        set_mode(stale)
        once()
        res[name].append(round(sorted(once() for _ in range(5))[2], 3))
print(json.dumps(res))
