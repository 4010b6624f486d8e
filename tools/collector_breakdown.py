#!/usr/bin/env python3
"""Where a self-play collector step spends its time (fused actor), N = 16384."""
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy

def t(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

n = 16384
b = DeviceEnvBatch(n, opponent="selfplay"); b.seed(np.arange(n, dtype=np.int64)); b.reset()
pol = InversusCNNPolicy(12, 10, 15, 4).cuda().eval(); f = FusedPolicy(pol)
g, x = b.render(0)
res = {}
res["trunk features ms"] = t(lambda: f.features(g))
feat = f.features(g)
res["heads first GEMM [16384x19200]x[19200x512] bf16 ms"] = t(lambda: feat @ f.heads.wf.t())
res["fused forward (trunk+heads) ms"] = t(lambda: f(g, x))
logits, v = f(g, x)
def sample():
    d = torch.distributions.Categorical(logits=logits); a = d.sample(); return a, d.log_prob(a)
res["Categorical sample + log_prob ms"] = t(sample)
def sample_fast():
    lp = torch.log_softmax(logits, -1); a = torch.multinomial(lp.exp(), 1).squeeze(1); return a, lp.gather(1, a[:, None]).squeeze(1)
res["log_softmax + multinomial ms"] = t(sample_fast)
res["alloc obs out buffers (2 x 118 MB) ms"] = t(lambda: b._alloc_out((), True, True, False, None))
out = b._alloc_out((), True, True, False, None)
a1 = b.uniform_actions(1, 0); a2 = b.uniform_actions(2, 0)
res["env step (P1+P2 obs) via step() ms"] = t(lambda: b.step(a1, a2, auto_reset=True, out=out))
res["done.any() sync ms"] = t(lambda: bool(out.done.bool().any()))
print(json.dumps(res, indent=1))
