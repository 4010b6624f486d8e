#!/usr/bin/env python3
"""Forward / forward+backward time of the CNN policy at rollout batch sizes (measurement helper)."""
import os, sys, json, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy

def t(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

res = {}
pol = InversusCNNPolicy(12, 10, 15, 4).cuda().eval()
for B in (2048, 16384):
    g = (torch.rand(B, 12, 10, 15, device="cuda") < 0.2).float(); x = torch.rand(B, 4, device="cuda")
    with torch.no_grad():
        res[f"fwd f32 B={B} ms"] = t(lambda: pol(g, x))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            res[f"fwd bf16 autocast B={B} ms"] = t(lambda: pol(g, x))
        gcl = g.contiguous(memory_format=torch.channels_last)
        polcl = InversusCNNPolicy(12, 10, 15, 4).cuda().eval().to(memory_format=torch.channels_last)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            res[f"fwd bf16 channels_last B={B} ms"] = t(lambda: polcl(gcl, x))
    flops = 92.6e6 * B
    for k in list(res):
        if f"B={B} ms" in k:
            res[k.replace(" ms", " TFLOP/s")] = flops / (res[k] * 1e-3) / 1e12
pol.train()
B = 2048
g = (torch.rand(B, 12, 10, 15, device="cuda") < 0.2).float(); x = torch.rand(B, 4, device="cuda")
def fb(dtype):
    def f():
        with torch.autocast("cuda", dtype=dtype, enabled=dtype is not None):
            lo, v = pol(g, x)
            loss = lo.float().square().mean() + v.float().square().mean()
        loss.backward()
    return f
res["fwd+bwd f32 B=2048 ms"] = t(fb(None))
res["fwd+bwd bf16 B=2048 ms"] = t(fb(torch.bfloat16))
print(json.dumps(res, indent=1))

# fused trunk (csrc/inv_policy.hip)
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
pol.eval()
fused = FusedPolicy(pol)
res2 = {}
for B in (2048, 16384):
    g = (torch.rand(B, 12, 10, 15, device="cuda") < 0.2).float(); x = torch.rand(B, 4, device="cuda")
    res2[f"fused trunk B={B} ms"] = t(lambda: fused.features(g))
    res2[f"fused trunk+heads B={B} ms"] = t(lambda: fused(g, x))
    res2[f"fused trunk B={B} TFLOP/s (72.9 MFLOP/sample)"] = 72.9e6 * B / (res2[f"fused trunk B={B} ms"] * 1e-3) / 1e12
print(json.dumps(res2, indent=1))
