#!/usr/bin/env python3
"""First Linear of both heads at rollout batch size: [16384 x 19200] bf16 features times the [512 x 19200] weight, as
feat @ W.t() (weight row-major, what FusedPolicy does) vs feat @ Wt (pre-transposed copy), bf16 and f32 outputs."""
import json, torch
B, K, N = 16384, 19200, 512
feat = torch.randn(B, K, device="cuda").to(torch.bfloat16)
w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
wt = w.t().contiguous()
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)
res = {"feat @ W.t() ms": t(lambda: feat @ w.t()), "feat @ Wt ms": t(lambda: feat @ wt),
       "F.linear ms": t(lambda: torch.nn.functional.linear(feat, w))}
res["TFLOP/s best"] = round(2 * B * K * N / (min(v for v in res.values()) * 1e-3) / 1e12, 1)
print(json.dumps(res))
