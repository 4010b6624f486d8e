#!/usr/bin/env python3
"""Forward GEMM of the heads' first Linear, [B x 19200] x [19200 x H] bf16, H = 512 (both heads) and 256 (actor only): formulations."""
import json, torch
B, NF = 16384, 19200
torch.manual_seed(0)
feat = (torch.randn(B, NF, device="cuda") * 0.5).clamp_min(0).to(torch.bfloat16)
def t(fn, reps=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)
res = {}
for H in (512, 256):
    wf = (torch.randn(H, NF, device="cuda") * 0.01).to(torch.bfloat16); wft = wf.t().contiguous()
    out = torch.empty(B, H, dtype=torch.bfloat16, device="cuda")
    r = {}
    r["feat @ wf.t()"] = t(lambda: feat @ wf.t())
    r["feat @ wft"] = t(lambda: feat @ wft)
    r["(wf @ feat.t()).t().contiguous()"] = t(lambda: (wf @ feat.t()).t().contiguous())
    r["mm(out=)"] = t(lambda: torch.mm(feat, wf.t(), out=out))
    # (batched views with interleaved or zero batch strides -- split-K / row-split through torch.bmm -- faulted inside the GEMM
    # library on this stack and are not measured)
    res[f"H={H}"] = r
print(json.dumps(res, indent=1))
