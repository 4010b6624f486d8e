#!/usr/bin/env python3
"""Formulations of the three big GEMMs of the heads' first Linear at B = 16384 (bf16, f32 accumulation): ms each."""
import json, torch
B, NF, H = 16384, 19200, 512
torch.manual_seed(0)
feat = (torch.randn(B, NF, device="cuda") * 0.5).clamp_min(0).to(torch.bfloat16)
wf = (torch.randn(H, NF, device="cuda") * 0.01).to(torch.bfloat16)          # [512, 19200]
wft = wf.t().contiguous()                                                     # [19200, 512]
dpre = (torch.randn(B, H, device="cuda") * 0.01).to(torch.bfloat16)
dpret = dpre.t().contiguous()                                                 # [512, B]
featt = None
def t(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)
res = {}
res["fwd  feat @ wf.t()            (NT)"] = t(lambda: feat @ wf.t())
res["fwd  feat @ wft               (NN)"] = t(lambda: feat @ wft)
res["fwd  (wf @ feat.t()).t()          "] = t(lambda: (wf @ feat.t()).t())
res["dfeat dpre @ wf               (NN)"] = t(lambda: dpre @ wf)
res["dfeat dpre @ wft.t()          (NT)"] = t(lambda: dpre @ wft.t())
res["dfeat (wft @ dpre.t()).t()        "] = t(lambda: (wft @ dpre.t()).t())
res["dfeat (wft @ dpret).t()           "] = t(lambda: (wft @ dpret).t())
res["dwf  dpre.t() @ feat          (TN)"] = t(lambda: dpre.t() @ feat)
res["dwf  dpret @ feat             (NN)"] = t(lambda: dpret @ feat)
res["dwf  (feat.t() @ dpre).t()    (TN)"] = t(lambda: (feat.t() @ dpre).t())
G = 4
res["dwf  split-K 4 bmm + sum          "] = t(lambda: torch.bmm(dpre.view(G, B // G, H).transpose(1, 2), feat.view(G, B // G, NF)).float().sum(0))
out = torch.empty(H, NF, dtype=torch.bfloat16, device="cuda")
res["dwf  mm(out=...) dpre.t() @ feat  "] = t(lambda: torch.mm(dpre.t(), feat, out=out))
print(json.dumps(res, indent=1))
