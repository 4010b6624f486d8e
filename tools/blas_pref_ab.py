#!/usr/bin/env python3
"""rocBLAS vs hipBLASLt (torch.backends.cuda.preferred_blas_library) on the three GEMMs of the heads' first Linear, B = 16384."""
import json, torch
B, NF, H = 16384, 19200, 512
torch.manual_seed(0)
feat = (torch.randn(B, NF, device="cuda") * 0.5).clamp_min(0).to(torch.bfloat16)
wf = (torch.randn(H, NF, device="cuda") * 0.01).to(torch.bfloat16); wft = wf.t().contiguous()
wf1 = wf[:256].contiguous()
dpre = (torch.randn(B, H, device="cuda") * 0.01).to(torch.bfloat16); dpret = dpre.t().contiguous()
def t(fn, reps=30):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)
res = {}
for lib in ("cublaslt", "cublas", "cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    r = res.setdefault(lib, [])
    r.append({"fwd 512": t(lambda: feat @ wf.t()), "fwd 256": t(lambda: feat @ wf1.t()), "dfeat (NT)": t(lambda: dpre @ wft.t()),
              "dwf (NN)": t(lambda: dpret @ feat)})
print(json.dumps(res, indent=1)); print("current:", torch.backends.cuda.preferred_blas_library())
