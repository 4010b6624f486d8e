#!/usr/bin/env python3
"""inv_policy_param_grads_all alone (the four LayerNorm gradient reductions of a 16 384-sample minibatch: 3.46 GB of dy / x^ read once):
us per launch and TB/s against the number of sample slices (each slice ends in 16 f32 atomics per thread)."""
import os, sys, json, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd import _lib as L
lib = L.load()
n, P, CO = 16384, 150, (32, 64, 128, 128)
dev = "cuda"
gr, sv, lng = L.InvPolicyGrads(), L.InvPolicySaved(), L.InvPolicyLnGrads()
keep = []
for i, c in enumerate(CO):
    dy = (torch.randn(n, P, c, device=dev) * 0.1).bfloat16(); xh = torch.randn(n, P, c, device=dev).bfloat16()
    gw = torch.zeros(c, 10, 15, device=dev); gb = torch.zeros(c, 10, 15, device=dev)
    keep += [dy, xh, gw, gb]
    gr.dy[i], sv.xhat[i], lng.gw[i], lng.gb[i] = dy.data_ptr(), xh.data_ptr(), gw.data_ptr(), gb.data_ptr()
nbytes = sum(2 * 2 * n * P * c for c in CO)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
out = {}
for sl in [int(x) for x in sys.argv[1:]] or [16, 32, 64, 96, 128, 256]:
    for _ in range(3):
        L.raise_for(lib.inv_policy_param_grads_all(C.byref(gr), C.byref(sv), C.byref(lng), sl, n, st), lib)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.inv_policy_param_grads_all(C.byref(gr), C.byref(sv), C.byref(lng), sl, n, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    out[f"slices={sl}"] = {"us": round(us, 1), "TBps": round(nbytes / us / 1e6, 2)}
print(json.dumps(out))
