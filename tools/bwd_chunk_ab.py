#!/usr/bin/env python3
"""Experiment (round 3, measurement only): would running the trunk's backward kernel and the LayerNorm reduction CHUNK BY CHUNK,
with dy (and dz) in buffers that are reused by every chunk and therefore stay in the 256 MiB Infinity Cache, beat the monolithic
launches that stream 1.7 GB of dy (and dz) through HBM?  Same kernels through the C ABI, pointers offset per chunk.
    monolithic : inv_policy_trunk_bwd(N) + inv_policy_param_grads_all(N)
    chunked    : for each chunk of M samples: trunk_bwd(chunk) + param_grads_all(chunk), dy reused; dz full-size or reused too
ms per 16 384-sample minibatch (median of 5)."""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd import _lib as L, fused_train as FT
from inversus_reinforcement_learning_amd.fused_policy import pack_conv, pack_ln
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
lib = L.load()
n = 16384
torch.manual_seed(0)
pol = InversusCNNPolicy(12, 10, 15, 4).cuda()
bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, 57), dtype=torch.int32, device="cuda") & torch.randint(-2 ** 31, 2 ** 31 - 1, (n, 57), dtype=torch.int32, device="cuda")
ctx = FT._DirectCtx()
feat = FT._TrunkTrainFn.forward(ctx, bits, *FT._trunk_params(pol))
sv = ctx.saved_tensors
obs, rstd = sv[0], sv[1]
xhat, act, lns, ws = sv[2:6], sv[6:9], sv[9:13], sv[13:17]
dfeat = (torch.randn(n, 19200, device="cuda") * 1e-3).to(torch.bfloat16)
keep = [pack_conv(ws[i], FT._COUT[i], transposed=True) for i in range(1, 4)]
bp = L.InvPolicyBwdParams()
for i in range(3):
    bp.wt[i] = keep[i].data_ptr()
for i in range(4):
    bp.ln[i] = lns[i].data_ptr()
gw = [torch.zeros(c, 10, 15, device="cuda") for c in FT._COUT]
gb = [torch.zeros(c, 10, 15, device="cuda") for c in FT._COUT]
lng = L.InvPolicyLnGrads()
for i in range(4):
    lng.gw[i], lng.gb[i] = gw[i].data_ptr(), gb[i].data_ptr()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(M, reuse_dy, reuse_dz):
    ndy = M if reuse_dy else n
    ndz = M if reuse_dz else n
    dy = [torch.empty((ndy, 150, c), dtype=torch.bfloat16, device="cuda") for c in FT._COUT]
    dz = [torch.empty((ndz, 150, c), dtype=torch.bfloat16, device="cuda") for c in FT._COUT]

    def once():
        for lo in range(0, n, M):
            saved, grads = L.InvPolicySaved(), L.InvPolicyGrads()
            for i in range(4):
                c = FT._COUT[i]
                saved.xhat[i] = xhat[i].data_ptr() + lo * 150 * c * 2
                grads.dy[i] = dy[i].data_ptr() + (0 if reuse_dy else lo * 150 * c * 2)
                grads.dz[i] = dz[i].data_ptr() + (0 if reuse_dz else lo * 150 * c * 2)
            for i in range(3):
                saved.act[i] = act[i].data_ptr() + lo * 150 * FT._COUT[i] * 2
            saved.rstd = rstd.data_ptr() + lo * 16
            L.raise_for(lib.inv_policy_trunk_bwd(C.byref(bp), C.byref(saved), C.c_void_p(dfeat.data_ptr() + lo * 19200 * 2), C.byref(grads), M, st), lib)
            L.raise_for(lib.inv_policy_param_grads_all(C.byref(grads), C.byref(saved), C.byref(lng), 64 if M >= 4096 else 16, M, st), lib)
    once(); torch.cuda.synchronize()
    out = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); once(); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return round(sorted(out)[2], 3)


res = {"monolithic (bwd + LN reduction, N = 16384)": run(n, False, False)}
for M in (4096, 2048, 1024, 512):
    res[f"chunks of {M}: dy reused, dz full-size"] = run(M, True, False)
    res[f"chunks of {M}: dy and dz reused"] = run(M, True, True)
print(json.dumps(res, indent=1))
