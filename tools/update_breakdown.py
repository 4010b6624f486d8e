#!/usr/bin/env python3
"""Where the fused trunk forward + backward spends its time at B samples (pieces timed separately)."""
import os, sys, json, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd import _lib as L
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd import fused_train as FT
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
torch.manual_seed(0)
pol = InversusCNNPolicy(12, 10, 15, 4).cuda().train()
ft = FT.FusedTrainPolicy(pol)
g = (torch.rand(B, 12, 10, 15, device="cuda") < 0.2).float()
def t(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / reps * 1e3, 3)
res = {"B": B}
with torch.no_grad():
    res["train-forward kernel (+ weight packing)"] = t(lambda: ft.features(g))
# capture the tensors of one backward by running the Function's pieces by hand
feat = ft.features(g)
df = torch.randn_like(feat)
ctx = feat.grad_fn
sv = ctx.saved_tensors
obs, rstd = sv[0], sv[1]; xhat, act, lns, ws = sv[2:6], sv[6:9], sv[9:13], sv[13:17]
lib = L.load()
dy = [torch.empty_like(x) for x in xhat]; dz = [torch.empty_like(x) for x in xhat]
bp, keep = L.InvPolicyBwdParams(), []
for i in range(1, 4):
    pk = FT._pack_conv(ws[i].detach().float().flip(2, 3).permute(1, 0, 2, 3).contiguous(), FT._COUT[i]); keep.append(pk); bp.wt[i - 1] = pk.data_ptr()
for i in range(4): bp.ln[i] = lns[i].data_ptr()
saved, grads = L.InvPolicySaved(), L.InvPolicyGrads()
for i in range(4):
    saved.xhat[i] = xhat[i].data_ptr(); grads.dy[i] = dy[i].data_ptr(); grads.dz[i] = dz[i].data_ptr()
for i in range(3): saved.act[i] = act[i].data_ptr()
saved.rstd = rstd.data_ptr()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
res["backward kernel"] = t(lambda: lib.inv_policy_trunk_bwd(C.byref(bp), C.byref(saved), C.c_void_p(df.data_ptr()), C.byref(grads), B, st))
x0 = torch.zeros((B, 150, 16), dtype=torch.bfloat16, device="cuda"); x0[:, :, :12] = obs.reshape(B, 12, 150).permute(0, 2, 1)
res["x0 build (12 planes -> [B,150,16] bf16)"] = t(lambda: torch.zeros((B, 150, 16), dtype=torch.bfloat16, device="cuda").__setitem__((slice(None), slice(None), slice(0, 12)), obs.reshape(B, 12, 150).permute(0, 2, 1)), 3, 1)
for i in range(4):
    x_in = x0 if i == 0 else act[i - 1]
    res[f"wgrad L{i+1} (HIP kernel)"] = t(lambda: FT._wgrad(i, x_in, dz[i]), 3, 1)
    res[f"ln/bias reductions L{i+1}"] = t(lambda: ((dy[i] * xhat[i]).sum(dim=0, dtype=torch.float32), dy[i].sum(dim=0, dtype=torch.float32)), 3, 1)
print(json.dumps(res, indent=1))
