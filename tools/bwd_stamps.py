#!/usr/bin/env python3
"""Phase shares of k_policy_trunk_bwd from the diagnostic build (tools/exp/libinversus_hip_stamps.so, built with
-DINV_POLICY_STAMPS; stamps accumulate in LDS, flushed once per workgroup).  Shares only -- the stamped build's run time is not
representative.  Slots: pass A / wait sums / pass B / wait dz tile per layer 4..1; transposed-convolution GEMM / wait / da write per layer 4..2."""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["INVERSUS_HIP_LIB"] = os.path.join(ROOT, "tools", "exp", "libinversus_hip_stamps.so")
import torch
from inversus_reinforcement_learning_amd import _lib as L
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd import fused_train as FT
lib = L.load()
lib.inv_policy_debug_stamps.argtypes = [C.c_void_p, C.c_int]
B = 16384
torch.manual_seed(0)
pol = InversusCNNPolicy(12, 10, 15, 4).cuda().train()
with torch.no_grad():
    for nm in (pol.norm1, pol.norm2, pol.norm3, pol.norm4):
        nm.weight.add_(0.2 * torch.randn_like(nm.weight)); nm.bias.add_(0.2 * torch.randn_like(nm.bias))
ft = FT.FusedTrainPolicy(pol)
g = (torch.rand(B, 12, 10, 15, device="cuda") < 0.2).float()
feat = ft.features(g)
sv = feat.grad_fn.saved_tensors
obs, rstd = sv[0], sv[1]; xhat, act, lns, ws = sv[2:6], sv[6:9], sv[9:13], sv[13:17]
df = (torch.randn_like(feat.float()) * 0.01).to(torch.bfloat16)
dy = [torch.zeros_like(x) for x in xhat]; dz = [torch.zeros_like(x) for x in xhat]
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
bwd = lambda: L.raise_for(lib.inv_policy_trunk_bwd(C.byref(bp), C.byref(saved), C.c_void_p(df.data_ptr()), C.byref(grads), B, st), lib)
for _ in range(3): bwd()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 128)()
lib.inv_policy_debug_stamps(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): bwd()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
lib.inv_policy_debug_stamps(buf, 0)
names = ([f"pass A L{l + 1}" for l in range(4)] + [f"wait sums L{l + 1}" for l in range(4)] + [f"pass B L{l + 1}" for l in range(4)] +
         [f"wait dz tile L{l + 1}" for l in range(4)] + [f"convT GEMM from L{l + 1}" for l in range(4)] + [f"wait GEMM from L{l + 1}" for l in range(4)] +
         [f"da write + barrier from L{l + 1}" for l in range(4)])
per = {}
for w in range(4):
    v = list(buf)[w * 32:w * 32 + 28]
    per[w] = v
agg = {n: sum(per[w][i] for w in range(4)) / 4 / (5 * B) for i, n in enumerate(names)}
tot = sum(agg.values())
print(json.dumps({"cycles per sample (mean over waves)": {n: round(x) for n, x in agg.items() if x > 0}}, indent=1))
groups = {"pass A": 0.0, "wait sums": 0.0, "pass B": 0.0, "wait dz tile": 0.0, "convT GEMM": 0.0, "wait GEMM": 0.0, "da write + barrier": 0.0}
for n, x in agg.items():
    for k in groups:
        if n.startswith(k):
            groups[k] += x
print(json.dumps({"% of the sample": {k: round(100 * x / tot, 1) for k, x in groups.items()}, "cycles per sample": round(tot),
                  "stamped build ms per launch": round(ms, 4)}, indent=1))
