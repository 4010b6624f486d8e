#!/usr/bin/env python3
"""A/B of library builds on the fused update kernels: python tools/bwd_ab.py lib1.so lib2.so ...
Per build (own process): ms of the training forward, the backward kernel, and a checksum of dz / dy."""
import os, sys, json, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, ROOT)
    from inversus_reinforcement_learning_amd import _lib as L
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    from inversus_reinforcement_learning_amd import fused_train as FT
    B = 16384
    torch.manual_seed(0)
    pol = InversusCNNPolicy(12, 10, 15, 4).cuda().train()
    with torch.no_grad():
        for nm in (pol.norm1, pol.norm2, pol.norm3, pol.norm4):
            nm.weight.add_(0.2 * torch.randn_like(nm.weight)); nm.bias.add_(0.2 * torch.randn_like(nm.bias))
    ft = FT.FusedTrainPolicy(pol)
    g = (torch.rand(B, 12, 10, 15, device="cuda") < 0.2).float()
    def timed(fn):
        out = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): fn()
            e1.record(); torch.cuda.synchronize(); out.append(e0.elapsed_time(e1) / 3)
        return round(min(out), 3)
    res = {}
    feat = ft.features(g)
    sv = feat.grad_fn.saved_tensors
    obs, rstd = sv[0], sv[1]; xhat, act, lns, ws = sv[2:6], sv[6:9], sv[9:13], sv[13:17]
    lib = L.load()
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
    bwd = lambda: lib.inv_policy_trunk_bwd(C.byref(bp), C.byref(saved), C.c_void_p(df.data_ptr()), C.byref(grads), B, st)
    bwd(); torch.cuda.synchronize()
    res["checksum dz"] = [round(float(t.float().abs().sum()), 3) for t in dz]
    res["checksum dy"] = [round(float(t.float().abs().sum()), 3) for t in dy]
    res["checksum xhat/act"] = [round(float(t.float().abs().sum()), 1) for t in list(xhat) + list(act)]
    with torch.no_grad():
        res["train fwd ms"] = timed(lambda: ft.features(g))
    res["bwd ms"] = timed(bwd)
    print(json.dumps(res)); sys.exit(0)
for rep in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ, INVERSUS_HIP_LIB=os.path.join(ROOT, lib))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        print(lib, r.stdout.strip() or r.stderr[-600:], flush=True)
