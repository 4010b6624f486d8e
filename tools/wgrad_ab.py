#!/usr/bin/env python3
"""A/B of library builds on the weight-gradient kernel (per layer, B = 16384): python tools/wgrad_ab.py lib1.so ..."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, ROOT)
    from inversus_reinforcement_learning_amd import fused_train as FT
    B = 16384
    torch.manual_seed(0)
    res = {}
    for layer in range(4):
        cinp, cout = FT._CINP[layer], FT._COUT[layer]
        x = torch.randn(B, 150, cinp, device="cuda").to(torch.bfloat16)
        dz = (torch.randn(B, 150, cout, device="cuda") * 0.1).to(torch.bfloat16)
        out = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): dw, db = FT._wgrad(layer, x, dz)
            e1.record(); torch.cuda.synchronize(); out.append(e0.elapsed_time(e1) / 3)
        res[f"L{layer + 1} ms"] = round(min(out), 3)
        res[f"L{layer + 1} sum"] = round(float(dw.abs().sum()), 1)
    # the 32 x 32 board's 128 -> 128 layer through the banded entry point (units of 4 rows)
    import ctypes as C
    from inversus_reinforcement_learning_amd import _lib as L
    lib = L.load()
    x = torch.randn(B, 1024, 128, device="cuda").to(torch.bfloat16)
    dz = (torch.randn(B, 1024, 128, device="cuda") * 0.1).to(torch.bfloat16)
    out = []
    for _ in range(4):
        dw = torch.zeros(128, 9, 128, device="cuda"); db = torch.zeros(128, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.raise_for(lib.inv_policy_band_wgrad(3, 32, 32, C.c_void_p(x.data_ptr()), C.c_void_p(dz.data_ptr()), C.c_void_p(dw.data_ptr()), C.c_void_p(db.data_ptr()), B,
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), lib)
        e1.record(); torch.cuda.synchronize(); out.append(e0.elapsed_time(e1))
    res["32x32 L4 ms"] = round(min(out), 3)
    res["32x32 L4 sum"] = round(float(dw.abs().sum()), 1)
    print(json.dumps(res)); sys.exit(0)
for lib in sys.argv[1:]:
    env = dict(os.environ, INVERSUS_HIP_LIB=os.path.join(ROOT, lib))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
    print(lib, r.stdout.strip() or r.stderr[-600:], flush=True)
