#!/usr/bin/env python3
"""A/B of alternative builds of the library on the fused trunk: python tools/trunk_ab.py lib1.so lib2.so[@VAR=VALUE[,VAR=VALUE]] ...
Each build runs in its own process (INVERSUS_HIP_LIB, plus the variables after `@`, under a 300-s timeout); prints ms per
16384-sample forward (min / median of 7)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
    torch.manual_seed(0)
    n = 16384
    pol = InversusCNNPolicy(12, 10, 15, 4).cuda().eval()
    with torch.no_grad():                                    # non-trivial LayerNorm affine parameters
        for nm in (pol.norm1, pol.norm2, pol.norm3, pol.norm4):
            nm.weight.add_(0.3 * torch.randn_like(nm.weight)); nm.bias.add_(0.3 * torch.randn_like(nm.bias))
    f = FusedPolicy(pol)
    if os.environ.get("INVERSUS_TRUNK32") == "1":
        # experiment builds only (profiles/r05_trunk_experiments.md, section 7): layers 3 / 4 of the inference kernel run on
        # v_mfma_f32_32x32x16_bf16 and read their weights in that shape's A-fragment order:
        #   [Cout/32][tap * Cin/16 + kk][lane = 32 h + r][j] = W[32 mt + r][16 kk + 8 h + j][tap]
        def pack32(w):
            cout, cin = w.shape[:2]
            wk = w.detach().float().permute(0, 2, 3, 1).reshape(cout // 32, 32, 9, cin // 16, 2, 8).permute(0, 2, 3, 4, 1, 5)
            return wk.reshape(cout // 32, 9 * cin // 16, 64, 8).contiguous().to(torch.bfloat16)
        keep32 = [pack32(pol.conv3.weight), pack32(pol.conv4.weight)]
        f.params.layer[2].w, f.params.layer[3].w = keep32[0].data_ptr(), keep32[1].data_ptr()
    g = (torch.rand(n, 12, 10, 15, device="cuda") < 0.2).float()
    ref = f.features(g).float()
    with torch.no_grad():
        want = f.policy.trunk(g[:512]).reshape(512, 128, 150).permute(0, 2, 1).reshape(512, 19200)
    err = (ref[:512] - want).abs()
    import copy
    pol16 = copy.deepcopy(pol)                               # the same module with its LayerNorm affine rounded to fp16
    with torch.no_grad():
        for nm in (pol16.norm1, pol16.norm2, pol16.norm3, pol16.norm4):
            nm.weight.copy_(nm.weight.half().float()); nm.bias.copy_(nm.bias.half().float())
        want16 = pol16.trunk(g[:512]).reshape(512, 128, 150).permute(0, 2, 1).reshape(512, 19200)
    err16 = (ref[:512] - want16).abs()
    out = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            f.features(g)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 5)
    gb = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, 57), dtype=torch.int32, device="cuda") & torch.randint(-2 ** 31, 2 ** 31 - 1, (n, 57), dtype=torch.int32, device="cuda")
    outb = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            f.features(gb)
        e1.record(); torch.cuda.synchronize()
        outb.append(e0.elapsed_time(e1) / 5)
    print(json.dumps({"packed_min_ms": round(min(outb), 4), "packed_median_ms": round(sorted(outb)[3], 4), "min_ms": round(min(out), 4), "median_ms": round(sorted(out)[3], 4), "checksum": float(ref.abs().sum()), "max_err": round(float(err.max()), 4), "mean_err": round(float(err.mean()), 5), "max_err_vs_fp16_ln_ref": round(float(err16.max()), 4), "feat_absmax": round(float(want.abs().max()), 2)}))
    sys.exit(0)
for rep in range(2):
    for spec in sys.argv[1:]:
        lib, _, extra = spec.partition("@")
        env = dict(os.environ, INVERSUS_HIP_LIB=os.path.join(ROOT, lib))
        env.update(dict(kv.split("=", 1) for kv in extra.split(",") if kv))
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=300)
            print(spec, r.stdout.strip() or r.stderr[-400:], flush=True)
        except subprocess.TimeoutExpired:
            print(spec, "TIMEOUT after 300 s -- stopping (no further GPU step after a hung one)", flush=True)
            sys.exit(1)
