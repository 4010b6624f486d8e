#!/usr/bin/env python3
"""A/B of library builds (tools/build_variant.py) on the step kernel, cache-resident and streaming regimes:
    python tools/step_variants_ab.py tools/exp/lib_a.so tools/exp/lib_b.so ...
Each build runs in its own process (INVERSUS_HIP_LIB), two alternations; us per vector step at N = 16384 (HIP events,
median of 5 regions) and a checksum of the observations / rewards a fixed replay produces (must be equal across builds)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch, StepResult
    n, K = 16384, 16
    cfgs = json.loads(os.environ.get("AB_CFGS", '[["easy",15,10,"dummy",false]]'))

    def timed(fn, reps, warm):
        for _ in range(warm): fn()
        out = []
        for _ in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / reps * 1e3)
        return round(sorted(out)[2], 2)

    res = {}
    for diff, W, H, opp, p2 in cfgs:
        for kind in os.environ.get("AB_KINDS", "row4,wave").split(","):
            row = {}
            for T in (1, 32):
                if W == 32 and T == 32:
                    T = 4                                            # 0.8 GB per buffer
                b = DeviceEnvBatch(n, W, H, K, opponent=opp, difficulty=diff, opp_seed=1, step_kernel=kind)
                b.seed(np.arange(n, dtype=np.int64)); b.reset()
                acts = torch.stack([b.uniform_actions(5, t) for t in range(64)])
                a2 = torch.stack([b.uniform_actions(6, t) for t in range(64)]) if opp == "selfplay" else None
                o = b._alloc_out((T,), True, p2, False, None, False)
                outs = [StepResult(**{k2: (getattr(o, k2)[j] if getattr(o, k2) is not None else None) for k2 in StepResult.__slots__}) for j in range(T)]
                st = [b.make_stepper(outs[j % T], None if a2 is None else a2[j], auto_reset=True) for j in range(64)]
                k = [0]
                def f():
                    j = k[0] % 64; k[0] += 1; st[j](acts[j].data_ptr())
                row["resident" if T == 1 else f"streaming{T}"] = timed(f, 256, 64)
                b.check()
                if T == 1:
                    row["checksum"] = [float(o.obs.double().sum()), float(o.reward.double().sum()), int(o.done.sum())]
                del b, o, outs, st
                torch.cuda.empty_cache()
            res[f"{diff} {W}x{H} {opp}{' +P2' if p2 else ''} {kind}"] = row
    print(json.dumps(res))
    sys.exit(0)
for rep in range(int(os.environ.get("AB_ROUNDS", "2"))):
    for lib in sys.argv[1:]:
        env = dict(os.environ, INVERSUS_HIP_LIB=os.path.join(ROOT, lib))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        print(os.path.basename(lib), r.stdout.strip() or r.stderr[-600:], flush=True)
