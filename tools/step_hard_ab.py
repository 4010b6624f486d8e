#!/usr/bin/env python3
"""A/B of library builds on the vs_dummy step (hard and easy): python tools/step_hard_ab.py lib1.so lib2.so ... (own process each)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    n = 16384
    res = {}
    for diff in ("hard", "easy"):
        b = DeviceEnvBatch(n, 15, 10, 16, opponent="dummy", difficulty=diff, opp_seed=1)
        b.seed(np.arange(n, dtype=np.int64)); b.reset()
        acts = torch.stack([b.uniform_actions(5, t) for t in range(64)])
        out = b._alloc_out((), True, False, False, None)
        stp = b.make_stepper(out, None, auto_reset=True)
        k = [0]
        def f():
            stp(acts[k[0] % 64].data_ptr()); k[0] += 1
        for _ in range(256): f()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(512): f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 512 * 1e3)
        res[diff + " us"] = round(sorted(ts)[2], 3)
        res[diff + " checksum"] = float(out.reward.double().sum()) + float(out.obs.double().sum())
    print(json.dumps(res)); sys.exit(0)
for rep in range(3):
    for lib in sys.argv[1:]:
        env = dict(os.environ, INVERSUS_HIP_LIB=os.path.join(ROOT, lib))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        print(lib, r.stdout.strip() or r.stderr[-600:], flush=True)
