#!/usr/bin/env python3
"""A/B of two builds of the library on the same box: hard-mode and easy-mode step time (subprocess per build)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np, torch
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
n = 16384
res = {}
for diff in ("easy", "hard"):
    b = DeviceEnvBatch(n, opponent="dummy", difficulty=diff, opp_seed=1)
    b.seed(np.arange(n, dtype=np.int64)); b.reset()
    out = b._alloc_out((), True, False, False, None)
    acts = torch.stack([b.uniform_actions(5, t) for t in range(64)])
    st = b.make_stepper(out, None, auto_reset=True)
    for t in range(200): st(acts[t %% 64].data_ptr())
    ts = []
    for rep in range(7):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(256): st(acts[t %% 64].data_ptr())
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 256 * 1e3)
    res[diff] = round(sorted(ts)[3], 2)
print(json.dumps(res))
''' % ROOT
for rnd in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ, INVERSUS_HIP_LIB=os.path.join(ROOT, "tools", "exp", name))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        print(name, out[-1] if out else "no output", flush=True)
