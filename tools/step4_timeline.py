#!/usr/bin/env python3
"""Where the waves of ONE k_step4 launch are on the wall clock (diagnostic build -DINV_STEP_STAMPS, tools/exp/lib_stamps.so):
per phase boundary the min / 10 % / median / 90 % / max over the 4 096 waves, in us after the first wave started; for the
cache-resident regime (one output buffer) and the streaming regime (32 distinct buffers)."""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["INVERSUS_HIP_LIB"] = os.path.join(ROOT, "tools", "exp", os.environ.get("STAMPS_LIB", "lib_stamps.so"))
import numpy as np, torch
from inversus_reinforcement_learning_amd import _lib
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch, StepResult
lib = _lib.load()
lib.inv_step_debug_stamps.argtypes = [C.c_void_p]
n = 16384
names = ["wave start", "state loaded", "tick+reward+scalar outputs", "resets", "obs stores issued", "all stores acked"]
for diff in sys.argv[1:] or ["easy"]:
    for T in (1, 32):
        b = DeviceEnvBatch(n, opponent="dummy", difficulty=diff, opp_seed=1, step_kernel="row4")
        b.seed(np.arange(n, dtype=np.int64)); b.reset()
        o = b._alloc_out((T,), True, False, False, None)
        outs = [StepResult(**{k2: (getattr(o, k2)[j] if getattr(o, k2) is not None else None) for k2 in StepResult.__slots__}) for j in range(T)]
        acts = torch.stack([b.uniform_actions(5, t) for t in range(64)])
        st = [b.make_stepper(outs[j % T], None, auto_reset=True) for j in range(64)]
        rows = []
        for t in range(200):
            st[t % 64](acts[t % 64].data_ptr())
            if t >= 150 and t % 5 == 0:
                torch.cuda.synchronize()
                buf = np.zeros((n, 8), np.uint64)
                lib.inv_step_debug_stamps(buf.ctypes.data_as(C.c_void_p))
                w = buf[: n // 4, :6].astype(np.int64)
                w = (w - w[:, 0].min()) / 100.0                              # 100 MHz -> us
                rows.append(np.stack([np.percentile(w, p, axis=0) for p in (0, 10, 50, 90, 100)]))
                for _ in range(8):                                           # refill the queue so that the sampled launch is back to back
                    pass
        r = np.mean(rows, axis=0)
        print(f"{diff}, {'one buffer (cache-resident)' if T == 1 else '32 buffers (streaming)'}: us after the first wave's start, min / p10 / median / p90 / max over waves")
        for i, nm in enumerate(names):
            print(f"  {nm:28s}" + "  ".join(f"{r[k, i]:7.2f}" for k in range(5)))
        del b, o, outs, st
        torch.cuda.empty_cache()
