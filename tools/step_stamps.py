#!/usr/bin/env python3
"""Phase shares of k_step from the diagnostic build (-DINV_STEP_STAMPS): resetting vs non-resetting waves."""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["INVERSUS_HIP_LIB"] = os.path.join(ROOT, "tools", "exp", "libinversus_hip_stepstamps.so")
import numpy as np, torch
from inversus_reinforcement_learning_amd import _lib
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
lib = _lib.load()
lib.inv_step_debug_stamps.argtypes = [C.c_void_p, C.c_int]
n = 16384
names = ["load state", "actions/dummy", "tick+reward", "scalar outputs", "reset", "render", "store+drain"]
for diff in ("easy", "hard"):
    b = DeviceEnvBatch(n, opponent="dummy", difficulty=diff, opp_seed=1)
    b.seed(np.arange(n, dtype=np.int64)); b.reset()
    out = b._alloc_out((), True, False, False, None)
    for t in range(150): b.step(b.uniform_actions(1, t), None, auto_reset=True, out=out)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)(); lib.inv_step_debug_stamps(buf, 1)
    for t in range(150, 200): b.step(b.uniform_actions(1, t), None, auto_reset=True, out=out)
    torch.cuda.synchronize(); lib.inv_step_debug_stamps(buf, 0)
    v = list(buf)
    for tag, off in (("no reset", 0), ("reset", 8)):
        cnt = v[off + 7]
        if cnt:
            print(diff, tag, "waves", cnt, {nm: round(v[off + i] / cnt) for i, nm in enumerate(names)}, "total", round(sum(v[off:off + 7]) / cnt))
