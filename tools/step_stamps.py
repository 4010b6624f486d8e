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
lib.inv_step_debug_stamps.argtypes = [C.c_void_p]
n = 16384
names = ["load state", "actions/dummy", "tick+reward", "scalar outputs", "reset", "render", "store+drain"]
for diff in ("easy", "hard"):
    b = DeviceEnvBatch(n, opponent="dummy", difficulty=diff, opp_seed=1)
    b.seed(np.arange(n, dtype=np.int64)); b.reset()
    out = b._alloc_out((), True, False, False, None)
    acc = np.zeros((2, 8)); cnt = np.zeros(2)
    for t in range(230):
        b.step(b.uniform_actions(1, t), None, auto_reset=True, out=out)
        if t >= 200:
            torch.cuda.synchronize()
            buf = np.zeros((n, 8), np.uint64)
            lib.inv_step_debug_stamps(buf.ctypes.data_as(C.c_void_p))
            for r in (0, 1):
                m = buf[:, 7] == r
                acc[r] += buf[m].astype(np.float64).sum(0); cnt[r] += m.sum()
    for r, tag in ((0, "no reset"), (1, "reset")):
        if cnt[r]:
            print(diff, tag, "waves/launch", round(cnt[r] / 30, 1), {nm: int(acc[r, i] / cnt[r]) for i, nm in enumerate(names)}, "total cycles", int(acc[r, :7].sum() / cnt[r]))
