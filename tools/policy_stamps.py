#!/usr/bin/env python3
"""Phase shares of the fused policy trunk from the diagnostic build (tools/exp/libinversus_hip_stamps.so,
built with -DINV_POLICY_STAMPS).  Shares only -- the stamped build's run time is not representative."""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["INVERSUS_HIP_LIB"] = os.path.join(ROOT, "tools", "exp", "libinversus_hip_stamps.so")
import torch
from inversus_reinforcement_learning_amd import _lib
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
lib = _lib.load()
lib.inv_policy_debug_stamps.argtypes = [C.c_void_p, C.c_int]
B = 16384
pol = InversusCNNPolicy(12, 10, 15, 4).cuda().eval(); f = FusedPolicy(pol)
g = (torch.rand(B, 12, 10, 15, device="cuda") < 0.2).float()
for _ in range(3): f.features(g)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 128)()
lib.inv_policy_debug_stamps(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): f.features(g)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
lib.inv_policy_debug_stamps(buf, 0)
names = (["obs load"] + [f"GEMM L{i+1}" for i in range(4)] + [f"wait stats barrier L{i+1}" for i in range(4)] +
         [f"wait end barrier L{i+1}" for i in range(4)] + ["feature write-out"] + [f"bias+stats L{i+1}" for i in range(4)] +
         [f"epilogue L{i+1}" for i in range(4)])
out = {}
for w in range(4):
    v = list(buf)[w * 32:w * 32 + 22]; tot = sum(v)
    out[f"wave {w}"] = {"cycles per sample": round(tot / (5 * B)), **{n: round(x / (5 * B)) for n, x in zip(names, v)}}
print(json.dumps(out, indent=1))
agg = {n: sum(out[f"wave {w}"][n] for w in range(4)) / 4 for n in names}
tot = sum(agg.values())
print(json.dumps({"mean over waves, % of the sample": {n: round(100 * x / tot, 1) for n, x in agg.items()}, "cycles per sample": round(tot)}, indent=1))

cyc = sum(list(buf)[0:22]) / 5 / B * (B / 512)          # wave 0 of one block: cycles per launch (512 blocks, B/512 samples each)
print(json.dumps({"stamped build ms per launch": round(ms, 4), "wave-0 cycles per launch": round(cyc), "shader clock GHz (s_memtime ticks / wall)": round(cyc / ms / 1e6, 3)}))
