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
buf = (C.c_ulonglong * 16)()
lib.inv_policy_debug_stamps(buf, 1)
for _ in range(5): f.features(g)
torch.cuda.synchronize()
lib.inv_policy_debug_stamps(buf, 0)
v = list(buf); tot = sum(v)
names = ["obs load"] + [f"GEMM L{i+1}" for i in range(4)] + [f"stats+barrier L{i+1}" for i in range(4)] + [f"epilogue L{i+1}" for i in range(4)] + ["feature write-out"]
print(json.dumps({n: round(100.0 * x / tot, 1) for n, x in zip(names, v)}, indent=1))
print("cycles per sample per block (wave 0):", tot / (5 * B))
