#!/usr/bin/env python3
"""Phase shares of the banded 32 x 32 trunk from the diagnostic build (tools/build_variant.py polstamps -DINV_POLICY_STAMPS).
Per layer: load (band input into LDS, LayerNorm on the way) / GEMM + statistics / wait at the barrier / z write-back."""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["INVERSUS_HIP_LIB"] = os.path.join(ROOT, "tools", "exp", "lib_polstamps.so")
import torch
from inversus_reinforcement_learning_amd import _lib
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
lib = _lib.load()
lib.inv_policy_debug_stamps.argtypes = [C.c_void_p, C.c_int]
B = 4096
pol = InversusCNNPolicy(12, 32, 32, 4).cuda().eval(); f = FusedPolicy(pol)
g = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 384), dtype=torch.int32, device="cuda") & torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 384), dtype=torch.int32, device="cuda")
for _ in range(2): f.features(g)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 128)()
lib.inv_policy_debug_stamps(buf, 1)
for _ in range(3): f.features(g)
torch.cuda.synchronize()
lib.inv_policy_debug_stamps(buf, 0)
names = []
for l in range(4):
    names += [f"L{l+1} load", f"L{l+1} GEMM+stats", f"L{l+1} barrier wait", f"L{l+1} z write-back"]
names += ["stats + feature pass"]
agg = [sum(list(buf)[w * 32 + i] for w in range(4)) / 4 / (3 * B) for i in range(17)]
tot = sum(agg)
print(json.dumps({"cycles per sample (mean over waves)": round(tot), "% of the sample": {n: round(100 * x / tot, 1) for n, x in zip(names, agg)}}, indent=1))
