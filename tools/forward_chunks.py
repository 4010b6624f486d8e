#!/usr/bin/env python3
"""Does the policy forward get faster when the batch is cut so that a chunk's feature vector (bf16 [chunk][128 P]) stays in the 256-MB
Infinity Cache between the trunk that writes it and the first Linear that reads it?  ms per 16 384-sample forward (trunk + heads),
one call vs 2 / 4 / 8 / 16 chunks; python tools/forward_chunks.py [15x10 | 32x32]"""
import sys, json, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
board = sys.argv[1] if len(sys.argv) > 1 else "15x10"
W, H = (15, 10) if board == "15x10" else (32, 32)
torch.manual_seed(0)
B = 16384
f = FusedPolicy(InversusCNNPolicy(12, H, W, 4).cuda().eval())
ow = (12 * W * H + 31) // 32
g = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, ow), dtype=torch.int32, device="cuda") & torch.randint(-2 ** 31, 2 ** 31 - 1, (B, ow), dtype=torch.int32, device="cuda")
if (12 * W * H) % 32:
    g[:, -1] &= (1 << ((12 * W * H) % 32)) - 1
x = torch.rand(B, 4, device="cuda")
res = {}
for chunks in (1, 2, 4, 8, 16, 1):
    n = B // chunks
    def run():
        for c in range(chunks):
            f(g[c * n:(c + 1) * n], x[c * n:(c + 1) * n])
    for _ in range(3): run()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ts.append(round(e0.elapsed_time(e1) / 10, 4))
    res.setdefault(f"{chunks} chunk(s) of {n}: features {n * 128 * W * H * 2 / 1e6:.0f} MB", []).append(ts)
print(json.dumps({"board": board, "ms per 16384-sample forward": res}, indent=1))
