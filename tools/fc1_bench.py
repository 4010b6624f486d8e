"""Same-process A/B of the first Linear's three products: this library's kernels (csrc/inv_fc1.hip) vs the GEMM library through
torch, interleaved rounds on random data (ranking builds by separate runs or zero-filled operands misleads: the chip's clock
depends on the data).  Prints one JSON object.  usage: python tools/fc1_bench.py [B] [rounds] [pixels = 150 | 1024 (the 32 x 32 board: K = 131 072)]"""
import ctypes as C
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from inversus_reinforcement_learning_amd import _lib as L  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    K, NH = 128 * P, 512
    lib = L.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    feat = (torch.randn(B, K, device=dev, generator=g).clamp_min(0) * 0.5).to(torch.bfloat16)
    wf = (torch.randn(NH, K, device=dev, generator=g) * 0.01).to(torch.bfloat16)
    wft = wf.t().contiguous()
    dpre = (torch.randn(B, NH, device=dev, generator=g) * 0.01).to(torch.bfloat16)
    dpre_t = dpre.t().contiguous()
    dwe = torch.zeros(NH, 4, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    slabs = int(lib.inv_fc1_fwd_slabs(B, NH, K))
    pre = torch.empty((slabs, B, NH), dtype=torch.float32, device=dev)
    dfeat = torch.empty((B, K), dtype=torch.bfloat16, device=dev)
    gw = [torch.empty((256, K + 4), dtype=torch.float32, device=dev) for _ in range(2)]
    ws = torch.empty(int(lib.inv_fc1_dw_workspace_bytes(B, NH, K)), dtype=torch.uint8, device=dev)

    ours = {
        "fwd": lambda: L.raise_for(lib.inv_fc1_fwd(p(feat), p(wf), B, NH, K, p(pre), st), lib),
        "dfeat": lambda: L.raise_for(lib.inv_fc1_dfeat(p(dpre), p(wft), B, NH, K, p(dfeat), st), lib),
        "dw": lambda: L.raise_for(lib.inv_fc1_dw(p(dpre_t), p(feat), p(dwe), B, NH, P, p(gw[0]), p(gw[1]), p(ws), st), lib),
    }
    theirs = {
        "fwd": lambda: feat @ wf.t(),
        "dfeat": lambda: dpre @ wft.t(),
        "dw": lambda: dpre_t @ feat,
    }

    def time(fn, n=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n

    for fn in list(ours.values()) + list(theirs.values()):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    res = {k: {"ours_ms": [], "library_ms": []} for k in ours}
    for _ in range(rounds):
        for k in ours:
            res[k]["ours_ms"].append(round(time(ours[k]), 4))
            res[k]["library_ms"].append(round(time(theirs[k]), 4))
    flop = 2.0 * B * NH * K
    out = {"B": B, "slabs": slabs, "gpu": torch.cuda.get_device_name(0)}
    for k, v in res.items():
        a, b = sorted(v["ours_ms"]), sorted(v["library_ms"])
        out[k] = {"ours_ms_median": a[len(a) // 2], "ours_ms_min": a[0], "library_ms_median": b[len(b) // 2], "library_ms_min": b[0],
                  "ours_pflops": round(flop / (a[len(a) // 2] * 1e-3) / 1e15, 3), "library_pflops": round(flop / (b[len(b) // 2] * 1e-3) / 1e15, 3)}
    # numerics on the random data (f32 accumulation both ways)
    ours["fwd"]()
    ref = feat[:1024].float() @ wf.float().t()
    out["fwd_rel_err"] = float(((pre.sum(0)[:1024] - ref).norm() / ref.norm()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
