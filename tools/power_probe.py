#!/usr/bin/env python3
"""Board power and shader clock (rocm-smi, polled from a thread) while one workload loops for a few seconds each:
the first Linear's forward GEMM (MFMA-bound), the LayerNorm parameter reduction (HBM-bound), both on two streams, the 15x10 trunk
forward, the banded 32x32 forward, the env step (store-bound).  Prints one JSON object; measurement only."""
import ctypes as C, json, re, subprocess, sys, threading, time
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from inversus_reinforcement_learning_amd import _lib as L
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch

lib = L.load()
samples, stop = [], False


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            card = d[sorted(d)[0]]
            pw = [float(v) for k, v in card.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))]
            sclk = [v for k, v in card.items() if k.lower().startswith("sclk")]
            mclk = [v for k, v in card.items() if k.lower().startswith("mclk")]
            samples.append((time.time(), pw[0] if pw else None, str(sclk[0]) if sclk else None, str(mclk[0]) if mclk else None))
        except Exception as e:                                    # the tool's output format is what it is: record and go on
            samples.append((time.time(), None, repr(e)[:80], None))
        time.sleep(0.15)


th = threading.Thread(target=poll, daemon=True)
th.start()
dev = "cuda"
B, K, NH = 16384, 19200, 512
g = torch.Generator(device=dev).manual_seed(0)
feat = (torch.randn(B, K, device=dev, generator=g).clamp_min(0) * 0.5).to(torch.bfloat16)
wf = (torch.randn(NH, K, device=dev, generator=g) * 0.01).to(torch.bfloat16)
slabs = int(lib.inv_fc1_fwd_slabs(B, NH, K))
pre = torch.empty((slabs, B, NH), dtype=torch.float32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)            # 1 GiB: an HBM-bound copy
big2 = torch.empty_like(big)
pol15 = FusedPolicy(InversusCNNPolicy(12, 10, 15, 4).cuda().eval())
pol32 = FusedPolicy(InversusCNNPolicy(12, 32, 32, 4).cuda().eval())
o15 = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 57), dtype=torch.int32, device=dev) & torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 57), dtype=torch.int32, device=dev)
o32 = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 384), dtype=torch.int32, device=dev) & torch.randint(-2 ** 31, 2 ** 31 - 1, (B, 384), dtype=torch.int32, device=dev)
env = DeviceEnvBatch(B, 15, 10, opponent="dummy", difficulty="easy", opp_seed=1)
env.seed(np.arange(B, dtype=np.int64)); env.reset()
acts = env.uniform_actions(1, 0)
side = torch.cuda.Stream()


def gemm():
    L.raise_for(lib.inv_fc1_fwd(p(feat), p(wf), B, NH, K, p(pre), C.c_void_p(torch.cuda.current_stream().cuda_stream)), lib)


def copy():
    big2.copy_(big)


def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        copy()
    gemm()
    torch.cuda.current_stream().wait_stream(side)


work = {"idle": None, "fc1 forward GEMM": gemm, "1 GiB device copy": copy, "GEMM + copy on two streams": both,
        "15x10 trunk forward": lambda: pol15.features(o15), "32x32 banded forward": lambda: pol32.features(o32),
        "env step + obs (16384 envs)": lambda: env.step(acts, None, auto_reset=True)}
res = {}
for name, fn in work.items():
    t0 = time.time()
    n = 0
    if fn is None:
        time.sleep(2.0)
    else:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.time()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < 4.0:
            for _ in range(20):
                fn()
            n += 20
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
    t1 = time.time()
    mine = [s for s in samples if t0 + 1.0 <= s[0] <= t1]          # skip the first second (ramp)
    pw = [s[1] for s in mine if s[1] is not None]
    res[name] = {"ms per call": round(e0.elapsed_time(e1) / n, 4) if n else None, "power W (mean, max)": [round(sum(pw) / len(pw), 1), max(pw)] if pw else None,
                 "sclk samples": sorted(set(s[2] for s in mine))[:6], "mclk samples": sorted(set(str(s[3]) for s in mine))[:3], "polls": len(mine)}
    print(json.dumps({name: res[name]}), flush=True)
stop = True
