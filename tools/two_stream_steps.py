#!/usr/bin/env python3
"""Experiment: the 16 384 envs of a GPU as TWO independent half-batches stepped on two HIP streams (env shards inside one GPU, like
the shards across GPUs: no data dependency between them), so that one half's launch boundary and the ~2 us in front of its first
store run under the other half's observation stream.  us per FULL vector step (both halves), cache-resident (one output buffer per
half) and streaming (32 distinct buffers per half); single-batch figures of the same process beside them."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch, StepResult
N, K = 16384, 16


def make(n, off, T, stream):
    with torch.cuda.stream(stream):
        b = DeviceEnvBatch(n, 15, 10, K, opponent="dummy", difficulty=os.environ.get("DIFF", "easy"), opp_seed=1, env_offset=off)
        b.seed(off + np.arange(n, dtype=np.int64)); b.reset()
        acts = torch.stack([b.uniform_actions(5, t) for t in range(64)])
        o = b._alloc_out((T,), True, False, False, None)
        outs = [StepResult(**{k: (getattr(o, k)[j] if getattr(o, k) is not None else None) for k in StepResult.__slots__}) for j in range(T)]
        st = [b.make_stepper(outs[j % T], None, auto_reset=True) for j in range(64)]      # bound to `stream`
    stream.synchronize()
    return b, acts, o, st


def run(parts, reps=512):
    for t in range(64):
        for b, acts, o, st in parts:
            st[t % 64](acts[t % 64].data_ptr())
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        t0 = time.perf_counter()
        for t in range(reps):
            for b, acts, o, st in parts:
                st[t % 64](acts[t % 64].data_ptr())
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / reps * 1e6)
    for b, *_ in parts:
        b.check()
    return round(sorted(best)[2], 2)


res = {}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for T, tag in ((1, "cache-resident"), (32, "streaming (32 buffers)")):
    one = [make(N, 0, T, s1)]
    res[f"{tag}: one batch of 16384, one stream"] = run(one)
    del one; torch.cuda.empty_cache()
    two = [make(N // 2, 0, T, s1), make(N // 2, N // 2, T, s2)]
    res[f"{tag}: two halves of 8192 on two streams"] = run(two)
    del two; torch.cuda.empty_cache()
    four = [make(N // 4, q * N // 4, T, torch.cuda.Stream()) for q in range(4)]
    res[f"{tag}: four quarters of 4096 on four streams"] = run(four)
    del four; torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
