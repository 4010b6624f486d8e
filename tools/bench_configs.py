#!/usr/bin/env python3
"""Side measurements for DESIGN.md (not the driver's bench): rollout kernel, 32x32 stress, self-play
obs pair, GAE, and the policy-in-the-loop collector.  HIP events, median of repeats."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch, gae
import bench

def timeit(fn, reps=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2] * 1e3

res = {}
n = 16384
# rollout kernel: T steps per launch, pre-supplied actions
for T in (8, 32):
    b = DeviceEnvBatch(n, opponent="dummy", difficulty="easy", opp_seed=1)
    b.seed(np.arange(n, dtype=np.int64)); b.reset()
    a1 = torch.stack([b.uniform_actions(7, t) for t in range(T)])
    out = b._alloc_out((T,), True, False, False, None)
    us = timeit(lambda: b.rollout(a1, None, out=out), reps=20, warm=3)
    res[f"rollout T={T} easy: us per vector step"] = us / T
    res[f"rollout T={T} easy: GB/s"] = bench.algorithmic_bytes_per_env_step() * n * T / us / 1e3
# the same T steps as back-to-back single-step launches (pre-marshalled), same box
b = DeviceEnvBatch(n, opponent="dummy", difficulty="easy", opp_seed=1)
b.seed(np.arange(n, dtype=np.int64)); b.reset()
T = 32
a1 = torch.stack([b.uniform_actions(7, t) for t in range(T)])
out1 = b._alloc_out((), True, False, False, None)
stepper = b.make_stepper(out1, None, auto_reset=True)
base = a1.data_ptr()
def many():
    for t in range(T):
        stepper(base + t * n)
res["32 single-step launches easy: us per vector step"] = timeit(many, reps=20, warm=3) / T
# 32x32 stress (C5)
for W, H in ((32, 32),):
    b = DeviceEnvBatch(n, W, H, opponent="dummy", difficulty="hard", opp_seed=1)
    b.seed(np.arange(n, dtype=np.int64)); b.reset()
    out = b._alloc_out((), True, False, False, None)
    for t in range(60): b.step(b.uniform_actions(3, t), None, auto_reset=True, out=out)
    a1 = b.uniform_actions(3, 61)
    us = timeit(lambda: b.step(a1, None, auto_reset=True, out=out))
    res["32x32 hard step+obs: us"] = us
    res["32x32 hard step+obs: GB/s"] = bench.algorithmic_bytes_per_env_step(32, 32, 16) * n / us / 1e3
    # adversarial: both players fire RIGHT from x=0 every tick (max live bullets)
    b2 = DeviceEnvBatch(n, W, H, opponent="selfplay")
    b2.seed(np.arange(n, dtype=np.int64)); b2.reset()
    st = b2.get_state(("players",))["players"]; st[:, :, 0] = 0; st[:, 0, 1] = 3; st[:, 1, 1] = 20
    b2.set_state(players=st)
    fire = torch.full((n,), 6, dtype=torch.uint8, device="cuda")
    mx = 0
    for t in range(80):
        b2.step(fire, fire, obs=False)
        mx = max(mx, int(b2.get_state(("nbullets",))["nbullets"].max()))
    b2.check()
    res["32x32 adversarial max live bullets"] = mx
# selfplay obs pair at 15x10
b = DeviceEnvBatch(n, opponent="selfplay")
b.seed(np.arange(n, dtype=np.int64)); b.reset()
out = b._alloc_out((), True, True, False, None)
a1 = b.uniform_actions(1, 0); a2 = b.uniform_actions(2, 0)
us = timeit(lambda: b.step(a1, a2, auto_reset=True, out=out))
res["selfplay step + P1 & P2 obs: us"] = us
res["selfplay step + P1 & P2 obs: GB/s"] = bench.algorithmic_bytes_per_env_step(selfplay=True) * n / us / 1e3
# GAE
T = 128
r = torch.randn(T, n, device="cuda"); v = torch.randn(T, n, device="cuda"); d = torch.rand(T, n, device="cuda") < 0.03
for layout in ("per_env", "reference_flat"):
    us = timeit(lambda: gae(r, v, d, layout=layout), reps=30, warm=3)
    res[f"GAE {layout} T=128 N=16384: us"] = us
    res[f"GAE {layout}: GB/s (17 B/transition)"] = 17 * T * n / us / 1e3
# policy in the loop (C3-style, selfplay): collector steps/s with the f32 torch CNN
from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
from inversus_reinforcement_learning_amd.training import RolloutCollector
from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
for fused in (False, True):
    for nn_ in (2048, 16384):
        b = DeviceEnvBatch(nn_, opponent="selfplay")
        b.seed(np.arange(nn_, dtype=np.int64)); b.reset()
        torch.manual_seed(0)
        pol = InversusCNNPolicy(12, 10, 15, 4).cuda(); opp = InversusCNNPolicy(12, 10, 15, 4).cuda()
        agent = PPOAgent(pol, fused_act=fused)
        col = RolloutCollector(b, agent, FusedPolicy(opp) if fused else opp)
        col.collect(2); agent.reset_buffers(); torch.cuda.synchronize()
        t0 = time.perf_counter(); col.collect(6); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        agent.reset_buffers()
        tag = "fused bf16 trunk" if fused else "f32 torch CNN"
        res[f"selfplay collector ({tag} x2) N={nn_}: env-steps/s"] = 6 * nn_ / dt
        del col, agent, pol, opp, b
        torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
