#!/usr/bin/env python3
"""Where does hard mode spend its extra time?  (measurement helper)"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
from tools.ablate import timeit
n = 16384
res = {}
def run(name, diff, a2mode, steady=150, max_steps=500, obs=False):
    b = DeviceEnvBatch(n, opponent="dummy", difficulty=diff, opp_seed=1, max_episode_steps=max_steps)
    b.seed(np.arange(n, dtype=np.int64)); b.reset()
    out = b._alloc_out((), obs, False, False, None)
    acts = [b.uniform_actions(1, t) for t in range(8)]
    a2s = [b.uniform_actions(2, t) for t in range(8)]
    k = [0]
    def f():
        k[0] += 1
        a2 = None if a2mode == "dummy" else (a2s[k[0] % 8] if a2mode == "random" else torch.zeros_like(acts[0]))
        b.step(acts[k[0] % 8], a2, auto_reset=True, out=out)
    for _ in range(steady): f()
    us = timeit(f)
    st = b.get_state(("nbullets", "step_count"))
    done_rate = float((st["step_count"] == 0).mean())
    res[name] = {"us": round(us, 2), "mean_bullets": round(float(st["nbullets"].mean()), 2), "frac_just_reset": round(done_rate, 4)}
run("easy dummy", "easy", "dummy")
run("hard dummy", "hard", "dummy")
run("a2 random (no dummy logic)", "easy", "random")
run("a2 zero", "easy", "zero")
run("easy dummy, max_steps=25 (timeouts force resets)", "easy", "dummy", max_steps=25)
run("easy dummy, max_steps=5", "easy", "dummy", max_steps=5)
print(json.dumps(res, indent=1))
