#!/usr/bin/env python3
"""Drop-in mode rates: MultiEnvRunner with the reference's NumPy contract (observations copied to the host every step,
info dicts built) at N = 4 (BASELINE config C1) and N = 16384, vs the device-resident mode.  vs_dummy easy."""
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd.env_wrappers import MultiEnvRunner
res = {}
for n, steps in ((4, 300), (16384, 20)):
    for output in ("numpy", "torch"):
        r = MultiEnvRunner(num_envs=n, opponent_type="dummy", difficulty="easy", seed=0, output=output, auto_reset=True)
        r.reset()
        rng = np.random.RandomState(0)
        acts = [rng.randint(0, 13, n) for _ in range(8)]
        for k in range(3): r.step(acts[k])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(steps): r.step(acts[k % 8])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[f"N={n} output={output}"] = {"env_steps_per_s": round(n * steps / dt), "ms_per_vector_step": round(dt / steps * 1e3, 3)}
print(json.dumps(res, indent=1))
