#!/usr/bin/env python3
"""Smallest check that rocprofv3 sees this package's roctx ranges on a GPU (profiles/r05_roctx.md): HIP initialised, a few env steps
inside ranges.  Run as `rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d DIR -- python3 tools/roctx_probe.py`
(python3 directly after `--`)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd import roctx          # resolves ROCTx before torch is imported
import numpy as np, torch
from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
print("roctx:", roctx.available(), roctx.source, flush=True)
b = DeviceEnvBatch(1024, opponent="dummy"); b.seed(np.arange(1024, dtype=np.int64)); b.reset()
for t in range(6):
    with roctx.range("probe.step"):
        b.step(b.uniform_actions(1, t), None, auto_reset=True)
        torch.cuda.synchronize()
with roctx.range("probe.check"):
    b.check()
print("probe done", flush=True)
