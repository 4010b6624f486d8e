#!/usr/bin/env python3
"""End-to-end training on the 32 x 32 board (BASELINE config C5) through the fused path: fused (banded) actor, fused update.
    python tools/train_c5.py [total_steps] [log_dir]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inversus_reinforcement_learning_amd import training
total = int(sys.argv[1]) if len(sys.argv) > 1 else 16384 * 128 * 8
log_dir = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/r03_c5_train"
res = training.train_vs_dummy(num_envs=16384, total_steps=total, log_dir=log_dir, opponent_difficulty="hard", seed=1, width=32, height=32,
                              agent_kwargs=dict(fused_act=True, fused_update=True, batch_size=16384, epochs=1, lr=3e-4))
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}))
