#!/usr/bin/env python3
"""Learning sanity run: PPO vs the easy dummy on the device env (not a benchmark)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inversus_reinforcement_learning_amd import training
torch.manual_seed(0)
n = int(os.environ.get("N", 1024)); updates = int(os.environ.get("UPDATES", 10))
t0 = time.time()
res = training.train_vs_dummy(num_envs=n, total_steps=n * 128 * updates, log_dir=os.environ.get("LOGDIR", "gpurun_out/train_sanity"), opponent_difficulty="easy",
                              seed=0, verbose=True, agent_kwargs=dict(batch_size=int(os.environ.get("BATCH", 4096)), epochs=4, lr=float(os.environ.get("LR", 1e-4)),
                                                fused_act=bool(int(os.environ.get("FUSED", "1"))),
                                                fused_update=bool(int(os.environ.get("FUSED_UPDATE", "0")))))
print(json.dumps(res), "wall", time.time() - t0)
print(open(os.path.join(os.environ.get("LOGDIR", "gpurun_out/train_sanity"), "training_log.csv")).read())
