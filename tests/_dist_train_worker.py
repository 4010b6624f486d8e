"""Worker of test_gpu_training.py::test_two_rank_training_rehearsal: one of two ranks sharing the GPU (gloo)."""
import hashlib
import os
import torch
from inversus_reinforcement_learning_amd import parallel, training

res = training.train_selfplay(num_envs=64, total_steps=64 * 128, log_dir=None, seed=11, verbose=False,
                              agent_kwargs=dict(batch_size=1024, epochs=1, **(dict(fused_act=True, fused_update=True)
                                                                              if os.environ.get("INVERSUS_TEST_FUSED") == "1" else {})))
rank, world = parallel.world()
assert world == 2
# both ranks must hold identical parameters after the synchronised updates
import torch.distributed as dist
mine = torch.tensor([res["param_abs_sum"]], dtype=torch.float64)
other = mine.clone()
dist.broadcast(other, src=0)
assert float(other) == float(mine), (float(other), float(mine))
print("TRAIN_OK", rank, res["steps"], res["param_abs_sum"], flush=True)
dist.barrier()
dist.destroy_process_group()
