"""Worker of tests/test_policy_dist_cpu.py: one of two gloo ranks on CPU.

Checks that the N>1 training path is correct by construction: each rank holds half of a
rollout; after ONE full-batch PPO step with the flat gradient all-reduce and the global
advantage statistics, both ranks hold identical parameters, equal (to f32 round-off) to a
single-process step over the whole rollout."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from inversus_reinforcement_learning_amd import parallel
from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent


class TinyPolicy(nn.Module):
    def __init__(self):
        super().__init__()
        self.body = nn.Linear(12 * 3 * 3 + 4, 32)
        self.fc_pi = nn.Linear(32, 13)          # "fc_" parameters = the early part of the gradient bucket (parallel.py)
        self.fc_v = nn.Linear(32, 1)

    def forward(self, grid, extra):
        h = torch.tanh(self.body(torch.cat([grid.flatten(1), extra], 1)))
        return self.fc_pi(h), self.fc_v(h)


def make_rollout(T, N, seed):
    g = torch.Generator().manual_seed(seed)
    grid = (torch.rand(T, N, 12, 3, 3, generator=g) < 0.3).float()
    extra = torch.rand(T, N, 4, generator=g)
    actions = torch.randint(0, 13, (T, N), generator=g)
    logp = -torch.rand(T, N, generator=g) - 2.0
    values = torch.randn(T, N, generator=g)
    rewards = torch.randn(T, N, generator=g)
    dones = torch.rand(T, N, generator=g) < 0.1
    adv = torch.randn(T, N, generator=g)          # injected advantages / returns (GAE itself is a GPU kernel)
    ret = adv + values
    return grid, extra, actions, logp, values, rewards, dones, adv, ret


def run(agent, data, cols):
    grid, extra, actions, logp, values, rewards, dones, adv, ret = (x[:, cols] for x in data)
    for t in range(grid.shape[0]):
        agent.store_batch(grid[t], extra[t], actions[t], logp[t], values[t], rewards[t], dones[t])
    return agent.update(advantages=adv.reshape(-1), returns=ret.reshape(-1))


def main():
    rank, world = parallel.init_from_env("gloo")
    assert world == 2
    T, N = 6, 8
    data = make_rollout(T, N, 5)

    # sharding rule
    off, cnt = parallel.shard_range(N)
    assert (off, cnt) == (rank * 4, 4)

    # global statistics == statistics of the concatenation
    x = data[7]
    m, s = parallel.global_mean_std(x[:, off:off + cnt].reshape(-1))
    assert abs(float(m) - float(x.mean())) < 1e-6 and abs(float(s) - float(x.std(unbiased=False))) < 1e-6

    # distributed one-step update (each rank: its env columns) vs single-process over all columns
    torch.manual_seed(1)
    pol_d = TinyPolicy()
    parallel.broadcast_parameters(pol_d)
    ref_state = {k: v.clone() for k, v in pol_d.state_dict().items()}
    kw = dict(lr=1e-2, epochs=1, batch_size=10 ** 6, device="cpu")
    agent_d = PPOAgent(pol_d, **kw)
    # plain SGD makes the parameter delta linear in the gradient (Adam's first step is ~sign(g), which
    # amplifies f32 round-off of near-zero gradients)
    agent_d.optimizer = torch.optim.SGD(pol_d.parameters(), lr=0.05)
    assert agent_d.reducer is not None and agent_d.reducer.flat.numel() == sum(p.numel() for p in pol_d.parameters())
    run(agent_d, data, slice(off, off + cnt))
    flat = torch.cat([p.detach().reshape(-1) for p in pol_d.parameters()])
    other = flat.clone()
    dist.broadcast(other, src=0)
    assert torch.equal(flat, other), "ranks diverged"

    pol_s = TinyPolicy()
    pol_s.load_state_dict(ref_state)
    agent_s = PPOAgent(pol_s, **kw)
    agent_s.optimizer = torch.optim.SGD(pol_s.parameters(), lr=0.05)
    agent_s.reducer.close()                        # single-process reference on the whole rollout: no hooks, no bucket
    agent_s.reducer = None
    # PPOAgent normalises with the all-rank statistics; feed it pre-normalised-equivalent data by running on all columns:
    # with world == 2 global_mean_std all-reduces, so give each rank the full data scaled to keep the sums equal
    grid, extra, actions, logp, values, rewards, dones, adv, ret = data
    for t in range(T):
        agent_s.store_batch(grid[t], extra[t], actions[t], logp[t], values[t], rewards[t], dones[t])
    # (sum, sumsq, count) all double across the two identical ranks -> same mean / std
    agent_s.update(advantages=adv.reshape(-1), returns=ret.reshape(-1))
    flat_s = torch.cat([p.detach().reshape(-1) for p in pol_s.parameters()])
    err = float((flat - flat_s).abs().max())
    assert err < 5e-6, f"distributed step differs from the single-process step: {err}"
    # episode statistics: the global last-100 window in completion order (vector step, then global env index) and the
    # global count, from per-rank tails (training.py:162-170 keeps ONE list; SURVEY 8e "episode-stat reduce")
    rng = np.random.RandomState(3)
    n_ep = 260
    steps = np.sort(rng.randint(0, 40, n_ep))
    envs = rng.randint(0, 8, n_ep)
    keys = steps.astype(np.float64) * 2.0 ** 32 + envs
    rows = np.stack([keys, rng.randn(n_ep), rng.randint(1, 500, n_ep).astype(np.float64), rng.randint(0, 2, n_ep).astype(np.float64)], 1)
    rows = rows[np.unique(keys, return_index=True)[1]]                     # one episode per (step, env)
    mine = rows[(rows[:, 0] % 2 ** 32 >= off) & (rows[:, 0] % 2 ** 32 < off + cnt)]
    merged, total = parallel.merge_episode_tails(mine[-100:].tolist(), len(mine), 100)
    assert total == len(rows), (total, len(rows))
    assert np.array_equal(merged.numpy(), rows[-100:]), "global last-100 window differs from the single-list window"

    # gradient bucket hygiene: zero_grad(set_to_none=True) detaches .grad from the bucket -> reduce() must refuse
    agent_d.optimizer.zero_grad()
    try:
        agent_d.reducer.reduce()
        raise SystemExit("reduce() accepted detached gradients")
    except RuntimeError:
        pass
    digests = parallel.gather_scalars(parallel.parameter_digest(pol_d))
    assert len(digests) == 2 and digests[0] == digests[1]
    parallel.assert_replicas_in_sync(pol_d)
    with torch.no_grad():
        if rank == 1:
            next(pol_d.parameters()).add_(1e-3)                            # a diverged replica must be caught on every rank
    try:
        parallel.assert_replicas_in_sync(pol_d)
        raise SystemExit("diverged replicas not detected")
    except RuntimeError:
        pass
    dist.barrier()
    print("DIST_OK", rank, err)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
