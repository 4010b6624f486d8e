"""Rollout collection + PPO loops of the reference (inversus_rl/training.py:53-201, 204-375) with
every tensor resident on the GPU: act -> step (HIP, auto-reset fused) -> store, then update.

    train_vs_dummy(num_envs, total_steps, log_dir, opponent_difficulty, load_model)
    train_selfplay(num_envs, total_steps, log_dir, load_model)

keep the reference's signatures, `steps_per_env = max(2048 // num_envs, 128)`, the terminal
transition stored before the reset, the 20 000-step self-play opponent refresh, the CSV columns
and the plain `state_dict` checkpoints (so the reference's play.py / visualize_training.py read
them).  Multi-GPU: launch with torchrun; each rank owns a contiguous env shard, gradients are
averaged with one RCCL all-reduce per optimizer step (parallel.py).
"""
import collections
import csv
import os
import time

import numpy as np
import torch

from . import _lib as L
from . import parallel, roctx
from .device_env import DeviceEnvBatch, unpack_obs
from .policies import InversusCNNPolicy
from .ppo_agent import PPOAgent

EPISODE_WINDOW = 100            # training.py:164-170: rolling means over the last 100 episodes
DIGEST_EVERY = 10               # cross-rank parameter-digest check every this many updates (and at the end)
CSV_COLUMNS = ["step", "episode", "avg_reward", "win_rate", "avg_ep_len", "policy_loss", "value_loss", "entropy"]


class TrainingLogger:
    """training_log.csv with the reference's 8 columns (training.py:16-50)."""

    def __init__(self, log_dir):
        self.log_dir = log_dir
        os.makedirs(log_dir, exist_ok=True)
        self.csv_path = os.path.join(log_dir, "training_log.csv")
        with open(self.csv_path, "w", newline="") as f:
            csv.writer(f).writerow(CSV_COLUMNS)

    def log(self, **row):
        with open(self.csv_path, "a", newline="") as f:
            csv.writer(f).writerow([row[c] for c in CSV_COLUMNS])


class RolloutCollector:
    """T vector steps of N envs without leaving the device.

    vs_dummy : a1 = policy sample, a2 = device dummy opponent (inside the step kernel).
    selfplay : one step kernel emits the P1 AND P2 observations of the new state; the next step's
               a1 / a2 come from one batched forward of `policy` / `opponent` each (the reference
               does N batch-of-1 forwards, training.py:249-258)."""

    def __init__(self, batch, agent, opponent=None, packed_obs=None, record_opponent_actions=False, opponent_stream=False, graph=False):
        self.batch, self.agent, self.opponent = batch, agent, opponent
        # graph: capture the act -> step -> store chain of a whole rollout in ONE HIP graph (training.py:117-157 is the
        # loop it replaces) and replay it per rollout: see _collect_graphed.  Needs fused actors (their packed weights are refreshed in
        # place, so the captured kernels read the updated policy) and the main stream only.
        self.graph = bool(graph)
        self._graph_state, self._graph_warm = None, False
        if self.graph and (agent.fused is None or (opponent is not None and hasattr(opponent, "parameters")) or opponent_stream):
            raise ValueError("graph=True needs fused actors (PPOAgent(fused_act=True), a FusedPolicy opponent) on one stream")
        # opponent_stream: the frozen opponent's forward (trunk, first Linear, heads, sampling) on a second HIP stream beside the
        # learner's, joined before the env step.  Safe since round 4: every kernel of both forwards is this library's own and none
        # waits on another workgroup (the library GEMM that used to sit there was a stream-K kernel sized for the whole chip, and
        # two of those at once deadlock: profiles/r04_incidents.md).  Same actions either way; off by default (measured: no gain).
        self._side = torch.cuda.Stream(device=batch.device) if (opponent_stream and opponent is not None) else None
        # record_opponent_actions: keep the a2 of every vector step (device tensors) so a rollout can be replayed elsewhere
        # (the parity tests replay sampled env columns through the oracle); off by default
        self.opponent_actions = [] if record_opponent_actions else None
        self.selfplay = batch.opponent == "selfplay"
        if self.selfplay and opponent is None:
            raise ValueError("opponent_policy required for selfplay mode")
        # packed_obs: the env writes 57-word bitstrings instead of f32 planes (INV_STEP_OBS_BITS); the fused trunk
        # reads them directly and the rollout buffer shrinks 32x.  Default: on whenever every actor is fused.
        if packed_obs is None:
            packed_obs = agent.fused is not None and (opponent is None or not hasattr(opponent, "parameters"))
        self.packed = bool(packed_obs)
        self.obs = batch.render(0, bits=self.packed)
        self.obs_p2 = batch.render(1, bits=self.packed) if self.selfplay else None
        # finished episodes: a bounded tail in completion order (vector step, then global env index -- the order of the
        # reference's single list, training.py:139-147) and a running count; nothing grows with the run
        self.tail = collections.deque(maxlen=EPISODE_WINDOW)      # rows (key, return, length, win)
        self.episodes_finished = 0
        self._serial = 0                                          # vector steps so far (the major part of the key)
        self._pending = []                                        # per-step (done, episode_return, episode_steps, info_flags)

    @torch.no_grad()
    def _opponent_actions(self):
        if hasattr(self.opponent, "eval"):
            self.opponent.eval()
        grid = self.obs_p2[0]
        if self.packed and hasattr(self.opponent, "parameters"):      # an nn.Module opponent wants f32 planes
            grid = unpack_obs(grid, self.batch.H, self.batch.W)
        logits, _ = self.opponent(grid, self.obs_p2[1])
        from .fused_policy import FusedPolicy
        if isinstance(self.opponent, FusedPolicy) and logits.is_cuda:
            from .flat_optim import sample_actions
            return sample_actions(logits, want_log_prob=False, want_i64=False, want_u8=True)[1]
        return torch.distributions.Categorical(logits=logits, validate_args=False).sample().to(torch.uint8)

    def collect(self, steps):
        with roctx.range("rollout.collect"):
            if self.graph:
                return self._collect_graphed(steps)
            return self._collect(steps)

    _AGENT_BUFFERS = ("obs_grid_buffer", "obs_extra_buffer", "action_buffer", "log_prob_buffer", "reward_buffer", "value_buffer", "done_buffer")

    def _collect_graphed(self, steps):
        """One rollout = one graph launch.  The first rollout runs eagerly (allocator, lazy initialisation); the second is CAPTURED
        (nothing executes during capture: the agent's buffers, the pending episode statistics and the recorded opponent actions are
        filled with the graph's own tensors, which every replay overwrites) and replayed; later rollouts of the same length put the
        same tensors back into the agent's (emptied) buffers and replay.  State that crosses rollouts lives at fixed addresses: the env
        batch, the current observations (copied to static buffers at the end of the graph, copied out of them at its start -- the
        rollout buffer keeps the first observation), the actors' packed weights (refreshed in place), torch's device generator
        (graph-safe Philox offsets: every replay draws new sampling keys).  A rollout of another length (the last, shorter one of a
        run) runs eagerly.  Host work per rollout: one graph launch + the drain, instead of ~10 launches per vector step."""
        agent = self.agent
        if not self._graph_warm:
            self._graph_warm = True
            return self._collect(steps)
        st = self._graph_state
        if st is not None and st["steps"] != steps:
            return self._collect(steps)
        if any(len(getattr(agent, name)) for name in self._AGENT_BUFFERS) or self._pending:
            raise RuntimeError("graph=True: collect() needs empty rollout buffers (call agent.update() or agent.reset_buffers() first)")
        if st is None:
            dev = self.batch.device
            static = [t.clone() for t in self.obs] + ([t.clone() for t in self.obs_p2] if self.selfplay else [])
            n_opp = len(self.opponent_actions) if self.opponent_actions is not None else 0
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.obs = (static[0].clone(), static[1].clone())          # the rollout buffer keeps these; the static ones are overwritten below
                if self.selfplay:
                    self.obs_p2 = (static[2].clone(), static[3].clone())
                self._collect_steps(steps)
                last = list(self.obs) + (list(self.obs_p2) if self.selfplay else [])
                for dst, src in zip(static, last):
                    dst.copy_(src)
            st = self._graph_state = {
                "steps": steps, "graph": g, "static": static,
                "buffers": {name: list(getattr(agent, name)) for name in self._AGENT_BUFFERS},
                "pending": list(self._pending),
                "opp": list(self.opponent_actions[n_opp:]) if self.opponent_actions is not None else None,
            }
            if self.opponent_actions is not None:
                del self.opponent_actions[n_opp:]
        else:
            # an eager rollout in between (another length) left the current observations in fresh tensors: the graph starts from the static ones
            cur = list(self.obs) + (list(self.obs_p2) if self.selfplay else [])
            for dst, src in zip(st["static"], cur):
                if dst is not src:
                    dst.copy_(src)
            for name in self._AGENT_BUFFERS:
                setattr(agent, name, list(st["buffers"][name]))
            self._pending = list(st["pending"])
        if self.opponent_actions is not None:
            self.opponent_actions.extend(st["opp"])
        st["graph"].replay()
        self.obs = (st["static"][0], st["static"][1])
        if self.selfplay:
            self.obs_p2 = (st["static"][2], st["static"][3])
        self._drain()
        return steps * self.batch.num_envs

    def _collect(self, steps):
        self._collect_steps(steps)
        self._drain()
        return steps * self.batch.num_envs

    def _collect_steps(self, steps):
        b, agent = self.batch, self.agent
        for _ in range(steps):
            grid, extra = self.obs
            if self.selfplay and self._side is not None:
                main = torch.cuda.current_stream(b.device)
                self._side.wait_stream(main)                       # the P2 observations come from the previous step (main stream)
                # host order = the one-stream collector's (learner first): the sampling kernels key their Philox streams by
                # draws from the device generator, which advances in call order -- same actions either way
                actions, log_probs, values, a1 = agent.act(grid, extra, with_u8=True)
                with torch.cuda.stream(self._side):
                    a2 = self._opponent_actions()
                main.wait_stream(self._side)
                a2.record_stream(main)
            else:
                actions, log_probs, values, a1 = agent.act(grid, extra, with_u8=True)
                a2 = self._opponent_actions() if self.selfplay else None
            if self.opponent_actions is not None and a2 is not None:
                self.opponent_actions.append(a2)
            # fresh obs buffers per step: the rollout buffer keeps references to them (ppo_agent.py:119)
            out = b._alloc_out((), True, self.selfplay, False, None, self.packed)
            r = b.step(a1 if a1 is not None else actions.to(torch.uint8), a2, auto_reset=True, out=out)
            agent.store_batch(grid, extra, actions, log_probs, values, r.reward, r.done)
            self.obs = (r.obs, r.extra)
            if self.selfplay:
                self.obs_p2 = (r.obs_p2, r.extra_p2)
            self._pending.append((r.done, r.episode_return, r.episode_steps, r.info_flags))

    def _drain(self):
        """Episode statistics of the steps collected since the last call: ONE gather + device->host copy per rollout
        (the per-step outputs are fresh tensors, so holding them costs nothing), not one synchronisation per vector step."""
        if not self._pending:
            return
        # the rollout's one synchronisation point: an invalid action id of ANY step since the last drain raises here like the
        # reference's ValueError (env_wrappers.py:66) -- not per vector step, which would stall the launch queue
        self.batch.check()
        done = torch.stack([p[0] for p in self._pending]).bool()                  # [T, N]
        t_idx, e_idx = done.nonzero(as_tuple=True)                                # row-major: step, then env == completion order
        T = len(self._pending)
        if t_idx.numel():
            ret = torch.stack([p[1] for p in self._pending])[t_idx, e_idx]
            length = torch.stack([p[2] for p in self._pending])[t_idx, e_idx].double()
            win = ((torch.stack([p[3] for p in self._pending])[t_idx, e_idx] & L.INV_INFO_WIN) != 0).double()
            key = (self._serial + t_idx).double() * float(2 ** 32) + (e_idx + self.batch.env_offset).double()
            rows = torch.stack([key, ret, length, win], dim=1)
            self.episodes_finished += int(rows.shape[0])
            self.tail.extend(rows[-EPISODE_WINDOW:].tolist())
        self._serial += T
        self._pending = []

    def episode_stats(self):
        """(global episode count, avg return, win rate, avg length) over the GLOBAL last-100 window (training.py:162-170),
        identical on every rank; None while no episode has finished anywhere."""
        rows, total = parallel.merge_episode_tails(list(self.tail), self.episodes_finished, EPISODE_WINDOW, self.batch.device)
        if rows.shape[0] == 0:
            return None
        m = rows[:, 1:].mean(dim=0)
        return total, float(m[0]), float(m[2]), float(m[1])


def _make(num_envs_total, opponent_type, difficulty, seed, width, height):
    rank, world_size = parallel.init_from_env()
    if world_size > 1 and num_envs_total % world_size:
        # every rank must run the same number of minibatches: each optimizer step is one gradient all-reduce
        raise ValueError(f"num_envs ({num_envs_total}) must be divisible by the number of ranks ({world_size})")
    offset, count = parallel.shard_range(num_envs_total, rank, world_size)
    base = int.from_bytes(os.urandom(7), "little") if seed is None else int(seed)
    if world_size > 1 and seed is None:
        t = torch.tensor([base], dtype=torch.int64, device="cuda")
        torch.distributed.broadcast(t, 0)
        base = int(t.item())
    if world_size > 1:
        parallel.seed_device_generator(base, rank)                  # sampling / permutation streams differ between shards
    batch = DeviceEnvBatch(count, width, height, opponent=opponent_type, difficulty=difficulty,
                           opp_seed=base, env_offset=offset)
    batch.seed(base + offset + np.arange(count, dtype=np.int64))    # seed + global env index (env_wrappers.py:463)
    batch.reset()                                                   # constructor reset (core.py:53)
    batch.reset()                                                   # env_runner.reset() (training.py:99)
    policy = InversusCNNPolicy(12, height, width, 4)
    return rank, world_size, batch, policy


def _train(mode, num_envs, total_steps, log_dir, difficulty, load_model, seed, width, height, agent_kwargs, verbose, graph=False):
    rank, world_size, batch, policy = _make(num_envs, "dummy" if mode == "vs_dummy" else "selfplay", difficulty,
                                            seed, width, height)
    if load_model:
        policy.load_state_dict(torch.load(load_model, map_location="cpu", weights_only=True))
    policy = policy.cuda()
    parallel.broadcast_parameters(policy)
    opponent = None
    if mode == "selfplay":
        opponent = InversusCNNPolicy(12, height, width, 4).cuda()
        opponent.load_state_dict(policy.state_dict())               # initial opponent is a clone (training.py:239)
        opponent.eval()
    agent = PPOAgent(policy, **(agent_kwargs or {}))
    opp_module = opponent
    if opponent is not None and agent.fused is not None:
        from .fused_policy import FusedPolicy
        opponent = FusedPolicy(opp_module, actor_only=True)         # the frozen opponent acts through the fused trunk too; no critic
    collector = RolloutCollector(batch, agent, opponent, graph=graph)
    logger = TrainingLogger(log_dir) if (log_dir and rank == 0) else None

    steps_per_env = max(2048 // num_envs, 128)                      # training.py:105-107
    step_count, episode_count, last_log, last_refresh, updates = 0, 0, 0, 0, 0
    stats, t0 = {}, time.time()
    while step_count < total_steps:
        todo = min(steps_per_env, -(-(total_steps - step_count) // num_envs))
        collector.collect(todo)                                         # (a roctx range of its own: "rollout.collect")
        step_count += todo * num_envs
        with roctx.range("ppo.update"):
            stats = agent.update()
        updates += 1
        if world_size > 1 and updates % DIGEST_EVERY == 0:
            parallel.assert_replicas_in_sync(policy)
        if mode == "selfplay" and step_count - last_refresh >= 20000:   # training.py:265,331-334
            opp_module.load_state_dict(policy.state_dict())
            if opponent is not opp_module:
                opponent.refresh()
            last_refresh = step_count
        if step_count - last_log >= 1000 or step_count >= total_steps:
            # every rank takes part: the global last-100 window and the global episode count are collectives
            es = collector.episode_stats()
            if es is not None:                                           # training.py:163: nothing is logged before the first episode ends
                last_log = step_count
                episode_count = es[0]
                row = dict(step=step_count, episode=episode_count, avg_reward=es[1], win_rate=es[2], avg_ep_len=es[3],
                           policy_loss=stats.get("policy_loss", 0.0), value_loss=stats.get("value_loss", 0.0),
                           entropy=stats.get("entropy", 0.0))
                if logger:
                    logger.log(**row)
                if verbose and rank == 0:
                    print(f"Step {step_count}/{total_steps} | Episodes: {episode_count} | Avg Reward: {row['avg_reward']:.3f} | "
                          f"Win Rate: {row['win_rate']:.3f} | Avg Ep Len: {row['avg_ep_len']:.1f} | Time: {time.time() - t0:.1f}s")
        if log_dir and rank == 0 and step_count % 50000 == 0:
            torch.save(policy.state_dict(), os.path.join(log_dir, f"policy_checkpoint_{step_count}.pt"))
    batch.check()
    if log_dir and rank == 0:
        torch.save(policy.state_dict(), os.path.join(log_dir, "policy_final.pt"))
    digest = parallel.assert_replicas_in_sync(policy)                                     # raises if the ranks diverged
    return {"steps": step_count, "episodes": episode_count, "seconds": time.time() - t0, "param_abs_sum": digest, **stats}


def train_vs_dummy(num_envs=1, total_steps=500_000, log_dir="runs/inversus_vs_dummy", opponent_difficulty="easy",
                   load_model=None, *, seed=None, width=15, height=10, agent_kwargs=None, verbose=True, graph=False):
    return _train("vs_dummy", num_envs, total_steps, log_dir, opponent_difficulty, load_model, seed, width, height,
                  agent_kwargs, verbose, graph)


def train_selfplay(num_envs=1, total_steps=500_000, log_dir="runs/inversus_selfplay", load_model=None, *,
                   seed=None, width=15, height=10, agent_kwargs=None, verbose=True, graph=False):
    return _train("selfplay", num_envs, total_steps, log_dir, "easy", load_model, seed, width, height, agent_kwargs, verbose, graph)


def main(argv=None):
    """`python -m inversus_reinforcement_learning_amd.training --mode vs_dummy --num_envs 16384 ...`:
    the reference's flags (training.py:378-407) without its 1..16 cap on --num_envs."""
    import argparse
    ap = argparse.ArgumentParser(description="Train an INVERSUS agent on the MI355X env")
    ap.add_argument("--mode", choices=["vs_dummy", "selfplay"], default="vs_dummy")
    ap.add_argument("--num_envs", type=int, default=4)
    ap.add_argument("--total_steps", type=int, default=100_000)
    ap.add_argument("--log_dir", type=str, default=None)
    ap.add_argument("--opponent_difficulty", choices=["easy", "hard"], default="easy")
    ap.add_argument("--load_model", type=str, default=None)
    ap.add_argument("--seed", type=int, default=None)
    # not in the reference: the fused MI355X paths and the PPO settings that make sense at thousands of envs
    ap.add_argument("--fused", action="store_true",
                    help="rollout forwards AND the PPO update's trunk forward/backward in the fused bf16 HIP kernels "
                         "(at thousands of envs also pass --batch_size, e.g. 16384: the reference default of 512 means "
                         "thousands of tiny optimizer steps per rollout)")
    ap.add_argument("--batch_size", type=int, default=None, help="PPO minibatch size (reference default 512)")
    ap.add_argument("--epochs", type=int, default=None, help="PPO epochs per rollout (reference default 4)")
    ap.add_argument("--lr", type=float, default=None)
    ap.add_argument("--graph", action="store_true",
                    help="with --fused: capture a whole rollout (act -> step -> store per vector step) in one HIP graph and replay it per "
                         "rollout, and a minibatch's forward + loss + backward in another -- one launch instead of ~10 per vector step / "
                         "~160 per minibatch; what small --num_envs / --batch_size runs are bound by")
    ap.add_argument("--gae_layout", choices=["per_env", "reference_flat"], default=None,
                    help="per_env (default): each env's own trajectory; reference_flat: the reference's single chain over the "
                         "env-interleaved buffer (ppo_agent.py:127-157) -- what its own runs were trained with")
    args = ap.parse_args(argv)
    log_dir = args.log_dir or f"runs/inversus_{args.mode}"
    kw = {k: v for k, v in (("batch_size", args.batch_size), ("epochs", args.epochs), ("lr", args.lr), ("gae_layout", args.gae_layout))
          if v is not None}
    if args.fused:
        kw.update(fused_act=True, fused_update=True, graph_update=bool(args.graph))
    if args.mode == "vs_dummy":
        return train_vs_dummy(args.num_envs, args.total_steps, log_dir, args.opponent_difficulty, args.load_model, seed=args.seed,
                              agent_kwargs=kw, graph=args.graph)
    return train_selfplay(args.num_envs, args.total_steps, log_dir, args.load_model, seed=args.seed, agent_kwargs=kw, graph=args.graph)


if __name__ == "__main__":
    main()
