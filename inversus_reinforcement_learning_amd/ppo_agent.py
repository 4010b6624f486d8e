"""PPO agent with the reference's interface (inversus_rl/ppo_agent.py:13-249), device resident.

Same constructor arguments and defaults, `act / store_step / reset_buffers / compute_advantages /
update` with the same meaning.  What changes is where things live: observations arrive as device
tensors written by the HIP kernels, the rollout buffer is a list of [N,...] device tensors (one
entry per vector step, i.e. the reference's env-minor order k = t*N + i), GAE runs in the HIP
scan kernel (csrc/inv_gae.hip) and the minibatch loop never leaves the GPU.

GAE layout:
  * "reference_flat" reproduces the reference bit for bit: ONE chain over the env-interleaved
    buffer, so for num_envs > 1 V[k+1] is the next env at the same timestep.
  * "per_env" (default) scans each env's own trajectory -- the reference's own num_envs == 1
    behaviour applied per env.  It is what multi-GPU sharding needs (no cross-env coupling).
Deviations from the reference that cannot be bit-matched in parallel (global-RNG consumers):
Categorical sampling and the minibatch permutation use torch's device generator (argument validation of the
distribution object is off: it costs a device->host synchronisation per call, which stalls the launch queue).
"""
import torch
import torch.nn as nn
import torch.optim as optim

from . import parallel
from .device_env import gae as device_gae, unpack_obs


class PPOAgent:
    def __init__(self, policy, lr=1e-4, gamma=0.99, lam=0.95, clip_ratio=0.2, epochs=4, batch_size=512,
                 entropy_coef=0.02, value_coef=0.1, device="cuda", *, gae_layout="per_env", max_grad_norm=0.5,
                 fused_act=False, update_autocast=None, fused_update=False, graph_update=False):
        self.device = torch.device(device)
        self.policy = policy.to(self.device)
        self.gamma, self.lam, self.clip_ratio = gamma, lam, clip_ratio
        self.epochs, self.batch_size = epochs, batch_size
        self.entropy_coef, self.value_coef = entropy_coef, value_coef
        self.gae_layout, self.max_grad_norm = gae_layout, max_grad_norm
        self.optimizer = optim.Adam(self.policy.parameters(), lr=lr)
        # heads first in the bucket: their gradients are ready before the trunk's backward starts (see FlatGradReducer)
        heads = [p for name, p in self.policy.named_parameters() if name.startswith("fc_")]
        self.reducer = parallel.FlatGradReducer(self.policy.parameters(), early=heads) if parallel.initialized() else None
        # fused_act: rollouts use the fused bf16 trunk kernel (csrc/inv_policy.hip) instead of the f32
        # module; log-probs / values stored in the buffer then come from bf16 operands (~1e-2 off the f32
        # recomputation in update(), i.e. the PPO ratio starts at 1 +- 1e-2).  Default off = reference numerics.
        # update_autocast: None = f32 like the reference; torch.bfloat16 runs the minibatch forward/backward
        # under autocast (MFMA; 1.5x on the 10 M-parameter CNN), parameters and Adam state stay f32.
        self.update_autocast = update_autocast
        # fused_update: the minibatch forward / backward of the conv trunk runs in the fused HIP kernels
        # (fused_train.py; bf16 operands like update_autocast=bf16, ~4x less time than autograd over the module)
        # and the scalar tail runs in csrc/inv_optim.hip: the minibatch loss with its backward as one kernel, clip_grad_norm_ +
        # Adam over one flat parameter buffer as two launches (flat_optim.py) instead of ~100 small torch kernels.
        self.train_fwd = None
        self.flat_adam = None
        if fused_update:
            from .fused_train import FusedTrainPolicy
            from .flat_optim import FlatAdam
            self.train_fwd = FusedTrainPolicy(self.policy)
            order = self.reducer.params if self.reducer is not None else [p for p in self.policy.parameters() if p.requires_grad]
            self.flat_adam = FlatAdam(order, lr=lr, max_grad_norm=max_grad_norm,
                                      grad_flat=self.reducer.flat if self.reducer is not None else None)
            # the torch Adam built above would never be stepped: agent.optimizer becomes a handle on the optimizer that IS
            # (learning rate through param_groups, moments / step count through state_dict)
            from .flat_optim import FlatAdamHandle
            self.optimizer = FlatAdamHandle(self.flat_adam)
        # every parameter gets its gradient from the fused kernels: the update then bypasses the autograd engine (loss_and_grads)
        self.direct_grads = False
        if fused_update:
            from . import heads as HD
            from .fused_train import _trunk_params
            covered = {id(q) for q in _trunk_params(self.policy)} | {id(t) for h in HD.head_tensors(self.policy) for t in h}
            self.direct_grads = HD.supported(self.policy) and covered == {id(q) for q in self.flat_adam.params}
        # graph_update (with fused_update, single rank): a full-size minibatch's forward + loss + backward -- ~160 launches that a batch of
        # 512 is bound by -- is captured ONCE in a HIP graph and replayed per minibatch: index_select into static input buffers, one graph
        # launch, then the clip + Adam kernels (their bias corrections are host-side scalars of the step count: outside the graph)
        self.graph_update = bool(graph_update) and self.direct_grads
        self._mb_graph, self._mb_graph_warm = None, False
        self.fused = None
        if fused_act:
            from .fused_policy import FusedPolicy
            # with the fused update the rollout forward uses the update's numerics: the stored log-probs / values are
            # then what the first minibatch recomputes, and the PPO ratio starts at 1 instead of 1 +- 1e-2
            self.fused = FusedPolicy(self.policy, train_numerics=fused_update)
        self.reset_buffers()

    # ------------------------------------------------------------------ buffers (ppo_agent.py:58-66, 108-125)
    def reset_buffers(self):
        self.obs_grid_buffer, self.obs_extra_buffer = [], []
        self.action_buffer, self.log_prob_buffer = [], []
        self.reward_buffer, self.value_buffer, self.done_buffer = [], [], []

    def _t(self, x, dtype):
        return torch.as_tensor(x, dtype=dtype, device=self.device)

    @staticmethod
    def _packed(x):
        """Packed observations (int32 bitstrings written under INV_STEP_OBS_BITS) instead of f32 planes?"""
        return torch.is_tensor(x) and x.dtype == torch.int32

    def _planes(self, grid):
        """f32 planes for the autograd module, expanding packed observations (exact: entries are 0.0 / 1.0)."""
        if self._packed(grid):
            _, h, w = self.policy.norm1.normalized_shape
            return unpack_obs(grid, h, w)
        return self._t(grid, torch.float32)

    def store_step(self, grid_tensor, extra_vector, action, log_prob, value, reward, done):
        """One transition (the reference's per-env call) -- kept for API compatibility."""
        self.store_batch(self._t(grid_tensor, torch.float32)[None], self._t(extra_vector, torch.float32)[None],
                         self._t([action], torch.int64), self._t([log_prob], torch.float32),
                         self._t([value], torch.float32), self._t([reward], torch.float32), self._t([done], torch.bool))

    def store_batch(self, grid, extra, actions, log_probs, values, rewards, dones):
        """One vector step: [N,...] tensors, appended in env order (== the reference's fill order).
        `grid` may be the packed int32 form; it is stored as is (228 B instead of 7 200 B per observation)
        and expanded per minibatch in update()."""
        self.obs_grid_buffer.append(grid if self._packed(grid) else self._t(grid, torch.float32))
        self.obs_extra_buffer.append(self._t(extra, torch.float32))
        self.action_buffer.append(self._t(actions, torch.int64))
        self.log_prob_buffer.append(self._t(log_probs, torch.float32))
        self.value_buffer.append(self._t(values, torch.float32))
        self.reward_buffer.append(self._t(rewards, torch.float32))
        self.done_buffer.append(self._t(dones, torch.bool))

    # ------------------------------------------------------------------ act (ppo_agent.py:68-106)
    @torch.no_grad()
    def act(self, grid_tensors, extra_vectors, *, with_u8=False):
        """Sample actions.  NumPy in -> NumPy out (reference contract); device tensors in -> device
        tensors out (actions int64, log_probs f32, values f32).  with_u8 (device tensors only): a fourth result, the
        actions as the uint8 buffer inv_step reads (the sampling kernel writes both forms) or None."""
        as_numpy = not isinstance(grid_tensors, torch.Tensor)
        self.policy.eval()
        if self.fused is not None:                                # reads f32 planes or packed observations directly
            logits, values = self.fused(grid_tensors if self._packed(grid_tensors) else self._t(grid_tensors, torch.float32),
                                        self._t(extra_vectors, torch.float32))
        else:
            logits, values = self.policy(self._planes(grid_tensors), self._t(extra_vectors, torch.float32))
        actions_u8 = None
        if self.fused is not None and logits.is_cuda:
            # one kernel instead of torch.distributions + multinomial (~25 launches); also leaves the uint8 form inv_step reads
            from .flat_optim import sample_actions
            actions, actions_u8, log_probs = sample_actions(logits, want_u8=True)
        else:
            dist = torch.distributions.Categorical(logits=logits, validate_args=False)
            actions = dist.sample()
            log_probs = dist.log_prob(actions)
        values = values.squeeze(-1)
        if as_numpy:
            return actions.cpu().numpy(), log_probs.cpu().numpy(), values.cpu().numpy()
        return (actions, log_probs, values, actions_u8) if with_u8 else (actions, log_probs, values)

    # ------------------------------------------------------------------ GAE (ppo_agent.py:127-157)
    def _stacked(self):
        r = torch.stack([x.reshape(-1) for x in self.reward_buffer])      # [T, N]
        v = torch.stack([x.reshape(-1) for x in self.value_buffer])
        d = torch.stack([x.reshape(-1) for x in self.done_buffer])
        return r, v, d

    def _advantages_device(self, last_value=0.0):
        r, v, d = self._stacked()
        if self.device.type != "cuda":
            raise RuntimeError("GAE runs in the HIP kernel; the agent must live on the GPU")
        lv = last_value
        if self.gae_layout == "per_env" and not torch.is_tensor(lv):
            lv = torch.full((r.shape[1],), float(lv), dtype=torch.float32, device=self.device)
        adv, ret = device_gae(r, v, d, self.gamma, self.lam, lv, layout=self.gae_layout)
        return adv.reshape(-1), ret.reshape(-1)

    def compute_advantages(self, last_value=0.0):
        """-> (advantages, returns) as flat NumPy f32 arrays in buffer order, like the reference."""
        adv, ret = self._advantages_device(last_value)
        return adv.cpu().numpy(), ret.cpu().numpy()

    def _minibatch_graphed(self, fields, idx):
        """loss_and_grads of one full-size minibatch through a captured HIP graph.  fields: the rollout's (obs_grid, obs_extra, actions,
        old_log_probs, advantages, returns); idx: the minibatch's sample indices.  The first such minibatch runs eagerly (lazy
        initialisation, allocator, the side stream), the second is captured and replayed, later ones gather into the same static input
        buffers and replay.  The graph is keyed by everything that is frozen into it (shapes, dtypes, the three loss coefficients);
        the gradients land in the flat bucket and `means` in a static tensor, exactly as in the eager call."""
        key = (tuple((tuple(t.shape[1:]), t.dtype) for t in fields), int(idx.numel()), float(self.clip_ratio), float(self.value_coef),
               float(self.entropy_coef))
        st = self._mb_graph
        if st is None or st["key"] != key:
            if not self._mb_graph_warm:
                self._mb_graph_warm = True
                return self.train_fwd.loss_and_grads(*[t[idx] for t in fields], self.clip_ratio, self.value_coef, self.entropy_coef)
            static = [torch.empty((idx.numel(),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in fields]
            for dst, src in zip(static, fields):
                torch.index_select(src, 0, idx, out=dst)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                means = self.train_fwd.loss_and_grads(*static, self.clip_ratio, self.value_coef, self.entropy_coef)
            st = self._mb_graph = {"key": key, "graph": g, "static": static, "means": means}
        else:
            for dst, src in zip(st["static"], fields):
                torch.index_select(src, 0, idx, out=dst)
        st["graph"].replay()
        return st["means"]

    # ------------------------------------------------------------------ update (ppo_agent.py:159-247)
    def update(self, advantages=None, returns=None, *, permutations=None, trace=False):
        """PPO update over the stored rollout.  `advantages` / `returns` (flat, buffer order) may be
        supplied by the caller; by default they come from the HIP GAE kernel.
        permutations: one index permutation of the T*N samples per epoch (what the reference's
        `np.random.shuffle(indices)` leaves in `indices`, ppo_agent.py:195) -- the injection hook of the parity
        fixture tests/golden/g8_ppo_update.npz; default: torch.randperm on the device.
        trace: also return the per-minibatch policy / value loss, entropy and pre-clip gradient norm
        (`stats["minibatches"]`, f64 arrays) and the normalised advantages (`stats["advantages"]`)."""
        if len(self.obs_grid_buffer) == 0:
            return {}
        from . import roctx
        if advantages is None:
            with roctx.range("ppo.gae"):
                advantages, returns = self._advantages_device()
        else:
            advantages, returns = self._t(advantages, torch.float32).reshape(-1), self._t(returns, torch.float32).reshape(-1)
        mean, std = parallel.global_mean_std(advantages)                  # population std, all ranks (:173)
        advantages = (advantages - mean) / (std + 1e-8)

        obs_grid = torch.cat(self.obs_grid_buffer)
        obs_extra = torch.cat(self.obs_extra_buffer)
        actions = torch.cat(self.action_buffer)
        old_log_probs = torch.cat(self.log_prob_buffer)

        self.policy.train()
        tot_p = torch.zeros((), device=self.device)
        tot_v = torch.zeros((), device=self.device)
        tot_e = torch.zeros((), device=self.device)
        num_samples = obs_grid.shape[0]
        num_updates = 0
        params = [p for p in self.policy.parameters() if p.requires_grad]
        if permutations is not None and len(permutations) != self.epochs:
            raise ValueError(f"permutations: expected {self.epochs} (one per epoch), got {len(permutations)}")
        mb_trace = []
        for epoch in range(self.epochs):
            if permutations is None:
                indices = torch.randperm(num_samples, device=self.device)
            else:
                indices = torch.as_tensor(permutations[epoch], dtype=torch.int64, device=self.device).reshape(-1)
                if indices.numel() != num_samples:
                    raise ValueError(f"permutations[{epoch}] must index all {num_samples} samples")
            for start in range(0, num_samples, self.batch_size):
                idx = indices[start:start + self.batch_size]
                if self.direct_grads:
                    with roctx.range("ppo.minibatch (fused forward + backward + clip/Adam)"):
                        # fused trunk / heads / loss kernels forward and backward without the autograd engine; every gradient is
                        # written into the flat bucket (nothing to zero), the heads' all-reduce starts before the trunk's backward
                        if self.graph_update and self.reducer is None and not trace and idx.numel() == self.batch_size:
                            means = self._minibatch_graphed((obs_grid, obs_extra, actions, old_log_probs, advantages, returns), idx)
                        else:
                            means = self.train_fwd.loss_and_grads(obs_grid[idx], obs_extra[idx], actions[idx], old_log_probs[idx],
                                                                  advantages[idx], returns[idx], self.clip_ratio, self.value_coef,
                                                                  self.entropy_coef,
                                                                  after_heads=self.reducer.start_early if self.reducer is not None else None)
                        if self.reducer is not None:
                            self.reducer.reduce(average=False)            # SUM over ranks; the 1/world_size rides in the Adam kernel
                            grad_norm = self.flat_adam.step(grad_scale=1.0 / parallel.world()[1])
                        else:
                            grad_norm = self.flat_adam.step()
                    policy_loss, value_loss, entropy = means[0], means[1], means[2]
                    if trace:
                        mb_trace.append(torch.stack([policy_loss, value_loss, entropy, grad_norm.detach()]))
                    tot_p += policy_loss
                    tot_v += value_loss
                    tot_e += entropy
                    num_updates += 1
                    continue
                if self.train_fwd is not None:                            # fused trunk kernels; reads packed observations as is
                    logits, values = self.train_fwd(obs_grid[idx], obs_extra[idx])
                else:
                    with torch.autocast(self.device.type, dtype=self.update_autocast or torch.bfloat16,
                                        enabled=self.update_autocast is not None):
                        logits, values = self.policy(self._planes(obs_grid[idx]), obs_extra[idx])
                logits, values = logits.float(), values.float()
                if self.flat_adam is not None:                            # HIP loss (+ its backward) and clip + Adam kernels
                    from .flat_optim import ppo_loss
                    loss, means = ppo_loss(logits, values.squeeze(-1), actions[idx], old_log_probs[idx], advantages[idx], returns[idx],
                                           self.clip_ratio, self.value_coef, self.entropy_coef)
                    if self.reducer is not None:
                        self.reducer.zero()
                        loss.backward()
                        self.reducer.reduce(average=False)                # SUM over ranks; the 1/world_size rides in the Adam kernel
                        grad_norm = self.flat_adam.step(grad_scale=1.0 / parallel.world()[1])
                    else:
                        self.flat_adam.zero_grad()
                        loss.backward()
                        grad_norm = self.flat_adam.step()
                    policy_loss, value_loss, entropy = means[0], means[1], means[2]
                else:
                    dist = torch.distributions.Categorical(logits=logits, validate_args=False)
                    new_log_probs = dist.log_prob(actions[idx])
                    entropy = dist.entropy().mean()
                    ratio = torch.exp(new_log_probs - old_log_probs[idx])
                    adv = advantages[idx]
                    surr1 = ratio * adv
                    surr2 = torch.clamp(ratio, 1.0 - self.clip_ratio, 1.0 + self.clip_ratio) * adv
                    policy_loss = -torch.min(surr1, surr2).mean()
                    value_loss = nn.functional.mse_loss(values.squeeze(-1), returns[idx])
                    loss = policy_loss + self.value_coef * value_loss - self.entropy_coef * entropy
                    if self.reducer is not None:
                        self.reducer.zero()                               # grads accumulate into the flat bucket
                        loss.backward()
                        self.reducer.reduce()                             # ONE RCCL all-reduce per optimizer step
                    else:
                        self.optimizer.zero_grad()
                        loss.backward()
                    grad_norm = torch.nn.utils.clip_grad_norm_(params, self.max_grad_norm)
                    self.optimizer.step()
                if trace:
                    mb_trace.append(torch.stack([policy_loss.detach(), value_loss.detach(), entropy.detach(), grad_norm.detach()]))
                tot_p += policy_loss.detach()
                tot_v += value_loss.detach()
                tot_e += entropy.detach()
                num_updates += 1
        self.reset_buffers()
        if self.fused is not None:
            self.fused.refresh()                                          # re-pack the updated weights for the actor
        stats = {"policy_loss": float(tot_p) / num_updates, "value_loss": float(tot_v) / num_updates,
                 "entropy": float(tot_e) / num_updates}
        if trace:
            m = torch.stack(mb_trace).double().cpu().numpy()
            stats["minibatches"] = {"policy_loss": m[:, 0], "value_loss": m[:, 1], "entropy": m[:, 2], "grad_norm": m[:, 3]}
            stats["advantages"] = advantages.detach().cpu().numpy()
        return stats
