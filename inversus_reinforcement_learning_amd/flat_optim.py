"""The scalar tail of the PPO update in HIP (csrc/inv_optim.hip):

  * `ppo_loss(logits, values, actions, old_log_probs, advantages, returns, ...)` -- the reference's minibatch loss
    (inversus_rl/ppo_agent.py:211-224) as ONE kernel that also produces d loss / d logits and d loss / d values; a
    `torch.autograd.Function`, so the two MLP heads upstream keep using autograd.
  * `FlatAdam(params)` -- all parameters re-homed into ONE flat f32 buffer (each `p.data` becomes a view of it, gradients
    likewise -- the same bucket `parallel.FlatGradReducer` all-reduces), Adam moments next to it, and
    `clip_grad_norm_(0.5)` + `Adam.step()` (ppo_agent.py:227-232) as two launches without a host round trip.

Used by `PPOAgent(fused_update=True)`; the default agent keeps torch's loss / clip / Adam (pinned to the reference by
tests/golden/g8_ppo_update.npz).  Both kernels are checked against the torch expressions they replace to 1e-6
(tests/test_gpu_flat_optim.py)."""
import ctypes as C

import torch

from . import _lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _PPOLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, values, actions, old_log_probs, advantages, returns, clip_ratio, value_coef, entropy_coef):
        lib = L.load()
        logits_c, values_c = logits.detach().float().contiguous(), values.detach().float().reshape(-1).contiguous()
        b = logits_c.shape[0]
        if logits_c.shape != (b, 13) or values_c.shape != (b,):
            raise ValueError(f"ppo_loss expects logits [B, 13] and values [B]; got {tuple(logits.shape)}, {tuple(values.shape)}")
        dlogits, dvalues = torch.empty_like(logits_c), torch.empty_like(values_c)
        nb = lib.inv_ppo_loss_blocks(b)
        partial = torch.empty((nb, 3), dtype=torch.float32, device=logits_c.device)
        with torch.cuda.device(logits_c.device):
            rc = lib.inv_ppo_loss(_ptr(logits_c), _ptr(values_c), _ptr(actions.contiguous()), _ptr(old_log_probs.float().contiguous()),
                                  _ptr(advantages.float().contiguous()), _ptr(returns.float().contiguous()), b, float(clip_ratio),
                                  float(value_coef), float(entropy_coef), _ptr(dlogits), _ptr(dvalues), _ptr(partial), _stream())
        L.raise_for(rc, lib)
        means = partial.sum(dim=0) / b                                   # policy loss, value loss, entropy
        ctx.save_for_backward(dlogits, dvalues)
        ctx.values_shape = values.shape
        loss = means[0] + value_coef * means[1] - entropy_coef * means[2]
        ctx.mark_non_differentiable(means)
        return loss, means

    @staticmethod
    def backward(ctx, g_loss, _g_means):
        dlogits, dvalues = ctx.saved_tensors
        return g_loss * dlogits, (g_loss * dvalues).reshape(ctx.values_shape), None, None, None, None, None, None, None


def ppo_loss(logits, values, actions, old_log_probs, advantages, returns, clip_ratio=0.2, value_coef=0.1, entropy_coef=0.02):
    """-> (loss, means) with means = [policy_loss, value_loss, entropy] (detached f32[3]); differentiable in logits / values."""
    if actions.dtype != torch.int64:
        actions = actions.to(torch.int64)
    return _PPOLossFn.apply(logits, values, actions, old_log_probs, advantages, returns, clip_ratio, value_coef, entropy_coef)


def sample_actions(logits, want_log_prob=True, want_i64=True, want_u8=False):
    """Categorical(logits=logits).sample() [+ .log_prob] as one kernel (csrc/inv_optim.hip: k_sample_actions).  The Philox key comes
    from torch's device generator (one tiny launch), so torch.manual_seed controls the stream.
    -> (actions int64 [B] | None, actions uint8 [B] | None, log_prob f32 [B] | None)."""
    lib = L.load()
    logits = logits.detach().float().contiguous()
    b, dev = logits.shape[0], logits.device
    if logits.dim() != 2 or logits.shape[1] != 13:
        raise ValueError(f"sample_actions expects logits [B, 13], got {tuple(logits.shape)}")
    seed = torch.randint(-2 ** 62, 2 ** 62, (2,), dtype=torch.int64, device=dev)
    a64 = torch.empty(b, dtype=torch.int64, device=dev) if want_i64 else None
    a8 = torch.empty(b, dtype=torch.uint8, device=dev) if want_u8 else None
    lp = torch.empty(b, dtype=torch.float32, device=dev) if want_log_prob else None
    with torch.cuda.device(dev):
        rc = lib.inv_sample_actions(_ptr(logits), b, _ptr(seed), _ptr(a64) if a64 is not None else None, _ptr(a8) if a8 is not None else None,
                                    _ptr(lp) if lp is not None else None, _stream())
    L.raise_for(rc, lib)
    return a64, a8, lp


class FlatAdam:
    """clip_grad_norm_ + Adam over one flat buffer (torch.optim.Adam defaults of the reference: betas (0.9, 0.999), eps 1e-8,
    no weight decay).  `grad_flat`: an existing flat gradient bucket whose slices already are the parameters' .grad in
    `params` order (parallel.FlatGradReducer.flat); otherwise one is created here."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=0.5, grad_flat=None):
        self.lib = L.load()
        self.params = [p for p in params if p.requires_grad]
        if not self.params or any(p.dtype != torch.float32 or not p.is_cuda for p in self.params):
            raise ValueError("FlatAdam needs float32 parameters on the GPU")
        self.lr, self.betas, self.eps, self.max_grad_norm = float(lr), betas, float(eps), float(max_grad_norm)
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.n = n
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:                                       # re-home the parameter: p.data becomes a view of the flat buffer
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + k].view_as(p)
                off += k
        if grad_flat is None:
            grad_flat = torch.zeros(n, dtype=torch.float32, device=dev)
            off = 0
            for p in self.params:
                p.grad = grad_flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        elif grad_flat.numel() != n:
            raise ValueError("grad_flat does not match the parameters")
        self.grad = grad_flat
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.workspace = torch.zeros(int(self.lib.inv_clip_adam_workspace_bytes()), dtype=torch.uint8, device=dev)
        self.steps = 0
        self.lr_source = None                                           # FlatAdamHandle's param group, once a handle exists

    def zero_grad(self):
        self.grad.zero_()

    def _check_aliases(self):
        off = 0
        for p in self.params:
            k = p.numel()
            if p.data_ptr() != self.flat.data_ptr() + 4 * off or p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                raise RuntimeError("FlatAdam: a parameter or its .grad no longer aliases the flat buffers "
                                   "(module.to() / zero_grad(set_to_none=True) after construction?)")
            off += k

    def step(self, grad_scale=1.0):
        """One optimizer step from the gradients in the flat bucket.  -> 0-dim f32 tensor: the pre-clip global gradient norm."""
        self._check_aliases()
        self.steps += 1
        if self.lr_source is not None:
            self.lr = float(self.lr_source["lr"])
        with torch.cuda.device(self.flat.device):
            rc = self.lib.inv_clip_adam_step(_ptr(self.flat), _ptr(self.grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq), self.n, self.lr,
                                             float(self.betas[0]), float(self.betas[1]), self.eps, self.steps, self.max_grad_norm,
                                             float(grad_scale), _ptr(self.workspace), _stream())
        L.raise_for(rc, self.lib)
        return self.workspace[-8:-4].view(torch.float32)[0]


class FlatAdamHandle:
    """What `agent.optimizer` is while the fused update runs: the torch.optim surface that code around a PPO agent actually
    touches -- `param_groups[0]["lr"]` (read / write: it IS FlatAdam's learning rate), `state_dict()` / `load_state_dict()`
    (Adam's moments and step count, so a checkpoint of the optimizer is a checkpoint of the optimizer that steps) and
    `zero_grad()` -- over FlatAdam's flat buffers.  There is no `step()`: the update calls FlatAdam.step() itself, and an
    accidental second optimizer stepping the same parameters would be a silent bug."""

    class _Group(dict):
        """A real dict: "lr" is an ordinary key (so `in`, keys(), items(), dict(group), copy(), setdefault see it), and every write of
        it is pushed into FlatAdam, which reads its learning rate back from here at each step -- the dict is the single source."""

        def __init__(self, adam):
            super().__init__(params=adam.params, lr=adam.lr, betas=adam.betas, eps=adam.eps, weight_decay=0.0)
            self._adam = adam

        def __setitem__(self, k, v):
            if k == "lr":
                v = float(v)
                self._adam.lr = v
            super().__setitem__(k, v)

        def update(self, *a, **kw):
            for k, v in dict(*a, **kw).items():
                self[k] = v

        def setdefault(self, k, default=None):
            if k not in self:
                self[k] = default
            return self[k]

    def __init__(self, adam):
        self.adam = adam
        self.param_groups = [self._Group(adam)]
        adam.lr_source = self.param_groups[0]                           # FlatAdam.step() takes param_groups[0]["lr"]

    def zero_grad(self, set_to_none=False):
        self.adam.zero_grad()                                            # never to None: the gradients alias the flat bucket

    def state_dict(self):
        a = self.adam
        return {"kind": "FlatAdam", "lr": float(self.param_groups[0]["lr"]), "betas": tuple(a.betas), "eps": a.eps, "max_grad_norm": a.max_grad_norm,
                "steps": a.steps, "exp_avg": a.exp_avg.detach().clone(), "exp_avg_sq": a.exp_avg_sq.detach().clone(),
                "param_shapes": [tuple(p.shape) for p in a.params]}       # the flat buffers mean nothing without the order they were cut in

    def load_state_dict(self, sd):
        a = self.adam
        if sd.get("kind") != "FlatAdam" or sd["exp_avg"].numel() != a.n:
            raise ValueError("not a FlatAdam state of this parameter set")
        shapes = sd.get("param_shapes")
        if shapes is not None and [tuple(x) for x in shapes] != [tuple(p.shape) for p in a.params]:
            raise ValueError("FlatAdam state was saved for parameters of other shapes / another order: the flat moments would land on the wrong weights")
        self.param_groups[0]["lr"] = float(sd["lr"])
        a.betas, a.eps, a.max_grad_norm = tuple(sd["betas"]), float(sd["eps"]), float(sd["max_grad_norm"])
        a.steps = int(sd["steps"])
        a.exp_avg.copy_(sd["exp_avg"].to(a.exp_avg.device))
        a.exp_avg_sq.copy_(sd["exp_avg_sq"].to(a.exp_avg_sq.device))

    def step(self, *a, **k):
        raise RuntimeError("the fused update steps FlatAdam inside PPOAgent.update(); agent.optimizer.step() would step the parameters twice")
