"""Multi-GPU plumbing: one process per GPU, env shards by global index, and the only collectives
of the training loop (policy-gradient all-reduce, advantage-normalisation statistics).

The env path itself needs no collective: rank r owns global envs [r*n, (r+1)*n) and all RNG
streams are keyed by the global index.  Backend "nccl" is RCCL on
ROCm (xGMI between the 8 GPUs of a node); the same code runs on "gloo" for the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def free_port():
    """A TCP port that is free right now on 127.0.0.1 (rendezvous of a single-rank group, or of ranks this process launches)."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def seed_device_generator(base, rank):
    """Give this rank's device generator a stream of its own.  Action sampling (inv_sample_actions takes its Philox key from
    torch's device generator; the counter is the LOCAL sample index) and the minibatch permutations draw from it, and every
    process starts it from the same default seed: without this, env i of every shard would see the same uniform at every step.
    Returns the seed that was set (a function of (base, rank) only, so a run is reproducible from its --seed)."""
    seed = (int(base) * 0x9E3779B97F4A7C15 + (int(rank) + 1) * 0xD1B54A32D192ED03) & 0x7FFFFFFFFFFFFFFF
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    return seed


def initialized():
    """A process group exists (possibly of ONE rank: INVERSUS_FORCE_DIST=1 runs the collectives' code path -- communicator
    creation, all-reduce launches -- on a single GPU, which is how the RCCL branch is exercised on a 1-GPU box)."""
    return dist.is_available() and dist.is_initialized()


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun, or bench.py's own
    child ranks).  INVERSUS_FORCE_DIST=1 initialises a group even for WORLD_SIZE=1."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("INVERSUS_FORCE_DIST") == "1"
    if (ws <= 1 and not force) or dist.is_initialized():
        return world()
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        if ws > 1:
            raise RuntimeError("WORLD_SIZE > 1 needs MASTER_PORT (torchrun and bench.py's own launcher set it)")
        os.environ["MASTER_PORT"] = str(free_port())     # a forced ONE-rank group: any free port (two such runs on one host must not collide)
    if backend is None:
        # INVERSUS_DIST_BACKEND=gloo + INVERSUS_SHARE_DEVICE=1 rehearse the multi-rank path on a 1-GPU box
        backend = os.environ.get("INVERSUS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    lr = 0 if os.environ.get("INVERSUS_SHARE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(lr)
    if backend == "nccl":
        dist.init_process_group(backend, device_id=torch.device("cuda", lr))
    else:
        dist.init_process_group(backend)
    return world()


def shard_range(total_envs, rank=None, world_size=None):
    """Contiguous global env range of a rank: (offset, count); remainder spread over the first ranks."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    base, rem = divmod(int(total_envs), int(world_size))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


class FlatGradReducer:
    """Averages gradients over ranks through a flat f32 bucket per optimizer step (10 249 582 params = 41 MB at
    15x10: over xGMI a ring moves 2*(G-1)/G*41 MB per GPU).  The bucket is a persistent buffer whose slices are the
    parameters' .grad tensors, so there is no pack / unpack copy.

    `early`: parameters whose gradients are complete first in the backward pass (the MLP heads: 39.3 of the 41 MB)
    form the front of the bucket; as soon as the last of them has accumulated (post-accumulate-grad hooks) their
    all-reduce is launched asynchronously, so it travels while the convolutional trunk's backward still computes.
    reduce() waits for it and reduces the rest.  Without `early` it is ONE all-reduce after the backward pass."""

    def __init__(self, params, early=()):
        params = [p for p in params if p.requires_grad]
        early_ids = {id(p) for p in early if p.requires_grad}
        self.early = [p for p in params if id(p) in early_ids]
        self.params = self.early + [p for p in params if id(p) not in early_ids]
        n = sum(p.numel() for p in self.params)
        self.n_early = sum(p.numel() for p in self.early)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self._pending, self._work = len(self.early), None
        self._hooks = []
        # comm = False skips the collectives (measurement only: bench.py times the update with and without them to read the
        # EXPOSED all-reduce time; the replicas diverge, so never in a training run)
        self.comm = True
        if initialized():
            for p in self.early:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._early_ready))

    def close(self):
        """Remove the hooks and release the parameters' .grad aliases (an agent that drops its reducer must call this)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            p.grad = None

    def _early_ready(self, _param):
        self._pending -= 1
        if self._pending == 0 and self._work is None and self.comm:
            self._work = dist.all_reduce(self.flat[:self.n_early], op=dist.ReduceOp.SUM, async_op=True)

    def start_early(self):
        """The `early` parameters' gradients are final (written without the autograd engine, so no hook fired): launch their
        all-reduce now; reduce() waits for it."""
        if initialized() and self.comm and self.n_early and self._work is None:
            self._work = dist.all_reduce(self.flat[:self.n_early], op=dist.ReduceOp.SUM, async_op=True)

    def zero(self):
        """Zero the gradients.  Always use this instead of optimizer.zero_grad() / module.zero_grad(): those default to
        set_to_none=True, which detaches the parameters' .grad from the bucket (reduce() repairs and reports that)."""
        self.flat.zero_()
        self._pending, self._work = len(self.early), None

    def _check_aliases(self):
        off = 0
        for p in self.params:
            want = self.flat[off:off + p.numel()]
            if p.grad is None or p.grad.data_ptr() != want.data_ptr():
                raise RuntimeError("FlatGradReducer: a parameter's .grad no longer aliases the flat bucket (zero_grad(set_to_none=True)?); "
                                   "zero gradients with reducer.zero()")
            off += p.numel()

    def reduce(self, average=True):
        """All-reduce the bucket (SUM); average=True also divides by the world size (False: the caller scales, e.g. inside the
        fused Adam kernel)."""
        self._check_aliases()
        _, w = world()
        if initialized() and self.comm:
            if self._work is not None:                      # the heads' part is already travelling (or done)
                self._work.wait()
                if self.n_early < self.flat.numel():
                    dist.all_reduce(self.flat[self.n_early:], op=dist.ReduceOp.SUM)
            else:                                           # no hooks fired (no early set, or a parameter went unused)
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self._work = None
            if average:
                self.flat.div_(w)


def global_mean_std(x):
    """Mean and population std of x over ALL ranks (advantage normalisation, ppo_agent.py:173):
    one all-reduce of (sum, sum of squares, count) in f64."""
    s = torch.stack([x.double().sum(), (x.double() ** 2).sum(),
                     torch.tensor(float(x.numel()), dtype=torch.float64, device=x.device)])
    _, w = world()
    if w > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    mean = s[0] / s[2]
    var = (s[1] / s[2] - mean * mean).clamp_min(0.0)
    return mean.to(x.dtype), var.sqrt().to(x.dtype)


def broadcast_parameters(module, src=0):
    _, w = world()
    if w > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


def gather_scalars(x, device=None):
    """[x of rank 0, x of rank 1, ...] as a list of floats on every rank (one all-reduce of a one-hot vector: works on every backend)."""
    r, w = world()
    if not initialized():
        return [float(x)]
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    v = torch.zeros(w, dtype=torch.float64, device=dev)
    v[r] = float(x)
    dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return [float(t) for t in v.tolist()]


def parameter_digest(module):
    """sum |p| over all parameters in f64: identical on every rank while the replicas are in sync (the all-reduced
    gradient is bit-identical on all ranks, so are Adam's updates)."""
    return float(sum(p.detach().double().abs().sum() for p in module.parameters()))


def assert_replicas_in_sync(module):
    """Cross-rank parameter-digest check: raises if any rank's parameters differ from rank 0's."""
    d = parameter_digest(module)
    if initialized():
        all_d = gather_scalars(d, device=next(module.parameters()).device if dist.get_backend() == "nccl" else "cpu")
        if any(x != all_d[0] for x in all_d):
            raise RuntimeError(f"policy replicas diverged across ranks: parameter digests {all_d}")
    return d


def merge_episode_tails(tail, total, window=100, device="cpu"):
    """Global last-`window` episodes (training.py:162-170 keeps ONE list in completion order: vector step, then env index).
    tail: this rank's last <= window episodes as rows (key, return, length, win) with key = step_serial * 2^32 + global env
    index; total: episodes this rank has finished so far.  Returns (rows [<= window, 4] sorted by key, global total).
    One all-gather of window x 4 f64 per rank and one scalar sum -- the episode-stat reduce of SURVEY 8(e)."""
    rows = torch.full((window, 4), -1.0, dtype=torch.float64)
    if len(tail):
        t = torch.as_tensor(tail, dtype=torch.float64).reshape(-1, 4)[-window:]
        rows[:t.shape[0]] = t
    if initialized():
        _, w = world()
        dev = device if dist.get_backend() == "nccl" else "cpu"
        mine = rows.to(dev)
        parts = [torch.empty_like(mine) for _ in range(w)]
        dist.all_gather(parts, mine)
        rows = torch.cat(parts).cpu()
        cnt = torch.tensor([float(total)], dtype=torch.float64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total = int(cnt.item())
    rows = rows[rows[:, 0] >= 0]
    rows = rows[torch.argsort(rows[:, 0])][-window:]
    return rows, int(total)
