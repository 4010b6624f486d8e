"""DeviceEnvBatch: N Inversus envs resident in HBM, stepped through the C ABI.

Thin host plumbing only: torch supplies device buffers and the current HIP stream; every
computation happens in libinversus_hip.so (csrc/).  All methods are asynchronous on the
current torch stream unless they return host data.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class StepResult:
    """Device tensors of one vector step (or of T steps for rollout(); leading dim T)."""
    __slots__ = ("obs", "extra", "obs_p2", "extra_p2", "reward", "reward64", "done", "info_flags",
                 "episode_steps", "episode_return")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


class DeviceEnvBatch:
    def __init__(self, num_envs, width=15, height=10, max_bullets=16, max_episode_steps=500,
                 opponent="dummy", difficulty="easy", device=None, opp_seed=0, env_offset=0, step_kernel=None, obs_store=None):
        """step_kernel: "auto" (default; env INVERSUS_STEP_KERNEL overrides the default), "wave" or "row4" -- which kernel runs
        full steps (inv_set_step_kernel); obs_store: "auto" (default; env INVERSUS_OBS_STORE) or "plain" -- the cache policy of
        the f32 observation stores (inv_set_obs_store_policy).  Results are identical either way."""
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise RuntimeError("inversus_reinforcement_learning_amd needs a ROCm GPU (torch.cuda is unavailable); "
                               "there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device in (None, "cuda") else torch.device(device)
        if opponent not in ("dummy", "selfplay"):
            raise ValueError(f"Unknown opponent_type: {opponent}")            # env_wrappers.py:316
        if difficulty not in ("easy", "hard"):
            difficulty = "hard"                                             # env_wrappers.py:81-89: anything else is the hard branch
        self.num_envs, self.W, self.H, self.K = int(num_envs), int(width), int(height), int(max_bullets)
        self.GW = (self.W * self.H + 31) // 32
        self.OW = (12 * self.W * self.H + 31) // 32          # u32 words of one packed observation
        self.opponent, self.difficulty = opponent, difficulty
        self.max_episode_steps = int(max_episode_steps)
        self.env_offset = int(env_offset)
        cfg = L.InvConfig(self.num_envs, self.W, self.H, self.K, self.max_episode_steps,
                          L.INV_OPP_DUMMY if opponent == "dummy" else L.INV_OPP_EXTERNAL,
                          L.INV_HARD if difficulty == "hard" else L.INV_EASY,
                          self.device.index or 0, int(opp_seed) & 0xFFFFFFFF, self.env_offset)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.raise_for(self.lib.inv_create(C.byref(cfg), C.byref(h)), self.lib)
        self.h = h
        kind = (step_kernel or os.environ.get("INVERSUS_STEP_KERNEL") or "auto").lower()
        if kind == "row4_if_any":                                       # the parity tests: force the four-env kernel where it exists
            kind = "row4" if (self.K == 16 and (self.W, self.H) in ((15, 10), (32, 32))) else "auto"
        if kind not in ("auto", "wave", "row4"):
            raise ValueError(f"step_kernel must be auto, wave or row4, got {kind!r}")
        self.step_kernel = kind
        if kind != "auto":
            code = {"wave": L.INV_STEP_KERNEL_WAVE, "row4": L.INV_STEP_KERNEL_ROW4}[kind]
            L.raise_for(self.lib.inv_set_step_kernel(self.h, code), self.lib)
        store = (obs_store or os.environ.get("INVERSUS_OBS_STORE") or "auto").lower()
        if store not in ("auto", "plain"):
            raise ValueError(f"obs_store must be auto or plain, got {store!r}")
        self.obs_store = store
        if store == "plain":
            L.raise_for(self.lib.inv_set_obs_store_policy(self.h, L.INV_OBS_STORE_PLAIN), self.lib)

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                self.lib.inv_destroy(h)
            except Exception:
                pass

    # ------------------------------------------------------------------ buffers
    def _new(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def obs_shape(self, lead=()):
        return tuple(lead) + (self.num_envs, 12, self.H, self.W)

    def _new_obs(self, lead, bits):
        """f32 planes [.., N, 12, H, W], or the packed form [.., N, OW] int32 (INV_STEP_OBS_BITS)."""
        if bits:
            return self._new(tuple(lead) + (self.num_envs, self.OW), torch.int32)
        return self._new(self.obs_shape(lead), torch.float32)

    def _is_bits(self, r):
        """Packed or f32 output buffers?  (both perspectives must agree: the flag is per launch)"""
        kinds = {t.dtype for t in (r.obs, r.obs_p2) if t is not None}
        if len(kinds) > 1 or not kinds <= {torch.float32, torch.int32}:
            raise ValueError("obs / obs_p2 buffers must both be float32 planes or both int32 packed bitstrings")
        bits = kinds == {torch.int32}
        for t in (r.obs, r.obs_p2):
            if t is not None and tuple(t.shape[-1:] if bits else t.shape[-3:]) != ((self.OW,) if bits else (12, self.H, self.W)):
                raise ValueError(f"observation buffer has the wrong trailing shape: {tuple(t.shape)}")
        return bits

    def _as_actions(self, a, lead=()):
        if a is None:
            return None
        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=torch.uint8).contiguous()
        else:
            arr = np.asarray(a)
            if arr.size and (arr.min() < 0 or arr.max() > 12):
                bad = int(arr.reshape(-1)[(arr.reshape(-1) < 0) | (arr.reshape(-1) > 12)][0])
                raise ValueError(f"Invalid action_id: {bad}, must be 0-12")  # env_wrappers.py:66
            t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint8)).to(self.device)
        if tuple(t.shape) != tuple(lead) + (self.num_envs,):
            raise ValueError(f"actions must have shape {tuple(lead) + (self.num_envs,)}, got {tuple(t.shape)}")
        return t

    # ------------------------------------------------------------------ control
    def seed(self, seeds, mask=None):
        s = np.ascontiguousarray(seeds, dtype=np.int64)
        assert s.shape == (self.num_envs,)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        with torch.cuda.device(self.device):
            L.raise_for(self.lib.inv_seed_masked(self.h, s.ctypes.data_as(C.c_void_p),
                                                 None if m is None else m.ctypes.data_as(C.c_void_p), _stream()), self.lib)
            torch.cuda.current_stream().synchronize()      # seeds staging buffer is reused

    def reset(self, mask=None):
        m = None if mask is None else torch.as_tensor(mask).to(device=self.device, dtype=torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            L.raise_for(self.lib.inv_reset(self.h, _ptr(m), _stream()), self.lib)

    def check(self):
        """Synchronise and raise the sticky device error, if any (bad action id / bullet overflow)."""
        with torch.cuda.device(self.device):
            L.raise_for(self.lib.inv_check(self.h, _stream()), self.lib)

    # ------------------------------------------------------------------ step / render
    def _alloc_out(self, lead, obs, obs_p2, reward64, out, bits=False):
        if out is not None:
            return out
        n = self.num_envs
        r = StepResult(
            obs=self._new_obs(lead, bits) if obs else None,
            extra=self._new(tuple(lead) + (n, 4), torch.float32) if obs else None,
            obs_p2=self._new_obs(lead, bits) if obs_p2 else None,
            extra_p2=self._new(tuple(lead) + (n, 4), torch.float32) if obs_p2 else None,
            reward=self._new(tuple(lead) + (n,), torch.float32),
            reward64=self._new(tuple(lead) + (n,), torch.float64) if reward64 else None,
            done=self._new(tuple(lead) + (n,), torch.uint8),
            info_flags=self._new(tuple(lead) + (n,), torch.uint8),
            episode_steps=self._new(tuple(lead) + (n,), torch.int32),
            episode_return=self._new(tuple(lead) + (n,), torch.float64),
        )
        return r

    @staticmethod
    def _c_out(r):
        return L.InvStepOut(*[_ptr(getattr(r, k)) for k in StepResult.__slots__])

    def step(self, a1, a2=None, auto_reset=False, flags=L.INV_STEP_DEFAULT, obs=True, obs_p2=False,
             reward64=False, out=None, mask=None, bits=False):
        """One vector step.  a2=None uses the device dummy opponent (opponent="dummy" only).
        bits=True (or int32 buffers in `out`): observations as packed bitstrings [N, OW] int32."""
        a1t, a2t = self._as_actions(a1), self._as_actions(a2)
        r = self._alloc_out((), obs, obs_p2, reward64, out, bits)
        if mask is not None and out is None:
            # a masked step leaves the slots of the envs that do not step untouched: in buffers allocated here they read as
            # zeros, not as whatever the allocator handed back (a caller's own `out` keeps its previous contents)
            for k in StepResult.__slots__:
                if getattr(r, k) is not None:
                    getattr(r, k).zero_()
        co = self._c_out(r)
        fl = int(flags) | (L.INV_STEP_AUTO_RESET if auto_reset else 0) | (L.INV_STEP_OBS_BITS if self._is_bits(r) else 0)
        with torch.cuda.device(self.device):
            if mask is None:
                rc = self.lib.inv_step(self.h, _ptr(a1t), _ptr(a2t), fl, C.byref(co), _stream())
            else:
                m = torch.as_tensor(mask).to(device=self.device, dtype=torch.uint8).contiguous()
                rc = self.lib.inv_step_masked(self.h, _ptr(m), _ptr(a1t), _ptr(a2t), fl, C.byref(co), _stream())
        L.raise_for(rc, self.lib)
        return r

    def make_stepper(self, out, a2=None, auto_reset=True, flags=L.INV_STEP_DEFAULT):
        """Pre-marshalled inv_step: returns step(a1_device_address) costing one ctypes call.  For
        launch-rate-sensitive loops (the per-step Python marshalling of step() is ~20 us)."""
        co = self._c_out(out)
        a2t = self._as_actions(a2)
        fl = int(flags) | (L.INV_STEP_AUTO_RESET if auto_reset else 0) | (L.INV_STEP_OBS_BITS if self._is_bits(out) else 0)
        fn, h, ref, a2p, st = self.lib.inv_step, self.h, C.byref(co), _ptr(a2t), _stream()
        lib = self.lib

        def step(a1_ptr):
            rc = fn(h, a1_ptr, a2p, fl, ref, st)
            if rc:
                L.raise_for(rc, lib)
        step.keepalive = (co, out, a2t)
        return step

    def rollout(self, a1, a2=None, obs=True, obs_p2=False, reward64=False, out=None, bits=False):
        """T vector steps in one launch with pre-supplied actions a1[T, N] (always auto-resets)."""
        T = int(a1.shape[0])
        a1t, a2t = self._as_actions(a1, (T,)), self._as_actions(a2, (T,))
        r = self._alloc_out((T,), obs, obs_p2, reward64, out, bits)
        co = self._c_out(r)
        fl = L.INV_STEP_OBS_BITS if self._is_bits(r) else 0
        with torch.cuda.device(self.device):
            L.raise_for(self.lib.inv_rollout(self.h, T, _ptr(a1t), _ptr(a2t), fl, C.byref(co), _stream()), self.lib)
        return r

    def render(self, player=0, out=None, bits=False):
        """build_observation for every env: (grid[N,12,H,W], extra[N,4]) device tensors; bits=True (or an
        int32 out[0]): the grid as packed bitstrings [N, OW] int32."""
        if out is None:
            out = (self._new_obs((), bits), self._new((self.num_envs, 4), torch.float32))
        fn = self.lib.inv_render_obs_bits if out[0].dtype == torch.int32 else self.lib.inv_render_obs
        with torch.cuda.device(self.device):
            L.raise_for(fn(self.h, int(player), _ptr(out[0]), _ptr(out[1]), _stream()), self.lib)
        return out

    def dummy_actions(self, difficulty=None, out=None):
        """dummy_opponent_policy (env_wrappers.py:69-170) for every env without stepping it: uint8[N] action ids (0..8) the
        device dummy takes for the current states -- what the next step(a1, None) applies when the batch's difficulty is
        `difficulty` ("easy" / "hard"; default: the batch's own).  Its counter RNG is read, not advanced."""
        d = self.difficulty if difficulty is None else difficulty
        if d not in ("easy", "hard"):
            raise ValueError(f"Unknown difficulty: {d}")
        if out is None:
            out = self._new((self.num_envs,), torch.uint8)
        with torch.cuda.device(self.device):
            L.raise_for(self.lib.inv_dummy_actions(self.h, L.INV_HARD if d == "hard" else L.INV_EASY, _ptr(out), _stream()), self.lib)
        return out

    def uniform_actions(self, seed, t, out=None):
        if out is None:
            out = self._new((self.num_envs,), torch.uint8)
        with torch.cuda.device(self.device):
            L.raise_for(self.lib.inv_uniform_actions(_ptr(out), self.num_envs, int(seed) & 0xFFFFFFFF,
                                                     self.env_offset, int(t) & 0xFFFFFFFF, _stream()), self.lib)
        return out

    # ------------------------------------------------------------------ state dump / poke
    _STATE_FIELDS = (("grid", torch.int32, lambda s: (s.num_envs, s.GW), np.uint32),
                     ("players", torch.int32, lambda s: (s.num_envs, 2, 5), np.int32),
                     ("nbullets", torch.int32, lambda s: (s.num_envs,), np.int32),
                     ("bullets", torch.int16, lambda s: (s.num_envs, s.K), np.uint16),
                     ("step_count", torch.int32, lambda s: (s.num_envs,), np.int32),
                     ("episode_return", torch.float64, lambda s: (s.num_envs,), np.float64),
                     ("prev_alive", torch.uint8, lambda s: (s.num_envs, 2), np.uint8),
                     ("opp_ctr", torch.int32, lambda s: (s.num_envs,), np.uint32),
                     ("mt", torch.int32, lambda s: (s.num_envs, 624), np.uint32),
                     ("mt_index", torch.int32, lambda s: (s.num_envs,), np.int32))

    def get_state(self, fields=None):
        """Canonical state as numpy arrays (see include/inversus_hip.h); synchronises."""
        want = [f for f in self._STATE_FIELDS if (fields is None and f[0] != "mt") or (fields and f[0] in fields)]
        bufs = {name: self._new(shape(self), tdt) for name, tdt, shape, _ in want}
        view = L.InvStateView(**{name: _ptr(t) for name, t in bufs.items()})
        with torch.cuda.device(self.device):
            L.raise_for(self.lib.inv_get_state(self.h, C.byref(view), _stream()), self.lib)
        return {name: bufs[name].cpu().numpy().view(npdt) for name, _, _, npdt in want}

    def set_state(self, **arrays):
        """Overwrite the given canonical state arrays (full [N, ...] arrays) on the device."""
        bufs = {}
        for name, tdt, shape, npdt in self._STATE_FIELDS:
            if name in arrays:
                a = np.ascontiguousarray(arrays[name]).astype(npdt, copy=False)
                if a.shape != shape(self):
                    raise ValueError(f"{name}: expected shape {shape(self)}, got {a.shape}")
                signed = a.view({np.uint32: np.int32, np.uint16: np.int16}.get(npdt, npdt))
                bufs[name] = torch.from_numpy(np.ascontiguousarray(signed)).to(self.device)
        unknown = set(arrays) - {f[0] for f in self._STATE_FIELDS}
        if unknown:
            raise ValueError(f"unknown state fields: {sorted(unknown)}")
        view = L.InvStateView(**{name: _ptr(t) for name, t in bufs.items()})
        with torch.cuda.device(self.device):
            L.raise_for(self.lib.inv_set_state(self.h, C.byref(view), _stream()), self.lib)
            torch.cuda.current_stream().synchronize()      # keep the staging tensors alive until consumed


class EnvShards:
    """The envs of ONE GPU as `shards` independent DeviceEnvBatch shards (contiguous global index ranges, so results do not depend on
    the cut), each bound to a HIP stream of its own.  Shards never exchange data -- the way the shards of different GPUs do not -- so
    their launch chains may run free: one shard's launch boundary and the ~2 us before its first observation store then hide under
    another shard's store stream (two shards of 8 192 envs: 18.6 -> 15.4 us per full 16 384-env step on a cache-resident buffer,
    25.3 -> 20.6 us streaming).  It is for callers that can consume shards independently (a double-buffered
    collector: step shard A while the policy evaluates shard B); behind a synchronous `step()` that returns every env's observation
    the chains would run in lockstep and nothing is gained.  Policy forwards of THIS library (FusedPolicy: trunk, inv_fc1_fwd, heads,
    sampling) may run on several streams at once -- none of their workgroups waits on another; a GEMM-LIBRARY call on two streams at
    once may not (stream-K kernels sized for the whole chip dead-lock each other: profiles/r04_incidents.md).

        shards = EnvShards(16384, 2, opponent="dummy")            # seeds: base + global index, like MultiEnvRunner
        for sh in shards:                                          # sh.batch, sh.stream, sh.offset, sh.count
            with torch.cuda.stream(sh.stream): ...                 # every call on a shard goes to its stream
    """

    class Shard:
        __slots__ = ("batch", "stream", "offset", "count")

        def __init__(self, batch, stream, offset, count):
            self.batch, self.stream, self.offset, self.count = batch, stream, offset, count

    def __init__(self, num_envs, shards=2, *, seed=0, env_offset=0, streams=None, **batch_kwargs):
        if shards < 1 or num_envs < shards:
            raise ValueError("EnvShards: need 1 <= shards <= num_envs")
        self.num_envs = int(num_envs)
        base, rem = divmod(self.num_envs, int(shards))
        self.shards, off = [], int(env_offset)
        for k in range(int(shards)):
            cnt = base + (1 if k < rem else 0)
            st = streams[k] if streams is not None else torch.cuda.Stream()
            with torch.cuda.stream(st):
                b = DeviceEnvBatch(cnt, env_offset=off, **batch_kwargs)
                b.seed(int(seed) + off + np.arange(cnt, dtype=np.int64))
                b.reset()
            st.synchronize()
            self.shards.append(EnvShards.Shard(b, st, off, cnt))
            off += cnt

    def __iter__(self):
        return iter(self.shards)

    def __len__(self):
        return len(self.shards)

    def synchronize(self):
        for sh in self.shards:
            sh.stream.synchronize()

    def check(self):
        for sh in self.shards:
            with torch.cuda.stream(sh.stream):
                sh.batch.check()


def unpack_obs(bits, height=10, width=15, out=None):
    """Packed observations [..., OW] int32 (INV_STEP_OBS_BITS) -> f32 planes [..., 12, H, W]; exact."""
    lib = L.load()
    if bits.dtype != torch.int32 or not bits.is_cuda:
        raise ValueError("unpack_obs expects a CUDA int32 tensor of packed observations")
    floats = 12 * int(height) * int(width)
    if bits.shape[-1] != (floats + 31) // 32:
        raise ValueError(f"packed observations of a {width}x{height} board have {(floats + 31) // 32} words, got {bits.shape[-1]}")
    b = bits.contiguous()
    lead = tuple(b.shape[:-1])
    n = int(np.prod(lead)) if lead else 1
    if out is None:
        out = torch.empty(lead + (12, int(height), int(width)), dtype=torch.float32, device=b.device)
    elif out.dtype != torch.float32 or out.numel() != n * floats or not out.is_contiguous():
        raise ValueError("unpack_obs: `out` must be a contiguous float32 tensor of matching size")
    with torch.cuda.device(b.device):
        L.raise_for(lib.inv_unpack_obs(_ptr(b), n, floats, _ptr(out), _stream()), lib)
    return out


def gae(rewards, values, dones, gamma=0.99, lam=0.95, last_value=None, layout="per_env"):
    """GAE / returns on device (ppo_agent.py:127-157), strict f32.  rewards/values f32 [T, N],
    dones bool/u8 [T, N].  layout: "per_env" or "reference_flat"."""
    lib = L.load()
    r = rewards.contiguous().float()
    v = values.contiguous().float()
    d = dones.contiguous()
    d = d.view(torch.uint8) if d.dtype == torch.bool else d.to(torch.uint8)     # bool -> u8 is a reinterpretation, no copy
    if r.dim() == 1:
        r, v, d = r[:, None], v[:, None], d[:, None]
    T, N = r.shape
    adv, ret = torch.empty_like(r), torch.empty_like(r)
    lv = None
    if last_value is not None:
        lv = torch.as_tensor(last_value, dtype=torch.float32, device=r.device).reshape(-1).contiguous()
    code = {"per_env": L.INV_GAE_PER_ENV, "reference_flat": L.INV_GAE_REFERENCE_FLAT}[layout]
    # gamma * lam is a Python-float (f64) product rounded once to f32, like the reference's
    # `self.gamma * self.lam * last_gae` under NumPy >= 2 weak-scalar promotion
    gl = float(np.float32(float(gamma) * float(lam)))
    with torch.cuda.device(r.device):
        rc = lib.inv_gae(_ptr(r), _ptr(v), _ptr(d), T, N, float(np.float32(gamma)), gl, _ptr(lv), code,
                         _ptr(adv), _ptr(ret), _stream())
    L.raise_for(rc, lib)
    return adv, ret
