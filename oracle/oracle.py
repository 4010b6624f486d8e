"""ctypes front-end of the CPU oracle (oracle/inv_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never by the product package.  Parity
status: pinned (see inv_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_inversus.so")

MAX_DIM = 64
MAX_BULLETS = 256


class OrcBullet(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("dir", C.c_int), ("owner", C.c_int)]


class OrcPlayer(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("ammo", C.c_int), ("max_ammo", C.c_int),
                ("reload", C.c_int), ("alive", C.c_int)]


class OrcEnv(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int),
                ("grid", C.c_uint8 * (MAX_DIM * MAX_DIM)),
                ("p", OrcPlayer * 2),
                ("bullets", OrcBullet * MAX_BULLETS),
                ("nb", C.c_int),
                ("mt", C.c_uint32 * 624),
                ("mti", C.c_int),
                ("max_episode_steps", C.c_int),
                ("step_count", C.c_int),
                ("episode_return", C.c_double),
                ("prev_alive", C.c_int * 2),
                ("opp_ctr", C.c_uint32)]


class OrcInfo(C.Structure):
    _fields_ = [("landed_hit", C.c_int), ("got_hit", C.c_int), ("win", C.c_int), ("lose", C.c_int),
                ("episode_steps", C.c_int), ("episode_return", C.c_double)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = [os.path.join(_HERE, f) for f in ("inv_oracle.c", "inv_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    override = os.environ.get("INVERSUS_ORACLE_LIB")     # another build of the same source (the sanitizer build of the tests)
    if not override:
        build()
    L = C.CDLL(override or _LIB_PATH)
    P = C.POINTER
    env_p = P(OrcEnv)
    sig = {
        "orc_mt_seed": (None, [env_p, C.c_int64]),
        "orc_mt_u32": (C.c_uint32, [env_p]),
        "orc_randbelow": (C.c_int, [env_p, C.c_int]),
        "orc_init": (C.c_int, [env_p, C.c_int, C.c_int, C.c_int]),
        "orc_core_reset": (None, [env_p]),
        "orc_try_move": (C.c_int, [env_p, C.c_int, C.c_int]),
        "orc_spawn_bullet": (C.c_int, [env_p, C.c_int, C.c_int]),
        "orc_spawn_wide": (C.c_int, [env_p, C.c_int, C.c_int]),
        "orc_reload": (None, [env_p]),
        "orc_update_bullets": (None, [env_p]),
        "orc_round_over": (C.c_int, [env_p]),
        "orc_winner": (C.c_int, [env_p]),
        "orc_step_players": (C.c_int, [env_p, C.c_int, C.c_int]),
        "orc_build_obs": (None, [env_p, C.c_int, C.c_void_p, C.c_void_p]),
        "orc_wrapper_reset": (None, [env_p]),
        "orc_wrapper_step": (C.c_int, [env_p, C.c_int, C.c_int, P(C.c_double), P(C.c_int), P(OrcInfo)]),
        "orc_dummy_decide": (C.c_int, [env_p, C.c_int, P(C.c_double), P(C.c_int), C.c_int, P(C.c_int)]),
        "orc_philox4x32": (None, [C.c_uint32] * 6 + [P(C.c_uint32)]),
        "orc_dummy_device": (C.c_int, [env_p, C.c_int, C.c_uint32, C.c_uint32]),
        "orc_uniform_action": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32]),
        "orc_gae_flat": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                C.c_float, C.c_void_p, C.c_void_p]),
        "orc_gae_per_env": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                   C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
        "orc_batch_create": (env_p, [C.c_int, C.c_int, C.c_int, C.c_int]),
        "orc_batch_destroy": (None, [env_p]),
        "orc_batch_at": (env_p, [env_p, C.c_int]),
        "orc_batch_seed_reset": (None, [env_p, C.c_int, C.c_void_p]),
        "orc_batch_step": (None, [env_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32,
                                  C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int]),
        "orc_get_grid_bits": (None, [env_p, C.c_void_p]),
        "orc_set_grid_bits": (None, [env_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pack_bullet(x, y, d, owner):
    """Canonical u16 bullet word shared with include/inversus_hip.h."""
    return (x & 63) | ((y & 63) << 6) | ((d & 3) << 12) | ((owner & 1) << 14)


def unpack_bullet(w):
    w = int(w)
    return (w & 63, (w >> 6) & 63, (w >> 12) & 3, (w >> 14) & 1)


class OracleEnv:
    """One env; method names follow the reference (core.py / SingleInversusRLEnv)."""

    def __init__(self, width=15, height=10, seed=None, max_episode_steps=500, _ref=None):
        self.L = lib()
        if _ref is None:
            self._own = OrcEnv()
            self.e = C.pointer(self._own)
            if self.L.orc_init(self.e, width, height, max_episode_steps) != 0:
                raise ValueError("empty range for randrange()")  # core.py:69-70 needs W,H >= 3
            if seed is not None:
                self.L.orc_mt_seed(self.e, int(seed))
                self.L.orc_wrapper_reset(self.e)
        else:
            self.e = _ref

    # --- state access -------------------------------------------------------
    @property
    def s(self):
        return self.e.contents

    @property
    def width(self):
        return self.s.W

    @property
    def height(self):
        return self.s.H

    def seed(self, seed):
        self.L.orc_mt_seed(self.e, int(seed))

    def reset(self, seed=None):
        if seed is not None:
            self.seed(seed)
        self.L.orc_wrapper_reset(self.e)

    def grid_bits(self):
        n = (self.s.W * self.s.H + 31) // 32
        out = np.zeros(n, dtype=np.uint32)
        self.L.orc_get_grid_bits(self.e, _ptr(out))
        return out

    def set_grid_bits(self, words):
        w = np.ascontiguousarray(words, dtype=np.uint32)
        self.L.orc_set_grid_bits(self.e, _ptr(w))

    def get_tile(self, x, y):
        return int(self.s.grid[y * self.s.W + x])

    def set_tile(self, x, y, white):
        self.s.grid[y * self.s.W + x] = 1 if white else 0

    def players(self):
        """[[x, y, ammo, reload, alive]] * 2 (canonical player row)."""
        return np.array([[p.x, p.y, p.ammo, p.reload, p.alive] for p in self.s.p], dtype=np.int32)

    def set_player(self, pid, x=None, y=None, ammo=None, reload=None, alive=None):
        p = self.s.p[pid]
        if x is not None: p.x = x
        if y is not None: p.y = y
        if ammo is not None: p.ammo = ammo
        if reload is not None: p.reload = reload
        if alive is not None: p.alive = int(alive)

    def bullets(self):
        return [(b.x, b.y, b.dir, b.owner) for b in self.s.bullets[:self.s.nb]]

    def bullets_packed(self, k):
        out = np.zeros(k, dtype=np.uint16)
        for i, b in enumerate(self.bullets()[:k]):
            out[i] = pack_bullet(*b)
        return out

    def set_bullets(self, lst):
        self.s.nb = len(lst)
        for i, (x, y, d, o) in enumerate(lst):
            self.s.bullets[i] = OrcBullet(x, y, d, o)

    # --- core.py ------------------------------------------------------------
    def try_move_player(self, d, pid): return bool(self.L.orc_try_move(self.e, pid, d))
    def spawn_bullet(self, d, pid): return bool(self.L.orc_spawn_bullet(self.e, pid, d))
    def spawn_wide_shot(self, pid, d): return bool(self.L.orc_spawn_wide(self.e, pid, d))
    def reload_ammo(self): self.L.orc_reload(self.e)
    def update_bullets(self): self.L.orc_update_bullets(self.e)
    def is_round_over(self): return bool(self.L.orc_round_over(self.e))
    def get_winner(self): return self.L.orc_winner(self.e)

    def step_players(self, a1, a2):
        if self.L.orc_step_players(self.e, int(a1), int(a2)) != 0:
            raise ValueError("Invalid action_id, must be 0-12")

    # --- env_wrappers.py ----------------------------------------------------
    def build_observation(self, pid=0):
        g = np.empty((12, self.s.H, self.s.W), dtype=np.float32)
        x = np.empty(4, dtype=np.float32)
        self.L.orc_build_obs(self.e, pid, _ptr(g), _ptr(x))
        return g, x

    def step(self, a1, a2):
        r = C.c_double()
        d = C.c_int()
        info = OrcInfo()
        if self.L.orc_wrapper_step(self.e, int(a1), int(a2), C.byref(r), C.byref(d), C.byref(info)) != 0:
            raise ValueError("Invalid action_id, must be 0-12")
        return r.value, bool(d.value), info

    def dummy_decide(self, hard, u, perm_a, swap_c, perm_b):
        ua = (C.c_double * 3)(*u)
        pa = (C.c_int * 4)(*perm_a)
        pb = (C.c_int * 4)(*perm_b)
        return self.L.orc_dummy_decide(self.e, int(hard), ua, pa, int(swap_c), pb)

    def dummy_device(self, hard, seed, global_env):
        return self.L.orc_dummy_device(self.e, int(hard), seed, global_env)


class OracleBatch:
    """N envs stepped like MultiEnvRunner.step + the training loop's auto-reset."""

    def __init__(self, n, width=15, height=10, max_episode_steps=500):
        self.L = lib()
        self.n, self.W, self.H = n, width, height
        self.envs = self.L.orc_batch_create(n, width, height, max_episode_steps)
        if not self.envs:
            raise ValueError("bad oracle batch dims")

    def __del__(self):
        if getattr(self, "envs", None):
            self.L.orc_batch_destroy(self.envs)
            self.envs = None

    def env(self, i):
        return OracleEnv(_ref=self.L.orc_batch_at(self.envs, i))

    def seed_reset(self, seeds):
        s = np.ascontiguousarray(seeds, dtype=np.int64)
        assert s.shape == (self.n,)
        self.L.orc_batch_seed_reset(self.envs, self.n, _ptr(s))

    def observe(self, pid=0):
        g = np.empty((self.n, 12, self.H, self.W), dtype=np.float32)
        x = np.empty((self.n, 4), dtype=np.float32)
        for i in range(self.n):
            self.L.orc_build_obs(self.L.orc_batch_at(self.envs, i), pid, _ptr(g[i]), _ptr(x[i]))
        return g, x

    def alloc_out(self, want_obs=True):
        """Output buffers for step(out=...): allocated once, untouched (the first step's writers fault the pages in)."""
        n = self.n
        return (np.empty((n, 12, self.H, self.W), dtype=np.float32) if want_obs else None,
                np.empty((n, 4), dtype=np.float32) if want_obs else None,
                np.empty(n, dtype=np.float32), np.empty(n, dtype=np.float64), np.empty(n, dtype=np.uint8), (OrcInfo * n)())

    def step(self, a1, a2=None, hard=False, opp_seed=0, env_offset=0, auto_reset=True,
             want_obs=True, nthreads=1, out=None):
        n = self.n
        a1 = np.ascontiguousarray(a1, dtype=np.uint8)
        a2c = None if a2 is None else np.ascontiguousarray(a2, dtype=np.uint8)
        obs, extra, reward, reward64, done, info = out if out is not None else self.alloc_out(want_obs)
        self.L.orc_batch_step(self.envs, n, _ptr(a1), _ptr(a2c), int(hard), opp_seed, env_offset,
                              int(auto_reset), _ptr(obs), _ptr(extra), _ptr(reward), _ptr(reward64),
                              _ptr(done), C.cast(info, C.c_void_p), nthreads)
        return obs, extra, reward, reward64, done, info

    def state(self, k):
        """Canonical state arrays (same packing as inv_get_state)."""
        n = self.n
        gw = (self.W * self.H + 31) // 32
        grid = np.zeros((n, gw), np.uint32)
        players = np.zeros((n, 2, 5), np.int32)
        nb = np.zeros(n, np.int32)
        bullets = np.zeros((n, k), np.uint16)
        steps = np.zeros(n, np.int32)
        epret = np.zeros(n, np.float64)
        prev = np.zeros((n, 2), np.uint8)
        for i in range(n):
            e = self.env(i)
            grid[i] = e.grid_bits()
            players[i] = e.players()
            nb[i] = e.s.nb
            bullets[i] = e.bullets_packed(k)
            steps[i] = e.s.step_count
            epret[i] = e.s.episode_return
            prev[i] = (e.s.prev_alive[0], e.s.prev_alive[1])
        return dict(grid=grid, players=players, nbullets=nb, bullets=bullets, step_count=steps,
                    episode_return=epret, prev_alive=prev)


def gae_flat(r, v, done, gamma=0.99, lam=0.95, last_value=0.0):
    r = np.ascontiguousarray(r, np.float32); v = np.ascontiguousarray(v, np.float32)
    d = np.ascontiguousarray(done, np.uint8)
    adv = np.empty_like(r); ret = np.empty_like(r)
    gl = np.float32(gamma * lam)           # python-float product, then weak-cast to f32
    lib().orc_gae_flat(_ptr(r), _ptr(v), _ptr(d), r.size, np.float32(gamma), gl, np.float32(last_value),
                       _ptr(adv), _ptr(ret))
    return adv, ret


def gae_per_env(r, v, done, gamma=0.99, lam=0.95, last_value=None):
    r = np.ascontiguousarray(r, np.float32); v = np.ascontiguousarray(v, np.float32)
    d = np.ascontiguousarray(done, np.uint8)
    T, N = r.shape
    adv = np.empty_like(r); ret = np.empty_like(r)
    lv = None if last_value is None else np.ascontiguousarray(last_value, np.float32)
    lib().orc_gae_per_env(_ptr(r), _ptr(v), _ptr(d), T, N, np.float32(gamma), np.float32(gamma * lam),
                          _ptr(lv), _ptr(adv), _ptr(ret))
    return adv, ret
