"""Drop-in for the reference's RL env boundary (inversus_rl/env_wrappers.py), backed by the
HIP kernels: same class / function names, argument order, return shapes and exceptions.

    MultiEnvRunner(num_envs, opponent_type, difficulty, max_episode_steps, seed)   :447-528
    SingleInversusRLEnv(opponent_type, difficulty, max_episode_steps, seed)        :248-444
    build_observation(env, player_id)                                              :173-245
    discrete_to_action(action_id)                                                  :20-66
    dummy_opponent_policy(env, difficulty)                                         :69-170

Differences a caller can see (all additive, keyword-only):
  * `output="numpy"` (default) returns fresh host arrays exactly like the reference;
    `output="torch"` returns device tensors and never leaves the GPU.
  * `width/height/max_bullets/device/env_offset/opp_seed/auto_reset` extend the constructor
    (the reference wrapper cannot pass width/height, env_wrappers.py:264).
  * the dummy opponent draws from a per-env counter RNG instead of the process-global
    `random` module (env_wrappers.py:96-155), so its action stream is reproducible per env
    but not bit-equal to the reference's; everything else is bit-identical.
  * `opponent_policy` may also be batched: a callable with attribute `batched = True` gets
    (grid[N,12,H,W], extra[N,4]) device tensors and returns N action ids.
There is no CPU fallback: without the HIP library and a GPU the constructors raise.
"""
import os

import numpy as np
import torch

from . import _lib as L
from .device_env import DeviceEnvBatch
from .vocab import Action, ActionType, Bullet, PlayerId, PlayerState, TileColor, CODE_BY_DIR, DIR_BY_CODE, TYPE_BY_CODE

MAX_AMMO = 6


def discrete_to_action(action_id):
    """0 NONE; 1-4 MOVE, 5-8 SHOOT, 9-12 CHARGE_SHOOT, each in the order UP, RIGHT, DOWN, LEFT."""
    a = int(action_id)
    if not 0 <= a <= 12:
        raise ValueError(f"Invalid action_id: {action_id}, must be 0-12")
    if a == 0:
        return Action(ActionType.NONE, None)
    return Action(TYPE_BY_CODE[(a + 3) >> 2], DIR_BY_CODE[(a - 1) & 3])


def dummy_opponent_policy(env, difficulty="easy"):
    """The scripted dummy's action for P2 in `env`'s current state (env_wrappers.py:69-170): 'easy' = the sitting duck (moves in
    0.1 % of the steps, never shoots), anything else = the hunter (20 % shots when aligned, 5 % random moves, otherwise towards P1).
    `env` is the core object the reference passes (`rl_env.env` / `runner.envs[i].env`; an rl env or a `runner.envs[i]` handle is
    accepted too).  The decision is the device dummy's (inv_dummy_actions): the same function of the state as the reference's, its
    draws from the env's counter RNG instead of the global `random` -- so repeated calls without a step in between agree, and the
    returned action is the one the env's next step against the dummy applies."""
    core = env.env if isinstance(env, (SingleInversusRLEnv, _EnvHandle)) else env
    if not isinstance(core, _CoreView):
        raise TypeError("dummy_opponent_policy expects an env object of this package")
    a2 = core._batch.dummy_actions("easy" if difficulty == "easy" else "hard")
    return discrete_to_action(int(a2[core._i].item()))


def _entropy_seed():
    return int.from_bytes(os.urandom(8), "little") >> 1


class _FrozenBullet(Bullet):
    """A bullet handed out by `env.bullets`: a copy of device state.  Changing a field of the copy would not reach the
    device, so it raises; assign a new list (`env.bullets = [...]`) instead, as the reference's tests do."""
    __slots__ = ("_frozen",)

    def __init__(self, x, y, dir, owner):
        Bullet.__init__(self, x, y, dir, owner)
        object.__setattr__(self, "_frozen", True)

    def __setattr__(self, name, value):
        if getattr(self, "_frozen", False):
            raise AttributeError("bullets read from the device are copies: assign env.bullets = [...] to change them")
        object.__setattr__(self, name, value)

    def __eq__(self, other):
        return isinstance(other, Bullet) and (self.x, self.y, self.dir, self.owner) == (other.x, other.y, other.dir, other.owner)

    __hash__ = None


class _BulletList(list):
    """`env.bullets`: a list whose mutations are written back to the device (append / extend / clear / pop / remove / insert /
    item assignment / del / += / sort / reverse)."""

    def __init__(self, view, items):
        super().__init__(items)
        self._view = view


def _writes_back(name):
    def method(self, *a, **k):
        out = getattr(list, name)(self, *a, **k)
        self._view.bullets = list(self)
        return self if name == "__iadd__" else out
    method.__name__ = name
    return method


for _m in ("append", "extend", "clear", "pop", "remove", "insert", "__setitem__", "__delitem__", "__iadd__", "sort", "reverse"):
    setattr(_BulletList, _m, _writes_back(_m))


class _PlayerView:
    """`env.player1` / `env.player2`: the reference's PlayerState fields (game_types.py:52-62) LIVE on the device --
    reads fetch the env's record, assignments (`env.player1.x = 3`, the idiom of every reference test) go through inv_set_state."""
    _COL = {"x": 0, "y": 1, "ammo": 2, "reload_counter": 3, "alive": 4}

    def __init__(self, view, q):
        object.__setattr__(self, "_view", view)
        object.__setattr__(self, "_q", q)

    player_id = property(lambda self: PlayerId.P2 if self._q else PlayerId.P1)
    color = property(lambda self: TileColor.WHITE if self._q else TileColor.BLACK)       # config.py:9,11; core.py:76,128
    max_ammo = property(lambda self: MAX_AMMO)

    def __getattr__(self, name):
        col = _PlayerView._COL.get(name)
        if col is None:
            raise AttributeError(name)
        v = int(self._view._field("players")[self._view._i, self._q, col])
        return bool(v) if name == "alive" else v

    def __setattr__(self, name, value):
        col = _PlayerView._COL.get(name)
        if col is None:
            raise AttributeError(f"PlayerState.{name} is not settable on the device view")
        v = int(bool(value)) if name == "alive" else int(value)
        if not 0 <= v <= 255:
            raise ValueError(f"{name}={value} does not fit the device record (0..255)")
        pl = self._view._field("players")
        pl[self._view._i, self._q, col] = v
        self._view._batch.set_state(players=pl)

    def __repr__(self):
        return (f"PlayerState(player_id={self.player_id}, x={self.x}, y={self.y}, color={self.color}, ammo={self.ammo}, "
                f"max_ammo={self.max_ammo}, reload_counter={self.reload_counter}, alive={self.alive})")


class _GridRow:
    def __init__(self, view, y):
        self._view, self._y = view, y

    def __len__(self):
        return self._view.width

    def __getitem__(self, x):
        if isinstance(x, slice):
            return [self[k] for k in range(*x.indices(self._view.width))]
        return self._view._get_tile(x if x >= 0 else x + self._view.width, self._y)

    def __setitem__(self, x, color):
        self._view._set_tile(x if x >= 0 else x + self._view.width, self._y, color)

    def __iter__(self):
        return iter(self[:])

    def count(self, color):
        return self[:].count(color)

    def __eq__(self, other):
        return self[:] == list(other)


class _GridView:
    """`env.grid`: `env.grid[y][x]` reads and writes the device bitstring."""

    def __init__(self, view):
        self._view = view

    def __len__(self):
        return self._view.height

    def __getitem__(self, y):
        if isinstance(y, slice):
            return [self[k] for k in range(*y.indices(self._view.height))]
        if not -self._view.height <= y < self._view.height:
            raise IndexError("list index out of range")
        return _GridRow(self._view, y % self._view.height)

    def __iter__(self):
        return (self[y] for y in range(self._view.height))


class _CoreView:
    """`runner.envs[i].env`: env i's core state with the reference's field and method names (InversusEnv, core.py:17-544),
    LIVE: attribute reads fetch from the device, attribute writes go through inv_set_state, and the core's own methods
    (`update_bullets`, `step_players`, `try_move_player`, `spawn_bullet`, `spawn_wide_shot`, `_reload_ammo`, `step`) run the
    corresponding phase of the step kernel on this env only (inv_step_masked with INV_PHASE_* flags), so the reference's
    state-poking tests read the same here.  Every access synchronises: this is the debugging / test surface, not the hot path."""

    def __init__(self, batch, i):
        self._batch, self._i = batch, int(i)
        self.width, self.height = batch.W, batch.H
        self.player_color = TileColor.BLACK
        self.player1, self.player2 = _PlayerView(self, 0), _PlayerView(self, 1)

    # ---- state plumbing
    def _field(self, name):
        return self._batch.get_state((name,))[name]

    def __setattr__(self, name, value):
        if name in ("player1", "player2") and name in self.__dict__:
            raise AttributeError(f"assign the fields of env.{name} (x, y, ammo, reload_counter, alive), not the object")
        if name == "grid":
            rows = [list(r) for r in value]
            if len(rows) != self.height or any(len(r) != self.width for r in rows):
                raise ValueError("grid must be height x width")
            g = self._field("grid")
            g[self._i] = 0
            for y, row in enumerate(rows):
                for x, c in enumerate(row):
                    if c == TileColor.WHITE:
                        p = y * self.width + x
                        g[self._i, p >> 5] |= np.uint32(1 << (p & 31))
            self._batch.set_state(grid=g)
            return
        object.__setattr__(self, name, value)

    @property
    def grid(self):
        return _GridView(self)

    @property
    def bullets(self):
        st = self._batch.get_state(("bullets", "nbullets"))
        out = []
        for w in st["bullets"][self._i][:int(st["nbullets"][self._i])]:
            w = int(w)
            out.append(_FrozenBullet(w & 63, (w >> 6) & 63, DIR_BY_CODE[(w >> 12) & 3], PlayerId.P2 if (w >> 14) & 1 else PlayerId.P1))
        return _BulletList(self, out)

    @bullets.setter
    def bullets(self, lst):
        lst = list(lst)
        if len(lst) > self._batch.K:
            raise ValueError(f"more than max_bullets={self._batch.K} bullets")
        st = self._batch.get_state(("bullets", "nbullets"))
        st["bullets"][self._i] = 0
        for k, b in enumerate(lst):
            if not (0 <= int(b.x) < 64 and 0 <= int(b.y) < 64):
                raise ValueError(f"bullet position ({b.x}, {b.y}) does not fit the device record")
            st["bullets"][self._i, k] = int(b.x) | (int(b.y) << 6) | (CODE_BY_DIR[b.dir] << 12) | ((0 if b.owner == PlayerId.P1 else 1) << 14)
        st["nbullets"][self._i] = len(lst)
        self._batch.set_state(bullets=st["bullets"], nbullets=st["nbullets"])

    # ---- legacy accessors of player 1's position (core.py:156-181; the reference's tests/test_core_basic.py:15-43 move the
    # player with them).  As in the reference, ASSIGNING player_y also clears the bullet list (core.py:179-181: the two lines sit
    # inside that setter).
    @property
    def player_x(self):
        return self.player1.x

    @player_x.setter
    def player_x(self, value):
        self.player1.x = value

    @property
    def player_y(self):
        return self.player1.y

    @player_y.setter
    def player_y(self, value):
        self.player1.y = value
        self.bullets = []

    # ---- reads (core.py:183-247, 477-495)
    def get_player(self, pid):
        if pid == PlayerId.P1:
            return self.player1
        if pid == PlayerId.P2:
            return self.player2
        raise ValueError(f"Invalid player ID: {pid}")

    def get_player_position(self, pid=None):
        p = self.get_player(PlayerId.P1 if pid is None else pid)
        return p.x, p.y

    def is_player_alive(self, pid):
        return self.get_player(pid).alive

    def get_bullets(self):
        return list(self.bullets)

    def get_grid_copy(self):
        words = self._field("grid")[self._i]
        W, H = self.width, self.height
        bits = ((words[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).reshape(-1)[:W * H]
        return [[TileColor.WHITE if bits[y * W + x] else TileColor.BLACK for x in range(W)] for y in range(H)]

    def _tile_in_bounds(self, x, y):
        return 0 <= x < self.width and 0 <= y < self.height

    def _get_tile(self, x, y):
        if not self._tile_in_bounds(x, y):
            raise IndexError(f"Tile ({x}, {y}) out of bounds")              # core.py:229
        p = y * self.width + x
        return TileColor.WHITE if (int(self._field("grid")[self._i, p >> 5]) >> (p & 31)) & 1 else TileColor.BLACK

    def _set_tile(self, x, y, color):
        if not self._tile_in_bounds(x, y):
            raise IndexError(f"Tile ({x}, {y}) out of bounds")              # core.py:235
        g = self._field("grid")
        p = y * self.width + x
        if color == TileColor.WHITE:
            g[self._i, p >> 5] |= np.uint32(1 << (p & 31))
        else:
            g[self._i, p >> 5] &= np.uint32(~(1 << (p & 31)) & 0xFFFFFFFF)
        self._batch.set_state(grid=g)

    def _is_walkable_for_player(self, x, y, player_color):
        return self._tile_in_bounds(x, y) and self._get_tile(x, y) != player_color

    def is_round_over(self):
        pl = self._field("players")[self._i]
        return not (pl[0, 4] and pl[1, 4])

    def get_winner(self):
        pl = self._field("players")[self._i]
        a1, a2 = bool(pl[0, 4]), bool(pl[1, 4])
        if a1 == a2:
            return None                                                      # still running, or a double KO (core.py:483-495)
        return PlayerId.P1 if a1 else PlayerId.P2

    # ---- the core's tick pieces, on this env only
    def _phase(self, flags, a1=0, a2=0):
        b = self._batch
        act1 = np.zeros(b.num_envs, np.uint8)
        act2 = np.zeros(b.num_envs, np.uint8)
        mask = np.zeros(b.num_envs, np.uint8)
        act1[self._i], act2[self._i], mask[self._i] = a1, a2, 1
        b.step(act1, act2, flags=flags, obs=False, mask=mask)
        b.check()

    @staticmethod
    def _action_id(action):
        if action.type == ActionType.NONE or action.direction is None:
            return 0
        return {ActionType.MOVE: 1, ActionType.SHOOT: 5, ActionType.CHARGE_SHOOT: 9}[action.type] + CODE_BY_DIR[action.direction]

    def _solo(self, pid, action_id):
        pid = PlayerId.P1 if pid is None else pid
        if pid not in (PlayerId.P1, PlayerId.P2):
            raise ValueError(f"Invalid player ID: {pid}")
        q = 0 if pid == PlayerId.P1 else 1
        before = self._batch.get_state(("players",))["players"][self._i, q].copy()
        self._phase(L.INV_PHASE_ACTIONS, action_id if q == 0 else 0, action_id if q == 1 else 0)
        after = self._batch.get_state(("players",))["players"][self._i, q]
        return before, after

    def try_move_player(self, direction, pid=None):
        before, after = self._solo(pid, 1 + CODE_BY_DIR[direction])
        return bool(before[0] != after[0] or before[1] != after[1])          # core.py:249-296: True iff the player moved

    def spawn_bullet(self, direction, pid=None):
        before, after = self._solo(pid, 5 + CODE_BY_DIR[direction])
        return bool(after[2] == before[2] - 1)                               # core.py:298-326: True iff a bullet was fired

    def spawn_wide_shot(self, pid, direction):
        before, after = self._solo(pid, 9 + CODE_BY_DIR[direction])
        return bool(after[2] == before[2] - 3)                               # core.py:328-381

    def _reload_ammo(self):
        self._phase(L.INV_PHASE_RELOAD)

    def update_bullets(self):
        self._phase(L.INV_PHASE_BULLETS)

    def step_players(self, action_p1, action_p2):
        self._phase(L.INV_PHASE_ACTIONS | L.INV_PHASE_RELOAD | L.INV_PHASE_BULLETS, self._action_id(action_p1), self._action_id(action_p2))

    def step(self, action):
        self.step_players(action, Action(ActionType.NONE, None))

    def reset(self, seed=None):
        b = self._batch
        mask = np.zeros(b.num_envs, np.uint8)
        mask[self._i] = 1
        if seed is not None:
            seeds = np.zeros(b.num_envs, np.int64)
            seeds[self._i] = int(seed)
            b.seed(seeds, mask)
        b.reset(mask)


class _EnvHandle:
    """`runner.envs[i]`: the per-env object the training loop resets (training.py:149)."""

    def __init__(self, runner, i):
        self._r, self._i = runner, i

    # the configuration fields of the reference's per-env object (SingleInversusRLEnv, env_wrappers.py:261-263)
    opponent_type = property(lambda self: self._r.opponent_type)
    difficulty = property(lambda self: self._r.difficulty)
    max_episode_steps = property(lambda self: self._r.max_episode_steps)

    @property
    def env(self):
        return _CoreView(self._r.batch, self._i)

    @property
    def step_count(self):
        return int(self._r.batch.get_state(("step_count",))["step_count"][self._i])

    @property
    def episode_return(self):
        return float(self._r.batch.get_state(("episode_return",))["episode_return"][self._i])

    def reset(self, seed=None):
        r, i = self._r, self._i
        mask = np.zeros(r.num_envs, np.uint8)
        mask[i] = 1
        if seed is not None:
            seeds = np.zeros(r.num_envs, np.int64)
            seeds[i] = int(seed)
            r.batch.seed(seeds, mask)
        r.batch.reset(mask)
        g, x = r.batch.render(0)
        return r._emit(g[i]), r._emit(x[i])

    def step(self, action_id, opponent_policy=None):
        r, i = self._r, self._i
        if not 0 <= int(action_id) <= 12:
            raise ValueError(f"Invalid action_id: {action_id}, must be 0-12")
        a1 = np.zeros(r.num_envs, np.uint8)
        a1[i] = int(action_id)
        mask = np.zeros(r.num_envs, np.uint8)
        mask[i] = 1
        a2 = r._opponent_actions(opponent_policy, only=i)
        res = r.batch.step(a1, a2, mask=mask, reward64=True)
        r.batch.check()
        info = r._info_dict(res, i)
        # the reference returns the Python-float (f64) reward here (env_wrappers.py:444)
        return (r._emit(res.obs[i]), r._emit(res.extra[i])), float(res.reward64[i].item()), bool(res.done[i].item()), info


def _info_of(f, steps, ret):
    """info dict of one env (env_wrappers.py:318-326, 360-369, 413-427) from its flag byte."""
    return {"landed_hit": bool(f & L.INV_INFO_LANDED_HIT), "got_hit": bool(f & L.INV_INFO_GOT_HIT),
            "win": bool(f & L.INV_INFO_WIN), "lose": bool(f & L.INV_INFO_LOSE),
            "episode_steps": steps, "episode_return": ret}


class MultiEnvRunner:
    """Runs `num_envs` envs as one device batch (reference: a Python loop over env objects)."""

    def __init__(self, num_envs, opponent_type="dummy", difficulty="easy", max_episode_steps=500, seed=None, *,
                 width=15, height=10, max_bullets=16, device=None, env_offset=0, opp_seed=None,
                 output="numpy", auto_reset=False):
        if output not in ("numpy", "torch"):
            raise ValueError("output must be 'numpy' or 'torch'")
        self.num_envs = int(num_envs)
        self.opponent_type, self.difficulty = opponent_type, difficulty
        self.max_episode_steps = max_episode_steps
        self.output, self.auto_reset = output, bool(auto_reset)
        base = _entropy_seed() if seed is None else int(seed)
        self.batch = DeviceEnvBatch(self.num_envs, width, height, max_bullets, max_episode_steps, opponent_type,
                                    difficulty, device, base if opp_seed is None else opp_seed, env_offset)
        # env i gets seed + global index (env_wrappers.py:463); the constructor of the reference's
        # core env already performs one reset (core.py:53)
        if seed is None:
            seeds = np.array([_entropy_seed() for _ in range(self.num_envs)], dtype=np.int64)
        else:
            seeds = base + int(env_offset) + np.arange(self.num_envs, dtype=np.int64)
        self.batch.seed(seeds)
        self.batch.reset()
        self.envs = [_EnvHandle(self, i) for i in range(self.num_envs)]
        self.episode_returns = [0.0] * self.num_envs
        self.episode_lengths = [0] * self.num_envs
        self.episode_wins = [0] * self.num_envs
        self.episode_losses = [0] * self.num_envs

    # -------------------------------------------------------------- helpers
    def _emit(self, t):
        return t.cpu().numpy() if self.output == "numpy" else t

    @staticmethod
    def _reward64(res, i):
        return res.reward64[i].item() if res.reward64 is not None else res.reward[i].item()

    @staticmethod
    def _to_host(*tensors):
        """Device tensors -> fresh NumPy arrays through pinned staging (torch's caching host allocator): all copies
        are queued, ONE synchronisation, 118 MB of observations at PCIe rate instead of a pageable memcpy.  Each
        array owns its pinned block, so holding on to earlier steps' arrays is as safe as in the reference."""
        hosts = []
        for t in tensors:
            h = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
            h.copy_(t, non_blocking=True)
            hosts.append(h)
        torch.cuda.current_stream().synchronize()
        return [h.numpy() for h in hosts]

    @staticmethod
    def _info_dict(res, i=None, host=None):
        fl, st, rt = host if host is not None else (res.info_flags.cpu().numpy(), res.episode_steps.cpu().numpy(),
                                                    res.episode_return.cpu().numpy())
        if i is not None:
            return _info_of(int(fl[i]), int(st[i]), float(rt[i]))
        # the reference's list of N dicts; plain Python lists first (one C loop each) -- 16 384 dicts per step
        return [_info_of(f, n, r) for f, n, r in zip(fl.tolist(), st.tolist(), rt.tolist())]

    def _opponent_actions(self, opponent_policy, only=None):
        if self.opponent_type == "dummy":
            return None                                                   # device dummy (env_wrappers.py:305-306)
        if self.opponent_type != "selfplay":
            raise ValueError(f"Unknown opponent_type: {self.opponent_type}")
        if opponent_policy is None:
            raise ValueError("opponent_policy required for selfplay mode")    # env_wrappers.py:308-309
        g, x = self.batch.render(1)                                       # P2's view of the pre-step state (:311)
        if getattr(opponent_policy, "batched", False):
            return opponent_policy(g, x)
        gh, xh = g.cpu().numpy(), x.cpu().numpy()
        idx = range(self.num_envs) if only is None else [only]
        a2 = np.zeros(self.num_envs, dtype=np.int64)
        for j in idx:                                                     # per-env callable, reference contract (:313)
            a2[j] = int(opponent_policy((gh[j], xh[j])))
        return a2

    # -------------------------------------------------------------- API
    def reset(self):
        """Reset all envs -> (grid[N,12,H,W], extra[N,4])."""
        self.batch.reset()
        g, x = self.batch.render(0)
        return self._emit(g), self._emit(x)

    def step(self, action_ids, opponent_policy=None):
        """-> ((grid, extra), rewards f32[N], dones bool[N], infos).

        output="numpy": host arrays and a list of N dicts, exactly like the reference.
        output="torch": device tensors; `infos` is an InfoBatch (device tensors, `infos[i]` builds the
        reference's dict on demand); one device->host round trip per step (did any episode end? + the error word)."""
        a2 = self._opponent_actions(opponent_policy)
        res = self.batch.step(action_ids, a2, auto_reset=self.auto_reset)
        if self.output == "torch":
            infos = InfoBatch(res)
            done = res.done.bool()
            any_done = bool(done.any())                                   # the step's ONE device->host round trip ...
            self.batch.check()                                            # ... the stream is drained, so the sticky error word costs a
            #                                                               4-byte copy: ValueError on a bad action id like the
            #                                                               reference (env_wrappers.py:66), also for device tensors
            if any_done:                                                  # stats only for finished envs
                idx = done.nonzero().flatten()                            # then ONE gather + D2H per field, not one per env
                for i, f, n, r in zip(idx.tolist(), res.info_flags[idx].tolist(), res.episode_steps[idx].tolist(),
                                      res.episode_return[idx].tolist()):
                    self.episode_returns[i], self.episode_lengths[i] = r, n
                    self.episode_wins[i] += int(bool(f & L.INV_INFO_WIN))
                    self.episode_losses[i] += int(bool(f & L.INV_INFO_LOSE))
            return (res.obs, res.extra), res.reward, done, infos
        self.batch.check()                                                # ValueError on a bad action id
        grid, extra, reward, done8, fl, st, rt = self._to_host(res.obs, res.extra, res.reward, res.done, res.info_flags,
                                                                res.episode_steps, res.episode_return)
        dones = done8.astype(bool)
        infos = self._info_dict(res, host=(fl, st, rt))
        for i in np.nonzero(dones)[0]:                                    # env_wrappers.py:513-519
            self.episode_returns[i] = infos[i]["episode_return"]
            self.episode_lengths[i] = infos[i]["episode_steps"]
            self.episode_wins[i] += int(infos[i]["win"])
            self.episode_losses[i] += int(infos[i]["lose"])
        return (grid, extra), reward, dones, infos


class InfoBatch:
    """Per-step info of a device batch: the reference's dict keys as device tensors; `infos[i]` -> dict."""

    def __init__(self, res):
        self.info_flags, self.episode_steps, self.episode_return = res.info_flags, res.episode_steps, res.episode_return

    def __len__(self):
        return int(self.info_flags.shape[0])

    def flag(self, bit):
        return (self.info_flags & bit) != 0

    def __getitem__(self, i):
        f = int(self.info_flags[i])
        return {"landed_hit": bool(f & L.INV_INFO_LANDED_HIT), "got_hit": bool(f & L.INV_INFO_GOT_HIT),
                "win": bool(f & L.INV_INFO_WIN), "lose": bool(f & L.INV_INFO_LOSE),
                "episode_steps": int(self.episode_steps[i]), "episode_return": float(self.episode_return[i])}


class SingleInversusRLEnv:
    """One env with the reference's reset/step API (a one-env device batch)."""

    def __init__(self, opponent_type="dummy", difficulty="easy", max_episode_steps=500, seed=None, **kw):
        self._runner = MultiEnvRunner(1, opponent_type, difficulty, max_episode_steps, seed, **kw)
        self.opponent_type, self.difficulty, self.max_episode_steps = opponent_type, difficulty, max_episode_steps

    @property
    def env(self):
        return self._runner.envs[0].env

    @property
    def step_count(self):
        return self._runner.envs[0].step_count

    @property
    def episode_return(self):
        return self._runner.envs[0].episode_return

    def reset(self, seed=None):
        return self._runner.envs[0].reset(seed)

    def step(self, action_id, opponent_policy=None):
        if not 0 <= int(action_id) <= 12:
            raise ValueError(f"Invalid action_id: {action_id}, must be 0-12")
        r = self._runner
        a2 = r._opponent_actions(opponent_policy)
        res = r.batch.step(np.array([int(action_id)], np.uint8), a2, reward64=True)
        r.batch.check()
        info = r._info_dict(res, 0)
        obs = (r._emit(res.obs[0]), r._emit(res.extra[0]))
        return obs, float(res.reward64[0].item()), bool(res.done[0].item()), info


def build_observation(env, player_id=PlayerId.P1):
    """Observation of `env` (a SingleInversusRLEnv, a runner.envs[i] handle or a MultiEnvRunner)
    from `player_id`'s perspective: (grid[12,H,W] f32, extra[4] f32)."""
    if player_id not in (PlayerId.P1, PlayerId.P2):
        raise ValueError(f"Invalid player ID: {player_id}")
    pid = 0 if player_id == PlayerId.P1 else 1
    if isinstance(env, MultiEnvRunner):
        g, x = env.batch.render(pid)
        return env._emit(g), env._emit(x)
    if isinstance(env, SingleInversusRLEnv):
        r, i = env._runner, 0
    elif isinstance(env, _EnvHandle):
        r, i = env._r, env._i
    else:
        raise TypeError("build_observation expects an env object of this package")
    g, x = r.batch.render(pid)
    return r._emit(g[i]), r._emit(x[i])
