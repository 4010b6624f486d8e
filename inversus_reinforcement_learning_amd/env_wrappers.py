"""Drop-in for the reference's RL env boundary (inversus_rl/env_wrappers.py), backed by the
HIP kernels: same class / function names, argument order, return shapes and exceptions.

    MultiEnvRunner(num_envs, opponent_type, difficulty, max_episode_steps, seed)   :447-528
    SingleInversusRLEnv(opponent_type, difficulty, max_episode_steps, seed)        :248-444
    build_observation(env, player_id)                                              :173-245
    discrete_to_action(action_id)                                                  :20-66

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
from .vocab import Action, ActionType, Bullet, PlayerId, PlayerState, TileColor, DIR_BY_CODE, TYPE_BY_CODE

MAX_AMMO = 6


def discrete_to_action(action_id):
    """0 NONE; 1-4 MOVE, 5-8 SHOOT, 9-12 CHARGE_SHOOT, each in the order UP, RIGHT, DOWN, LEFT."""
    a = int(action_id)
    if not 0 <= a <= 12:
        raise ValueError(f"Invalid action_id: {action_id}, must be 0-12")
    if a == 0:
        return Action(ActionType.NONE, None)
    return Action(TYPE_BY_CODE[(a + 3) >> 2], DIR_BY_CODE[(a - 1) & 3])


def _entropy_seed():
    return int.from_bytes(os.urandom(8), "little") >> 1


class _CoreView:
    """Read-only snapshot of one env's core state with the reference's field names
    (InversusEnv: width, height, grid, player1, player2, bullets, get_player, ...)."""

    def __init__(self, batch, i, state):
        self.width, self.height = batch.W, batch.H
        self.player_color = TileColor.BLACK
        W, H = batch.W, batch.H
        words = state["grid"][i]
        bits = ((words[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).reshape(-1)[:W * H]
        self.grid = [[TileColor.WHITE if bits[y * W + x] else TileColor.BLACK for x in range(W)] for y in range(H)]
        pl = state["players"][i]
        mk = lambda q, pid, col: PlayerState(pid, int(pl[q, 0]), int(pl[q, 1]), col, int(pl[q, 2]), MAX_AMMO,
                                             int(pl[q, 3]), bool(pl[q, 4]))
        self.player1 = mk(0, PlayerId.P1, TileColor.BLACK)
        self.player2 = mk(1, PlayerId.P2, TileColor.WHITE)
        self.bullets = []
        for w in state["bullets"][i][:int(state["nbullets"][i])]:
            w = int(w)
            self.bullets.append(Bullet(w & 63, (w >> 6) & 63, DIR_BY_CODE[(w >> 12) & 3],
                                       PlayerId.P2 if (w >> 14) & 1 else PlayerId.P1))

    def get_player(self, pid):
        if pid == PlayerId.P1:
            return self.player1
        if pid == PlayerId.P2:
            return self.player2
        raise ValueError(f"Invalid player ID: {pid}")

    def get_bullets(self):
        return list(self.bullets)

    def get_grid_copy(self):
        return [row[:] for row in self.grid]

    def is_round_over(self):
        return not (self.player1.alive and self.player2.alive)

    def get_winner(self):
        if not self.is_round_over():
            return None
        if not self.player1.alive and self.player2.alive:
            return PlayerId.P2
        if not self.player2.alive and self.player1.alive:
            return PlayerId.P1
        return None


class _EnvHandle:
    """`runner.envs[i]`: the per-env object the training loop resets (training.py:149)."""

    def __init__(self, runner, i):
        self._r, self._i = runner, i

    @property
    def env(self):
        return _CoreView(self._r.batch, self._i, self._r.batch.get_state())

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
        reference's dict on demand) and nothing synchronises unless an episode ended this step."""
        a2 = self._opponent_actions(opponent_policy)
        res = self.batch.step(action_ids, a2, auto_reset=self.auto_reset)
        if self.output == "torch":
            infos = InfoBatch(res)
            done = res.done.bool()
            if bool(done.any()):                                          # one scalar D2H; stats only for finished envs
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
