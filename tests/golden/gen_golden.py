#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the Python reference.

Runs ONLY in the build container, where the reference checkout is mounted read-only
at /root/reference (it never travels to the GPU box, and nothing in tests/ reads it at
run time).  It imports the reference's own modules, drives them with seeded inputs and
stores inputs + outputs as small .npz files.  No reference source is copied.

    python tests/golden/gen_golden.py            # rewrites every fixture

Canonical encodings (shared with include/inversus_hip.h and oracle/):
  grid   : u32 words, bit (y*W + x) set <=> tile WHITE
  player : [x, y, ammo, reload_counter, alive]
  bullet : u16  x | y<<6 | dir<<12 | owner<<14, dir 0=UP 1=RIGHT 2=DOWN 3=LEFT, owner 0=P1 1=P2
  obs    : np.packbits(obs.reshape(-1) != 0) -- every obs value is asserted to be 0.0 or 1.0
"""
import os
import random
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = os.environ.get("INVERSUS_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

from inversus.core import InversusEnv                                   # noqa: E402
from inversus.game_types import TileColor, Direction, PlayerId, ActionType  # noqa: E402
from inversus_rl import env_wrappers as ew                              # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
K = 16
DIR_IDX = {Direction.UP: 0, Direction.RIGHT: 1, Direction.DOWN: 2, Direction.LEFT: 3}


def grid_words(env):
    W, H = env.width, env.height
    words = np.zeros((W * H + 31) // 32, dtype=np.uint32)
    for y in range(H):
        for x in range(W):
            if env.grid[y][x] == TileColor.WHITE:
                i = y * W + x
                words[i >> 5] |= np.uint32(1 << (i & 31))
    return words


def player_rows(env):
    return np.array([[p.x, p.y, p.ammo, p.reload_counter, int(p.alive)]
                     for p in (env.player1, env.player2)], dtype=np.int32)


def bullet_words(env):
    out = np.zeros(K, dtype=np.uint16)
    assert len(env.bullets) <= K, "bullet bound exceeded"
    for i, b in enumerate(env.bullets):
        owner = 0 if b.owner == PlayerId.P1 else 1
        out[i] = (b.x & 63) | ((b.y & 63) << 6) | (DIR_IDX[b.dir] << 12) | (owner << 14)
    return out


def winner_code(env):
    w = env.get_winner()
    return 0 if w is None else (1 if w == PlayerId.P1 else 2)


def obs_bits(obs_grid):
    flat = np.asarray(obs_grid).reshape(-1)
    assert flat.dtype == np.float32
    assert np.all((flat == 0.0) | (flat == 1.0))
    return np.packbits(flat != 0)


def action_stream(rng, shape, biased):
    if not biased:
        return rng.randint(0, 13, size=shape).astype(np.uint8)
    a = rng.randint(0, 13, size=shape)
    shoot = rng.randint(5, 13, size=shape)
    pick = rng.random_sample(shape) < 0.7
    return np.where(pick, shoot, a).astype(np.uint8)


# --------------------------------------------------------------------------- G1
def gen_spawn(W, H):
    seeds = list(range(64)) + [2 ** 31, 2 ** 32 - 1, 2 ** 32, 2 ** 40 + 7, -5, 2 ** 62 + 12345]
    R = 4
    pos = np.zeros((len(seeds), R, 4), np.int32)
    grid = np.zeros((len(seeds), R, (W * H + 31) // 32), np.uint32)
    for si, s in enumerate(seeds):
        env = InversusEnv(width=W, height=H, seed=s)       # reset #0 happens in __init__
        for r in range(R):
            if r > 0:
                env.reset()
            pos[si, r] = (env.player1.x, env.player1.y, env.player2.x, env.player2.y)
            grid[si, r] = grid_words(env)
    np.savez_compressed(os.path.join(OUT, f"g1_spawn_{W}x{H}.npz"),
                        seeds=np.array(seeds, np.int64), pos=pos, grid=grid, W=W, H=H)


# --------------------------------------------------------------------------- G2
def gen_core_trace(name, W, H, E, T, base_seed, act_seed, biased):
    rng = np.random.RandomState(act_seed)
    a1 = action_stream(rng, (T, E), biased)
    a2 = action_stream(rng, (T, E), biased)
    envs = [InversusEnv(width=W, height=H, seed=base_seed + i) for i in range(E)]
    GW = (W * H + 31) // 32
    grid = np.zeros((T, E, GW), np.uint32)
    players = np.zeros((T, E, 2, 5), np.int32)
    nb = np.zeros((T, E), np.int32)
    bullets = np.zeros((T, E, K), np.uint16)
    over = np.zeros((T, E), np.uint8)
    winner = np.zeros((T, E), np.uint8)
    init_players = np.stack([player_rows(e) for e in envs])
    init_grid = np.stack([grid_words(e) for e in envs])
    for t in range(T):
        for i, env in enumerate(envs):
            env.step_players(ew.discrete_to_action(int(a1[t, i])), ew.discrete_to_action(int(a2[t, i])))
            grid[t, i] = grid_words(env)
            players[t, i] = player_rows(env)
            nb[t, i] = len(env.bullets)
            bullets[t, i] = bullet_words(env)
            over[t, i] = env.is_round_over()
            winner[t, i] = winner_code(env)
            if env.is_round_over():
                env.reset()                      # training-loop style: same MT stream continues
    np.savez_compressed(os.path.join(OUT, f"g2_core_{name}.npz"), W=W, H=H, base_seed=base_seed,
                        a1=a1, a2=a2, grid=grid, players=players, nbullets=nb, bullets=bullets,
                        over=over, winner=winner, init_players=init_players, init_grid=init_grid)
    print(f"g2 {name}: resets={int(over.sum())} max_bullets={int(nb.max())}")


# --------------------------------------------------------------------------- G3
def hunt_action(env, pid):
    """An action id that closes in on the enemy: shoot along the shared row / column, else step towards alignment along the
    shorter axis (shooting the way open first: a player only walks on tiles of the other colour).  Only used to CHOOSE the
    recorded action streams of a fixture (so that a big board sees kills); parity replays the recorded ids."""
    me, en = (env.player1, env.player2) if pid == PlayerId.P1 else (env.player2, env.player1)
    dx, dy = en.x - me.x, en.y - me.y
    if dx == 0 and dy == 0:
        return 0
    if dx == 0:
        return 5 + (0 if dy < 0 else 2)
    if dy == 0:
        return 5 + (3 if dx < 0 else 1)
    if abs(dx) <= abs(dy):
        d, nx, ny = (3 if dx < 0 else 1), me.x + (-1 if dx < 0 else 1), me.y
    else:
        d, nx, ny = (0 if dy < 0 else 2), me.x, me.y + (-1 if dy < 0 else 1)
    return (1 if env._is_walkable_for_player(nx, ny, me.color) else 5) + d


def gen_wrapper_trace(name, W, H, E, T, base_seed, act_seed, biased, max_steps, hunt=(0.0, 0.0)):
    rng = np.random.RandomState(act_seed)
    a1 = action_stream(rng, (T, E), biased)
    a2 = action_stream(rng, (T, E), biased)
    hunt_draw = rng.random_sample((T, E, 2)) if max(hunt) > 0 else None
    envs = []
    for i in range(E):
        w = ew.SingleInversusRLEnv(opponent_type="selfplay", max_episode_steps=max_steps, seed=base_seed + i)
        if (W, H) != (15, 10):
            w.env = InversusEnv(width=W, height=H, seed=base_seed + i)
        envs.append(w)
    nbits = (12 * W * H + 7) // 8
    obs0 = np.zeros((E, nbits), np.uint8)
    extra0 = np.zeros((E, 4), np.float32)
    for i, w in enumerate(envs):
        g, x = ew.build_observation(w.env, PlayerId.P1)
        obs0[i], extra0[i] = obs_bits(g), x
    obs_step = np.zeros((T, E, nbits), np.uint8)      # obs returned by step()
    extra_step = np.zeros((T, E, 4), np.float32)
    obs_next = np.zeros((T, E, nbits), np.uint8)      # what the training loop carries forward
    extra_next = np.zeros((T, E, 4), np.float32)
    obs_p2 = np.zeros((T, E, nbits), np.uint8)        # P2-perspective obs of the PRE-step state
    extra_p2 = np.zeros((T, E, 4), np.float32)
    reward = np.zeros((T, E), np.float64)
    done = np.zeros((T, E), np.uint8)
    info_flags = np.zeros((T, E, 4), np.uint8)        # landed_hit, got_hit, win, lose
    ep_steps = np.zeros((T, E), np.int32)
    ep_return = np.zeros((T, E), np.float64)
    for t in range(T):
        for i, w in enumerate(envs):
            seen = {}
            if hunt_draw is not None:            # the streams are recorded below, after the loop filled them in
                if hunt_draw[t, i, 0] < hunt[0]:
                    a1[t, i] = hunt_action(w.env, PlayerId.P1)
                if hunt_draw[t, i, 1] < hunt[1]:
                    a2[t, i] = hunt_action(w.env, PlayerId.P2)

            def opp(obs, _i=i, _t=t, _seen=seen):
                _seen["g"], _seen["x"] = obs
                return int(a2[_t, _i])

            (g, x), r, d, info = w.step(int(a1[t, i]), opponent_policy=opp)
            obs_p2[t, i], extra_p2[t, i] = obs_bits(seen["g"]), seen["x"]
            obs_step[t, i], extra_step[t, i] = obs_bits(g), x
            reward[t, i], done[t, i] = r, d
            info_flags[t, i] = (info["landed_hit"], info["got_hit"], info["win"], info["lose"])
            ep_steps[t, i], ep_return[t, i] = info["episode_steps"], info["episode_return"]
            if d:
                g, x = w.reset()                 # training.py:148-151
            obs_next[t, i], extra_next[t, i] = obs_bits(g), x
    np.savez_compressed(os.path.join(OUT, f"g3_wrapper_{name}.npz"), W=W, H=H, base_seed=base_seed,
                        max_steps=max_steps, a1=a1, a2=a2, obs0=obs0, extra0=extra0,
                        obs_step=obs_step, extra_step=extra_step, obs_next=obs_next, extra_next=extra_next,
                        obs_p2=obs_p2, extra_p2=extra_p2, reward=reward, done=done,
                        info_flags=info_flags, ep_steps=ep_steps, ep_return=ep_return)
    print(f"g3 {name}: dones={int(done.sum())} wins={int(info_flags[..., 2].sum())} "
          f"losses={int(info_flags[..., 3].sum())} distinct rewards={len(np.unique(reward))}")


# --------------------------------------------------------------------------- G4
def gen_gae():
    import torch
    from inversus_rl.ppo_agent import PPOAgent
    out = {}
    rng = np.random.RandomState(7)
    for name, T, N, pdone in [("128x1", 128, 1, 0.02), ("128x16", 128, 16, 0.02),
                              ("512x4", 512, 4, 0.05), ("64x8_nodone", 64, 8, 0.0)]:
        r = (rng.randn(T, N) * np.array([1.0, 0.01, 10.0, 0.1])[rng.randint(0, 4, (T, N))]).astype(np.float32)
        v = rng.randn(T, N).astype(np.float32)
        d = (rng.random_sample((T, N)) < pdone)
        agent = PPOAgent(torch.nn.Linear(1, 1))
        # the reference loop fills the buffer env-minor inside the time loop (training.py:128-137)
        for t in range(T):
            for i in range(N):
                agent.store_step(None, None, 0, 0.0, float(v[t, i]), float(r[t, i]), bool(d[t, i]))
        adv, ret = agent.compute_advantages()
        out[f"{name}_r"], out[f"{name}_v"], out[f"{name}_d"] = r, v, d.astype(np.uint8)
        out[f"{name}_adv_flat"], out[f"{name}_ret_flat"] = adv.reshape(T, N), ret.reshape(T, N)
        # per-env: the same reference routine applied to each env's own column
        adv_pe = np.zeros((T, N), np.float32)
        ret_pe = np.zeros((T, N), np.float32)
        for i in range(N):
            agent = PPOAgent(torch.nn.Linear(1, 1))
            for t in range(T):
                agent.store_step(None, None, 0, 0.0, float(v[t, i]), float(r[t, i]), bool(d[t, i]))
            a, rr = agent.compute_advantages()
            adv_pe[:, i], ret_pe[:, i] = a, rr
        out[f"{name}_adv_env"], out[f"{name}_ret_env"] = adv_pe, ret_pe
        assert adv.dtype == np.float32 and ret.dtype == np.float32
    np.savez_compressed(os.path.join(OUT, "g4_gae.npz"), **out)


# --------------------------------------------------------------------------- G6
def gen_dummy(n_cases=3000):
    """dummy_opponent_policy as a pure function of (state, draws): the global `random`
    module's random()/shuffle() are replaced by recorded draws for the duration of a call."""
    rng = np.random.RandomState(11)
    env = InversusEnv(seed=5)
    recs = {k: [] for k in ("grid", "players", "hard", "u", "perm_a", "swap_c", "perm_b", "action")}
    HIST_REPS = 200
    hist = [np.zeros((n_cases, 13), np.uint16), np.zeros((n_cases, 13), np.uint16)]
    real_random, real_shuffle = random.random, random.shuffle

    def skewed():
        m = rng.randint(0, 3)
        return float(rng.uniform(0, 1) if m == 0 else rng.uniform(0, 0.3) if m == 1 else rng.uniform(0, 0.002))

    try:
        for c in range(n_cases):
            # wander the state with random play so grids / positions / ammo vary
            for _ in range(rng.randint(1, 6)):
                env.step_players(ew.discrete_to_action(int(rng.randint(0, 13))),
                                 ew.discrete_to_action(int(rng.randint(0, 13))))
                if env.is_round_over():
                    env.reset()
            if rng.random_sample() < 0.15:          # force alignment cases
                env.player2.x = env.player1.x
            if rng.random_sample() < 0.15:
                env.player2.y = env.player1.y
            if rng.random_sample() < 0.05:
                env.player2.ammo = 0
            if rng.random_sample() < 0.03:
                env.player2.alive = False
            hard = int(rng.randint(0, 2))
            u = [skewed(), skewed(), skewed()]
            perm_a = rng.permutation(4)
            perm_b = rng.permutation(4)
            swap_c = int(rng.randint(0, 2))
            draws = iter(u)
            state = {"n4": 0}
            a_expected = hard and (u[1] < 0.05)

            def fake_random():
                return next(draws)

            def fake_shuffle(lst):
                if len(lst) == 4:
                    perm = perm_a if (a_expected and state["n4"] == 0) else perm_b
                    state["n4"] += 1
                    lst[:] = [lst[j] for j in perm]
                elif len(lst) == 2 and swap_c:
                    lst[0], lst[1] = lst[1], lst[0]

            random.random, random.shuffle = fake_random, fake_shuffle
            act = ew.dummy_opponent_policy(env, difficulty="hard" if hard else "easy")
            random.random, random.shuffle = real_random, real_shuffle
            if act.type == ActionType.NONE:
                aid = 0
            elif act.type == ActionType.MOVE:
                aid = 1 + DIR_IDX[act.direction]
            else:
                assert act.type == ActionType.SHOOT
                aid = 5 + DIR_IDX[act.direction]
            recs["grid"].append(grid_words(env)); recs["players"].append(player_rows(env))
            recs["hard"].append(hard); recs["u"].append(u); recs["perm_a"].append(perm_a)
            recs["swap_c"].append(swap_c); recs["perm_b"].append(perm_b); recs["action"].append(aid)
            # G6 as SURVEY 8(c) words it: the action HISTOGRAM of the reference's own dummy_opponent_policy on this fixed state,
            # with the real (seeded) global `random`, both difficulties -- what the device dummy's counter-hash draws are
            # compared with by a chi-square test (tests/test_gpu_dummy_hist.py).  The global generator's state is restored.
            saved = random.getstate()
            random.seed(99991 + c)
            for di, dname in enumerate(("easy", "hard")):
                for _ in range(HIST_REPS):
                    a = ew.dummy_opponent_policy(env, difficulty=dname)
                    hid = 0 if a.type == ActionType.NONE else (1 if a.type == ActionType.MOVE else 5) + DIR_IDX.get(a.direction, 0)
                    hist[di][c, hid] += 1
            random.setstate(saved)
            env.player2.alive = True      # undo a forced death; real deaths were reset above
    finally:
        random.random, random.shuffle = real_random, real_shuffle
    arr = {k: np.array(v) for k, v in recs.items()}
    np.savez_compressed(os.path.join(OUT, "g6_dummy.npz"), **arr)
    print("g6 dummy action histogram:", np.bincount(arr["action"], minlength=13))
    # a file of its own, so that g6_dummy.npz stays byte-identical to round 1's; rows line up with g6_dummy.npz's states
    np.savez_compressed(os.path.join(OUT, "g6_dummy_hist.npz"), hist_easy=hist[0], hist_hard=hist[1], reps=np.int32(HIST_REPS))
    print("g6 reference histograms (all states): easy", hist[0].sum(0, dtype=np.int64), "hard", hist[1].sum(0, dtype=np.int64))


# --------------------------------------------------------------------------- G5
def gen_policy():
    """Reference policy graph under a fixed seed: outputs for 8 fixed observations, state_dict keys
    and per-tensor sums (the 41 MB state_dict itself is not committed)."""
    import torch
    from inversus_rl.policies import InversusCNNPolicy
    g3 = np.load(os.path.join(OUT, "g3_wrapper_15x10_uniform.npz"))
    bits = g3["obs_next"][100, :8]                       # 8 mid-game observations
    grid = np.unpackbits(bits, axis=1)[:, :1800].astype(np.float32).reshape(8, 12, 10, 15)
    extra = g3["extra_next"][100, :8].astype(np.float32)
    torch.manual_seed(0)
    pol = InversusCNNPolicy(12, 10, 15, 4)
    pol.eval()
    with torch.no_grad():
        logits, value = pol(torch.from_numpy(grid), torch.from_numpy(extra))
    sd = pol.state_dict()
    np.savez_compressed(os.path.join(OUT, "g5_policy.npz"), grid=grid, extra=extra,
                        logits=logits.numpy(), value=value.numpy(),
                        keys=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
                        sums=np.array([float(v.double().sum()) for v in sd.values()]),
                        abs_sums=np.array([float(v.double().abs().sum()) for v in sd.values()]),
                        n_params=sum(p.numel() for p in pol.parameters()))


# --------------------------------------------------------------------------- G7
def gen_arbitrary(n_cases=1500, W=15, H=10):
    """One wrapper step from ARBITRARY states, set the way the reference's own tests set them
    (env._set_tile, env.player1.x = ..., env.bullets = [Bullet(...)]): many bullets on one tile with mixed
    owners, players on top of each other / on their own colour, any ammo / reload / alive combination."""
    from inversus.game_types import Bullet
    rng = np.random.RandomState(77)
    DIRS = [Direction.UP, Direction.RIGHT, Direction.DOWN, Direction.LEFT]
    keys = ("grid", "players", "nbullets", "bullets", "steps", "epret", "prev", "a1", "a2",
            "o_grid", "o_players", "o_nbullets", "o_bullets", "o_obs", "o_extra", "o_obs_p2", "o_reward", "o_done", "o_info", "o_steps", "o_epret")
    rec = {k: [] for k in keys}
    for c in range(n_cases):
        w = ew.SingleInversusRLEnv(opponent_type="selfplay", max_episode_steps=50, seed=c)
        env = w.env
        dens = rng.choice([0.05, 0.5, 0.95])
        for y in range(H):
            for x in range(W):
                env._set_tile(x, y, TileColor.WHITE if rng.random_sample() < dens else TileColor.BLACK)
        for p in (env.player1, env.player2):
            p.x, p.y = int(rng.randint(0, W)), int(rng.randint(0, H))
            p.ammo, p.reload_counter, p.alive = int(rng.randint(0, 7)), int(rng.randint(0, 30)), bool(rng.random_sample() < 0.85)
        if rng.random_sample() < 0.1:
            env.player2.x, env.player2.y = env.player1.x, env.player1.y
        nb = int(rng.randint(0, 11))
        cx, cy = rng.randint(0, W), rng.randint(0, H)
        env.bullets = [Bullet(x=int(np.clip(cx + rng.randint(-2, 3), 0, W - 1)), y=int(np.clip(cy + rng.randint(-2, 3), 0, H - 1)),
                              dir=DIRS[rng.randint(0, 4)], owner=PlayerId.P1 if rng.randint(0, 2) == 0 else PlayerId.P2)
                       for _ in range(nb)]
        w.step_count = int(rng.randint(0, 52))
        w.episode_return = float(rng.randn())
        w.prev_p1_alive, w.prev_p2_alive = bool(rng.random_sample() < 0.9), bool(rng.random_sample() < 0.9)
        a1, a2 = int(rng.randint(0, 13)), int(rng.randint(0, 13))
        rec["grid"].append(grid_words(env)); rec["players"].append(player_rows(env)); rec["nbullets"].append(nb)
        rec["bullets"].append(bullet_words(env)); rec["steps"].append(w.step_count); rec["epret"].append(w.episode_return)
        rec["prev"].append([w.prev_p1_alive, w.prev_p2_alive]); rec["a1"].append(a1); rec["a2"].append(a2)
        (g, x), r, d, info = w.step(a1, opponent_policy=lambda obs, a=a2: a)
        g2, _ = ew.build_observation(env, PlayerId.P2)
        rec["o_grid"].append(grid_words(env)); rec["o_players"].append(player_rows(env)); rec["o_nbullets"].append(len(env.bullets))
        rec["o_bullets"].append(bullet_words(env)); rec["o_obs"].append(obs_bits(g)); rec["o_extra"].append(x)
        rec["o_obs_p2"].append(obs_bits(g2)); rec["o_reward"].append(r); rec["o_done"].append(d)
        rec["o_info"].append([info["landed_hit"], info["got_hit"], info["win"], info["lose"]])
        rec["o_steps"].append(info["episode_steps"]); rec["o_epret"].append(info["episode_return"])
    arr = {k: np.array(v) for k, v in rec.items()}
    arr["o_reward"] = arr["o_reward"].astype(np.float64); arr["epret"] = arr["epret"].astype(np.float64)
    np.savez_compressed(os.path.join(OUT, "g7_arbitrary_15x10.npz"), W=W, H=H, max_steps=50, **arr)
    print("g7: cases", n_cases, "dones", int(arr["o_done"].sum()), "max bullets after", int(arr["o_nbullets"].max()))


# --------------------------------------------------------------------------- G8
def gen_ppo_update():
    """PPOAgent.update (ppo_agent.py:159-247) on a small rollout collected by the reference's own loop
    (training.py:117-157): 4 envs x 32 steps vs the hard dummy, `torch.manual_seed(0)` policy (the G5 init).
    The update is observed, not re-stated: np.random.shuffle / torch.FloatTensor / torch.min / mse_loss /
    Categorical.entropy / clip_grad_norm_ are wrapped for the duration of the call and record what the
    reference passed through them (permutation per epoch, normalised advantages, returns, per-minibatch
    policy / value loss, entropy, pre-clip gradient norm); every wrapper returns the real result."""
    import copy
    import torch
    from inversus_rl.ppo_agent import PPOAgent
    from inversus_rl.policies import InversusCNNPolicy
    N, T = 4, 32
    random.seed(5)
    np.random.seed(11)
    torch.manual_seed(0)
    policy0 = InversusCNNPolicy(12, 10, 15, 4)
    runner = ew.MultiEnvRunner(N, "dummy", "hard", max_episode_steps=12, seed=4242)
    agent = PPOAgent(policy0)
    obs_grid, obs_extra = runner.reset()
    for _ in range(T):                                           # training.py:119-153
        actions, log_probs, values = agent.act(obs_grid, obs_extra)
        (ng, nx), rewards, dones, infos = runner.step(actions)
        for i in range(N):
            agent.store_step(obs_grid[i], obs_extra[i], int(actions[i]), float(log_probs[i]), float(values[i]),
                             float(rewards[i]), bool(dones[i]))
            if dones[i]:
                g, x = runner.envs[i].reset()
                ng[i], nx[i] = g, x
        obs_grid, obs_extra = ng, nx
    buf = dict(obs=np.stack([obs_bits(g) for g in agent.obs_grid_buffer]),
               extra=np.stack(agent.obs_extra_buffer).astype(np.float32),
               actions=np.array(agent.action_buffer, np.int64),
               log_probs=np.array(agent.log_prob_buffer, np.float32),        # float(f32) round-trips exactly
               values=np.array(agent.value_buffer, np.float32),
               rewards=np.array(agent.reward_buffer, np.float32),
               dones=np.array(agent.done_buffer, np.uint8))
    for k in ("log_probs", "values", "rewards"):
        assert np.array_equal(buf[k].astype(np.float64), np.array(getattr(agent, {"log_probs": "log_prob_buffer", "values": "value_buffer",
                                                                              "rewards": "reward_buffer"}[k]), np.float64))
    out = {"N": N, "T": T, **buf}
    adv_raw, ret_raw = agent.compute_advantages()
    out["adv_flat"], out["ret_flat"] = adv_raw, ret_raw
    sd0 = {k: v.clone() for k, v in policy0.state_dict().items()}

    for case, epochs, batch in (("e1_full", 1, N * T), ("e2_b32", 2, 32)):
        pol = InversusCNNPolicy(12, 10, 15, 4)
        pol.load_state_dict(sd0)
        ag = PPOAgent(pol, epochs=epochs, batch_size=batch)
        for name in ("obs_grid_buffer", "obs_extra_buffer", "action_buffer", "log_prob_buffer", "reward_buffer",
                     "value_buffer", "done_buffer"):
            setattr(ag, name, copy.deepcopy(getattr(agent, name)))
        rec = {"perm": [], "ft": [], "pol": [], "val": [], "ent": [], "gn": []}
        real = (np.random.shuffle, torch.FloatTensor, torch.min, torch.nn.functional.mse_loss,
                torch.distributions.Categorical.entropy, torch.nn.utils.clip_grad_norm_)

        def shuffle(x):
            real[0](x)
            rec["perm"].append(np.array(x, np.int64))

        def float_tensor(x):
            t = real[1](x)
            rec["ft"].append(t.clone())
            return t

        def tmin(a, b):
            r = real[2](a, b)
            rec["pol"].append(float(-r.mean()))
            return r

        def mse(a, b):
            r = real[3](a, b)
            rec["val"].append(float(r))
            return r

        def entropy(self):
            r = real[4](self)
            rec["ent"].append(float(r.mean()))
            return r

        def clip(params, max_norm):
            r = real[5](params, max_norm)
            rec["gn"].append(float(r))
            return r

        np.random.seed(1000 + epochs)
        np.random.shuffle, torch.FloatTensor, torch.min = shuffle, float_tensor, tmin
        torch.nn.functional.mse_loss, torch.distributions.Categorical.entropy = mse, entropy
        torch.nn.utils.clip_grad_norm_ = clip
        try:
            stats = ag.update()
        finally:
            (np.random.shuffle, torch.FloatTensor, torch.min, torch.nn.functional.mse_loss,
             torch.distributions.Categorical.entropy, torch.nn.utils.clip_grad_norm_) = real
        n_mb = epochs * -(-N * T // batch)
        assert len(rec["perm"]) == epochs and len(rec["ft"]) == 5 and all(len(rec[k]) == n_mb for k in ("pol", "val", "ent", "gn"))
        sd1 = pol.state_dict()
        out[f"{case}_epochs"], out[f"{case}_batch"] = epochs, batch
        out[f"{case}_perm"] = np.stack(rec["perm"])
        out[f"{case}_adv_norm"] = rec["ft"][3].numpy()          # (A - mean) / (std + 1e-8), ppo_agent.py:173
        out[f"{case}_returns"] = rec["ft"][4].numpy()
        for k, name in (("pol", "mb_policy_loss"), ("val", "mb_value_loss"), ("ent", "mb_entropy"), ("gn", "mb_grad_norm")):
            out[f"{case}_{name}"] = np.array(rec[k], np.float64)
        out[f"{case}_stats"] = np.array([stats["policy_loss"], stats["value_loss"], stats["entropy"]], np.float64)
        out[f"{case}_sums"] = np.array([float(v.double().sum()) for v in sd1.values()])
        out[f"{case}_abs_sums"] = np.array([float(v.double().abs().sum()) for v in sd1.values()])
        out[f"{case}_delta_sums"] = np.array([float((sd1[k].double() - sd0[k].double()).sum()) for k in sd1])
        out[f"{case}_delta_abs_sums"] = np.array([float((sd1[k].double() - sd0[k].double()).abs().sum()) for k in sd1])
        print(f"g8 {case}: stats {stats} grad norms {rec['gn'][:3]} dones {int(buf['dones'].sum())}")
    out["keys"] = np.array(list(sd0.keys()))
    np.savez_compressed(os.path.join(OUT, "g8_ppo_update.npz"), **out)


def main():
    only = set(sys.argv[1:])                     # e.g. `gen_golden.py g8 g3_32` regenerates just those
    if only:
        if "g8" in only:
            gen_ppo_update()
        if "g6" in only:
            gen_dummy()
        if "g3_32" in only:
            gen_wrapper_trace("32x32_biased", 32, 32, 8, 384, base_seed=900, act_seed=203, biased=True, max_steps=150, hunt=(0.6, 0.45))
        return
    gen_spawn(15, 10)
    gen_spawn(32, 32)
    gen_spawn(3, 3)
    gen_core_trace("15x10_uniform", 15, 10, 64, 512, base_seed=0, act_seed=100, biased=False)
    gen_core_trace("15x10_biased", 15, 10, 64, 512, base_seed=1000, act_seed=101, biased=True)
    gen_core_trace("32x32_biased", 32, 32, 16, 512, base_seed=2000, act_seed=102, biased=True)
    gen_core_trace("7x5_biased", 7, 5, 16, 256, base_seed=3000, act_seed=103, biased=True)
    gen_wrapper_trace("15x10_uniform", 15, 10, 32, 512, base_seed=0, act_seed=200, biased=False, max_steps=500)
    gen_wrapper_trace("15x10_biased", 15, 10, 32, 512, base_seed=500, act_seed=201, biased=True, max_steps=500)
    gen_wrapper_trace("15x10_timeout20", 15, 10, 16, 128, base_seed=700, act_seed=202, biased=False, max_steps=20)
    gen_wrapper_trace("32x32_biased", 32, 32, 8, 384, base_seed=900, act_seed=203, biased=True, max_steps=150, hunt=(0.6, 0.45))
    gen_gae()
    gen_dummy()
    gen_policy()
    gen_arbitrary()
    gen_ppo_update()


if __name__ == "__main__":
    main()
