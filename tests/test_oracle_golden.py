"""Pin the CPU oracle (oracle/inv_oracle.c) against the golden vectors produced by the
Python reference (tests/golden/gen_golden.py).  Bit-exact everywhere: integer state,
obs bits, f32 extras, f64 rewards, f32 GAE."""
import glob
import os
import random

import numpy as np
import pytest

from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
K = 16


def load(name):
    # materialise once: NpzFile.__getitem__ re-reads and decompresses on every access
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


# ------------------------------------------------------------------ MT19937 / CPython
@pytest.mark.parametrize("seed", [0, 1, 2, 42, 12345, 2 ** 31, 2 ** 32 - 1, 2 ** 32, 2 ** 40 + 7, -5, 2 ** 62 + 12345])
def test_mt_matches_cpython_random(seed):
    e = orc.OracleEnv(15, 10)
    e.seed(seed)
    rr = random.Random(seed)
    for _ in range(1500):                      # crosses two twists
        assert e.L.orc_mt_u32(e.e) == rr.getrandbits(32)


@pytest.mark.parametrize("n", [1, 2, 3, 8, 13, 30, 62])
def test_randbelow_matches_cpython(n):
    e = orc.OracleEnv(15, 10)
    e.seed(99)
    rr = random.Random(99)
    for _ in range(300):
        assert e.L.orc_randbelow(e.e, n) == rr.randrange(n)


# ------------------------------------------------------------------ G1 spawns
@pytest.mark.parametrize("fname", ["g1_spawn_15x10.npz", "g1_spawn_32x32.npz", "g1_spawn_3x3.npz"])
def test_spawn_golden(fname):
    g = load(fname)
    W, H = int(g["W"]), int(g["H"])
    for si, s in enumerate(g["seeds"]):
        e = orc.OracleEnv(W, H, seed=int(s))
        for r in range(g["pos"].shape[1]):
            if r > 0:
                e.reset()
            p = e.players()
            assert (p[0, 0], p[0, 1], p[1, 0], p[1, 1]) == tuple(g["pos"][si, r]), (s, r)
            assert np.array_equal(e.grid_bits(), g["grid"][si, r]), (s, r)
            assert p[0, 2] == 6 and p[1, 2] == 6 and p[0, 4] == 1 and p[1, 4] == 1


def test_survey_spawn_constants():
    # SURVEY.md section 8a R1 goldens
    want = {0: (7, 7, 1, 5), 1: (3, 2, 8, 8), 2: (1, 2, 2, 6), 3: (4, 3, 6, 8), 42: (11, 2, 1, 5),
            12345: (7, 1, 13, 5), 2 ** 31: (6, 7, 9, 3), 2 ** 40 + 7: (10, 8, 5, 3)}
    for s, w in want.items():
        p = orc.OracleEnv(15, 10, seed=s).players()
        assert (p[0, 0], p[0, 1], p[1, 0], p[1, 1]) == w


def test_bad_dims_raise():
    with pytest.raises(ValueError):
        orc.OracleEnv(2, 10)
    with pytest.raises(ValueError):
        orc.OracleEnv(15, 2)


# ------------------------------------------------------------------ G2 core traces
@pytest.mark.parametrize("fname", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g2_core_*.npz"))))
def test_core_trace_golden(fname):
    g = load(fname)
    W, H, base = int(g["W"]), int(g["H"]), int(g["base_seed"])
    T, E = g["a1"].shape
    envs = [orc.OracleEnv(W, H, seed=base + i) for i in range(E)]
    for i, e in enumerate(envs):
        assert np.array_equal(e.players(), g["init_players"][i])
        assert np.array_equal(e.grid_bits(), g["init_grid"][i])
    for t in range(T):
        for i, e in enumerate(envs):
            e.step_players(g["a1"][t, i], g["a2"][t, i])
            assert np.array_equal(e.players(), g["players"][t, i]), (t, i)
            assert e.s.nb == g["nbullets"][t, i], (t, i)
            assert np.array_equal(e.bullets_packed(K), g["bullets"][t, i]), (t, i)
            assert np.array_equal(e.grid_bits(), g["grid"][t, i]), (t, i)
            assert e.is_round_over() == bool(g["over"][t, i])
            assert e.get_winner() == g["winner"][t, i]
            if e.is_round_over():
                e.L.orc_core_reset(e.e)


# ------------------------------------------------------------------ G3 wrapper traces
def _bits(obs):
    return np.packbits(obs.reshape(-1) != 0)


@pytest.mark.parametrize("fname", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g3_wrapper_*.npz"))))
def test_wrapper_trace_golden(fname):
    g = load(fname)
    W, H, base, max_steps = int(g["W"]), int(g["H"]), int(g["base_seed"]), int(g["max_steps"])
    T, E = g["a1"].shape
    envs = [orc.OracleEnv(W, H, seed=base + i, max_episode_steps=max_steps) for i in range(E)]
    for i, e in enumerate(envs):
        o, x = e.build_observation(0)
        assert np.array_equal(_bits(o), g["obs0"][i])
        assert np.array_equal(x.view(np.uint32), g["extra0"][i].view(np.uint32))
    for t in range(T):
        for i, e in enumerate(envs):
            o2, x2 = e.build_observation(1)
            assert np.array_equal(_bits(o2), g["obs_p2"][t, i]), (t, i)
            assert np.array_equal(x2.view(np.uint32), g["extra_p2"][t, i].view(np.uint32))
            r, d, info = e.step(g["a1"][t, i], g["a2"][t, i])
            o, x = e.build_observation(0)
            assert set(np.unique(o)) <= {0.0, 1.0}
            assert np.array_equal(_bits(o), g["obs_step"][t, i]), (t, i)
            assert np.array_equal(x.view(np.uint32), g["extra_step"][t, i].view(np.uint32))
            # f64 reward: identical bits, and therefore identical f32 after rounding
            assert np.float64(r).view(np.uint64) == g["reward"][t, i].view(np.uint64), (t, i, r, g["reward"][t, i])
            assert d == bool(g["done"][t, i])
            assert (info.landed_hit, info.got_hit, info.win, info.lose) == tuple(g["info_flags"][t, i])
            assert info.episode_steps == g["ep_steps"][t, i]
            assert np.float64(info.episode_return).view(np.uint64) == g["ep_return"][t, i].view(np.uint64)
            if d:
                e.reset()
                o, x = e.build_observation(0)
            assert np.array_equal(_bits(o), g["obs_next"][t, i]), (t, i)
            assert np.array_equal(x.view(np.uint32), g["extra_next"][t, i].view(np.uint32))


# ------------------------------------------------------------------ G4 GAE
@pytest.mark.parametrize("name", ["128x1", "128x16", "512x4", "64x8_nodone"])
def test_gae_golden(name):
    g = load("g4_gae.npz")
    r, v, d = g[f"{name}_r"], g[f"{name}_v"], g[f"{name}_d"]
    adv, ret = orc.gae_flat(r.reshape(-1), v.reshape(-1), d.reshape(-1))
    assert np.array_equal(adv.view(np.uint32), g[f"{name}_adv_flat"].reshape(-1).view(np.uint32))
    assert np.array_equal(ret.view(np.uint32), g[f"{name}_ret_flat"].reshape(-1).view(np.uint32))
    adv, ret = orc.gae_per_env(r, v, d)
    assert np.array_equal(adv.view(np.uint32), g[f"{name}_adv_env"].view(np.uint32))
    assert np.array_equal(ret.view(np.uint32), g[f"{name}_ret_env"].view(np.uint32))


# ------------------------------------------------------------------ G6 dummy opponent
def test_dummy_decide_golden():
    g = load("g6_dummy.npz")
    e = orc.OracleEnv(15, 10, seed=0)
    n = len(g["action"])
    for c in range(n):
        e.set_grid_bits(g["grid"][c])
        for pid in range(2):
            x, y, ammo, rel, alive = g["players"][c, pid]
            e.set_player(pid, int(x), int(y), int(ammo), int(rel), int(alive))
        a = e.dummy_decide(int(g["hard"][c]), [float(u) for u in g["u"][c]],
                           [int(v) for v in g["perm_a"][c]], int(g["swap_c"][c]),
                           [int(v) for v in g["perm_b"][c]])
        assert a == g["action"][c], c


def test_dummy_device_statistics():
    """The counter-RNG dummy is not bit-comparable with the reference's global `random`;
    check the rates the reference's constants imply (env_wrappers.py:81-89)."""
    e = orc.OracleEnv(15, 10, seed=1)
    # aligned, ammo>0, hard: P(shoot) = 0.2
    shots = 0
    n = 20000
    for i in range(n):
        e.set_player(0, 3, 3, 6, 0, 1)
        e.set_player(1, 3, 7, 6, 0, 1)
        a = e.dummy_device(True, 77, 5)
        shots += (a == 5)          # SHOOT UP (p1.y < p2.y)
    assert abs(shots / n - 0.2) < 0.01
    # easy: moves with probability 0.001
    moves = 0
    for i in range(n):
        a = e.dummy_device(False, 77, 6)
        moves += (a != 0)
    assert moves / n < 0.004


# ------------------------------------------------------------------ G7 arbitrary (unreachable) states
def test_arbitrary_states_golden():
    g = load("g7_arbitrary_15x10.npz")
    W, H = int(g["W"]), int(g["H"])
    n = len(g["a1"])
    e = orc.OracleEnv(W, H, seed=0, max_episode_steps=int(g["max_steps"]))
    for c in range(n):
        e.set_grid_bits(g["grid"][c])
        for pid in range(2):
            e.set_player(pid, *[int(v) for v in g["players"][c, pid]])
        e.set_bullets([orc.unpack_bullet(w) for w in g["bullets"][c, :int(g["nbullets"][c])]])
        e.s.step_count = int(g["steps"][c]); e.s.episode_return = float(g["epret"][c])
        e.s.prev_alive[0], e.s.prev_alive[1] = int(g["prev"][c, 0]), int(g["prev"][c, 1])
        r, d, info = e.step(int(g["a1"][c]), int(g["a2"][c]))
        assert np.array_equal(e.grid_bits(), g["o_grid"][c]), c
        assert np.array_equal(e.players(), g["o_players"][c]), c
        assert e.s.nb == g["o_nbullets"][c] and np.array_equal(e.bullets_packed(K), g["o_bullets"][c]), c
        o, x = e.build_observation(0)
        assert np.array_equal(_bits(o), g["o_obs"][c]) and np.array_equal(x.view(np.uint32), g["o_extra"][c].view(np.uint32)), c
        assert np.array_equal(_bits(e.build_observation(1)[0]), g["o_obs_p2"][c]), c
        assert np.float64(r).view(np.uint64) == g["o_reward"][c].view(np.uint64) and d == bool(g["o_done"][c]), c
        assert (info.landed_hit, info.got_hit, info.win, info.lose) == tuple(g["o_info"][c]), c
        assert info.episode_steps == g["o_steps"][c]
        assert np.float64(info.episode_return).view(np.uint64) == np.float64(g["o_epret"][c]).view(np.uint64), c
