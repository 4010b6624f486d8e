"""GPU parity tests: the HIP path (through the C ABI, via DeviceEnvBatch) against
  (a) the golden vectors produced by the Python reference (tests/golden/),
  (b) the CPU oracle on seeded random inputs,
  (c) size-independent properties at the BASELINE.json size (16 384 envs).
Bar: bit-exact for all integer state, observation bits, f32 extras, f64 rewards."""
import glob
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("both_step_kernels")]

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CORE = 7      # INV_PHASE_ACTIONS | RELOAD | BULLETS (no wrapper bookkeeping)


def load(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def make(n, W=15, H=10, **kw):
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    return DeviceEnvBatch(n, W, H, **kw)


def obs_bits(t):
    """packbits of a [N,12,H,W] device tensor, asserting every value is exactly 0.0 or 1.0."""
    a = t.cpu().numpy()
    assert np.all((a == 0.0) | (a == 1.0))
    return np.packbits(a.reshape(a.shape[0], -1) != 0, axis=1)


def bits_eq(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


# ------------------------------------------------------------------ G1: seeding + spawns
@pytest.mark.parametrize("fname", ["g1_spawn_15x10.npz", "g1_spawn_32x32.npz", "g1_spawn_3x3.npz"])
def test_spawn_golden(torch_cuda, fname):
    g = load(fname)
    W, H = int(g["W"]), int(g["H"])
    b = make(len(g["seeds"]), W, H, opponent="selfplay")
    b.seed(g["seeds"])
    for r in range(g["pos"].shape[1]):
        b.reset()
        s = b.get_state()
        pos = np.concatenate([s["players"][:, 0, :2], s["players"][:, 1, :2]], axis=1)
        assert np.array_equal(pos, g["pos"][:, r]), r
        assert np.array_equal(s["grid"], g["grid"][:, r]), r
        assert np.all(s["players"][:, :, 2] == 6) and np.all(s["players"][:, :, 4] == 1)
        assert np.all(s["nbullets"] == 0) and np.all(s["step_count"] == 0)


def test_mt_state_matches_cpython(torch_cuda):
    import random
    seeds = np.array([0, 1, 42, 2 ** 31, 2 ** 32 - 1, 2 ** 32, 2 ** 40 + 7, -5, 2 ** 62 + 12345], dtype=np.int64)
    b = make(len(seeds), opponent="selfplay")
    b.seed(seeds)
    s = b.get_state(("mt", "mt_index"))
    for i, sd in enumerate(seeds):
        st = random.Random(int(sd)).getstate()[1]
        assert np.array_equal(s["mt"][i], np.array(st[:624], dtype=np.uint32)), sd
        assert s["mt_index"][i] == st[624] == 624


def test_many_resets_cross_mt_twists(torch_cuda):
    """>= 3 regenerations of the 624-word block per env, against the oracle."""
    from oracle import oracle as orc
    n = 64
    seeds = 9000 + np.arange(n, dtype=np.int64)
    b = make(n, opponent="selfplay")
    b.seed(seeds)
    ob = orc.OracleBatch(n)
    ob.seed_reset(seeds)           # oracle: seed + first reset
    b.reset()
    for r in range(400):
        s = b.get_state()
        o = ob.state(b.K)
        assert np.array_equal(s["players"], o["players"]), r
        assert np.array_equal(s["grid"], o["grid"]), r
        b.reset()
        for i in range(n):
            ob.L.orc_wrapper_reset(ob.L.orc_batch_at(ob.envs, i))
    idx = b.get_state(("mt_index",))["mt_index"]
    assert np.all((idx >= 0) & (idx <= 624))


# ------------------------------------------------------------------ G2: core traces
@pytest.mark.parametrize("fname", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g2_core_*.npz"))))
def test_core_trace_golden(torch_cuda, fname):
    g = load(fname)
    W, H, base = int(g["W"]), int(g["H"]), int(g["base_seed"])
    T, E = g["a1"].shape
    b = make(E, W, H, opponent="selfplay")
    b.seed(base + np.arange(E, dtype=np.int64))
    b.reset()
    s = b.get_state()
    assert np.array_equal(s["players"], g["init_players"]) and np.array_equal(s["grid"], g["init_grid"])
    for t in range(T):
        b.step(g["a1"][t], g["a2"][t], flags=CORE, obs=False)
        s = b.get_state()
        assert np.array_equal(s["players"], g["players"][t]), t
        assert np.array_equal(s["nbullets"], g["nbullets"][t]), t
        assert np.array_equal(s["bullets"], g["bullets"][t]), t
        assert np.array_equal(s["grid"], g["grid"][t]), t
        over = ~((s["players"][:, 0, 4] == 1) & (s["players"][:, 1, 4] == 1))
        assert np.array_equal(over.astype(np.uint8), g["over"][t])
        if over.any():
            b.reset(over.astype(np.uint8))
    b.check()


# ------------------------------------------------------------------ G3: wrapper traces
@pytest.mark.parametrize("auto_reset", [True, False])
@pytest.mark.parametrize("fname", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g3_wrapper_*.npz"))))
def test_wrapper_trace_golden(torch_cuda, fname, auto_reset):
    g = load(fname)
    W, H, base, max_steps = int(g["W"]), int(g["H"]), int(g["base_seed"]), int(g["max_steps"])
    T, E = g["a1"].shape
    b = make(E, W, H, opponent="selfplay", max_episode_steps=max_steps)
    b.seed(base + np.arange(E, dtype=np.int64))
    b.reset()
    o, x = b.render(0)
    assert np.array_equal(obs_bits(o), g["obs0"]) and bits_eq(x.cpu().numpy(), g["extra0"])
    for t in range(T):
        o2, x2 = b.render(1)                         # P2's view of the pre-step state
        assert np.array_equal(obs_bits(o2), g["obs_p2"][t]), t
        assert bits_eq(x2.cpu().numpy(), g["extra_p2"][t]), t
        r = b.step(g["a1"][t], g["a2"][t], auto_reset=auto_reset, obs=True, obs_p2=True, reward64=True)
        assert bits_eq(r.reward64.cpu().numpy(), g["reward"][t]), t
        assert bits_eq(r.reward.cpu().numpy(), g["reward"][t].astype(np.float32)), t
        done = r.done.cpu().numpy()
        assert np.array_equal(done, g["done"][t]), t
        fl = r.info_flags.cpu().numpy()
        want = g["info_flags"][t]
        assert np.array_equal(fl, want[:, 0] | (want[:, 1] << 1) | (want[:, 2] << 2) | (want[:, 3] << 3)), t
        assert np.array_equal(r.episode_steps.cpu().numpy(), g["ep_steps"][t]), t
        assert bits_eq(r.episode_return.cpu().numpy(), g["ep_return"][t]), t
        if auto_reset:
            assert np.array_equal(obs_bits(r.obs), g["obs_next"][t]), t
            assert bits_eq(r.extra.cpu().numpy(), g["extra_next"][t]), t
        else:
            assert np.array_equal(obs_bits(r.obs), g["obs_step"][t]), t
            assert bits_eq(r.extra.cpu().numpy(), g["extra_step"][t]), t
            if done.any():
                b.reset(done)
            o, x = b.render(0)
            assert np.array_equal(obs_bits(o), g["obs_next"][t]), t
            assert bits_eq(x.cpu().numpy(), g["extra_next"][t]), t
        if t + 1 < T and auto_reset:
            # the P2 obs emitted by the step is the next step's opponent input
            assert np.array_equal(obs_bits(r.obs_p2), g["obs_p2"][t + 1]), t
    b.check()


# ------------------------------------------------------------------ oracle cross-check, seeded random play
def _compare_step(b, ob, r, o_obs, o_extra, o_r64, o_done, o_info, t):
    assert bits_eq(r.reward64.cpu().numpy(), o_r64), t
    assert np.array_equal(r.done.cpu().numpy(), o_done), t
    fl = np.array([i.landed_hit | (i.got_hit << 1) | (i.win << 2) | (i.lose << 3) for i in o_info], np.uint8)
    assert np.array_equal(r.info_flags.cpu().numpy(), fl), t
    assert np.array_equal(r.episode_steps.cpu().numpy(), np.array([i.episode_steps for i in o_info], np.int32)), t
    assert bits_eq(r.episode_return.cpu().numpy(), np.array([i.episode_return for i in o_info])), t
    assert bits_eq(r.obs.cpu().numpy(), o_obs), t
    assert bits_eq(r.extra.cpu().numpy(), o_extra), t
    s, o = b.get_state(), ob.state(b.K)
    for k in ("grid", "players", "nbullets", "bullets", "step_count", "episode_return", "prev_alive"):
        assert bits_eq(s[k], o[k]), (t, k)


@pytest.mark.parametrize("W,H,n,T,hard", [(15, 10, 2048, 160, True), (15, 10, 2048, 160, False),
                                          (32, 32, 512, 120, True), (5, 4, 256, 200, True), (64, 64, 64, 100, True)])
def test_random_play_vs_oracle_dummy_opponent(torch_cuda, W, H, n, T, hard):
    """vs_dummy config: a1 from the synthetic action stream, a2 from the device dummy; every
    output and the whole state compared with the oracle each step (auto-reset on)."""
    from oracle import oracle as orc
    seeds = 1234 + np.arange(n, dtype=np.int64)
    off = 100
    b = make(n, W, H, opponent="dummy", difficulty="hard" if hard else "easy", opp_seed=77, env_offset=off,
             max_episode_steps=60)
    b.seed(seeds)
    b.reset()
    ob = orc.OracleBatch(n, W, H, max_episode_steps=60)
    ob.seed_reset(seeds)
    L = orc.lib()
    for t in range(T):
        a1 = b.uniform_actions(5, t)
        a1h = a1.cpu().numpy()
        want = np.array([L.orc_uniform_action(5, off + i, t) for i in range(0, n, max(1, n // 64))], np.uint8)
        assert np.array_equal(a1h[::max(1, n // 64)], want)
        r = b.step(a1, None, auto_reset=True, reward64=True)
        o_obs, o_extra, _, o_r64, o_done, o_info = ob.step(a1h, None, hard=hard, opp_seed=77, env_offset=off, nthreads=8)
        if t % 8 == 0 or t == T - 1:
            _compare_step(b, ob, r, o_obs, o_extra, o_r64, o_done, o_info, t)
        else:
            assert bits_eq(r.reward64.cpu().numpy(), o_r64), t
    b.check()


def test_biased_play_vs_oracle_many_bullets(torch_cuda):
    """70 % shoot / charge for both players: merges, cancels, kills; compared every step."""
    from oracle import oracle as orc
    n, T = 1024, 200
    rng = np.random.RandomState(3)
    seeds = 50 + np.arange(n, dtype=np.int64)
    b = make(n, opponent="selfplay")
    b.seed(seeds)
    b.reset()
    ob = orc.OracleBatch(n)
    ob.seed_reset(seeds)
    for t in range(T):
        a = [np.where(rng.random_sample(n) < 0.7, rng.randint(5, 13, n), rng.randint(0, 13, n)).astype(np.uint8) for _ in range(2)]
        r = b.step(a[0], a[1], auto_reset=True, reward64=True)
        o_obs, o_extra, _, o_r64, o_done, o_info = ob.step(a[0], a[1], nthreads=8)
        _compare_step(b, ob, r, o_obs, o_extra, o_r64, o_done, o_info, t)
    b.check()


def test_rollout_equals_single_steps(torch_cuda):
    """inv_rollout (T steps in one launch) == T inv_step launches."""
    torch = torch_cuda
    n, T = 512, 40
    seeds = 7 + np.arange(n, dtype=np.int64)
    rng = np.random.RandomState(1)
    a1 = rng.randint(0, 13, (T, n)).astype(np.uint8)
    outs = []
    for mode in ("steps", "rollout"):
        b = make(n, opponent="dummy", difficulty="hard", opp_seed=3)
        b.seed(seeds)
        b.reset()
        if mode == "steps":
            rs = [b.step(a1[t], None, auto_reset=True, obs_p2=True, reward64=True) for t in range(T)]
            pack = {k: torch.stack([getattr(r, k) for r in rs]) for k in rs[0].__slots__}
        else:
            r = b.rollout(a1, None, obs_p2=True, reward64=True)
            pack = {k: getattr(r, k) for k in r.__slots__}
        pack["state"] = b.get_state()
        outs.append(pack)
        b.check()
    for k in outs[0]:
        if k == "state":
            for f in outs[0][k]:
                assert bits_eq(outs[0][k][f], outs[1][k][f]), f
        else:
            assert torch.equal(outs[0][k], outs[1][k]), k


def test_sharding_is_invisible(torch_cuda):
    """Two half-shards with env_offset reproduce one full batch (multi-GPU partitioning rule)."""
    torch = torch_cuda
    n, T = 256, 50
    seeds = 11 + np.arange(n, dtype=np.int64)
    full = make(n, opponent="dummy", difficulty="hard", opp_seed=9)
    halves = [make(n // 2, opponent="dummy", difficulty="hard", opp_seed=9, env_offset=o) for o in (0, n // 2)]
    full.seed(seeds); full.reset()
    for h, sl in zip(halves, (slice(0, n // 2), slice(n // 2, n))):
        h.seed(seeds[sl]); h.reset()
    for t in range(T):
        rf = full.step(full.uniform_actions(1, t), None, auto_reset=True)
        rh = [h.step(h.uniform_actions(1, t), None, auto_reset=True) for h in halves]
        assert torch.equal(rf.obs, torch.cat([r.obs for r in rh]))
        assert torch.equal(rf.reward, torch.cat([r.reward for r in rh]))
        assert torch.equal(rf.done, torch.cat([r.done for r in rh]))


# ------------------------------------------------------------------ error behaviour
def test_errors(torch_cuda):
    with pytest.raises(ValueError):
        make(4, 2, 10)                                   # empty range for randrange(), core.py:69-70
    b = make(4, opponent="selfplay")
    b.seed(np.arange(4)); b.reset()
    with pytest.raises(ValueError):
        b.step(np.array([0, 13, 0, 0]), np.zeros(4))     # host-side id check
    with pytest.raises(ValueError):
        b.step(np.zeros(4), None)                        # selfplay without opponent actions
    import torch
    s0 = b.get_state()
    b.step(torch.tensor([0, 200, 0, 0], dtype=torch.uint8, device="cuda"), torch.zeros(4, dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError):
        b.check()                                        # device-side flag
    b.check()                                            # cleared
    s1 = b.get_state()
    assert np.array_equal(s0["players"][1], s1["players"][1]) and s1["step_count"][1] == 0   # bad env untouched
    assert s1["step_count"][0] == 1


# ------------------------------------------------------------------ full size (BASELINE.json: 16 384 envs)
def test_full_size_properties(torch_cuda):
    torch = torch_cuda
    n, T = 16384, 64
    seeds = np.arange(n, dtype=np.int64)
    b = make(n, opponent="dummy", difficulty="hard", opp_seed=1)
    b.seed(seeds); b.reset()
    ndone = 0
    for t in range(T):
        r = b.step(b.uniform_actions(1234, t), None, auto_reset=True, obs_p2=True)
        o = r.obs
        assert torch.all((o == 0) | (o == 1))
        assert torch.all(o[:, 0] + o[:, 1] == 1)                            # tests/test_rl_env_wrapper.py:118-121
        assert torch.all(o[:, 2].sum((1, 2)) == r.extra[:, 2])              # own marker iff alive
        assert torch.all(o[:, 3].sum((1, 2)) == r.extra[:, 3])
        # P2's view: same tiles, players and bullet planes swapped
        p = r.obs_p2
        assert torch.equal(p[:, 0:2], o[:, 0:2]) and torch.equal(p[:, 2], o[:, 3]) and torch.equal(p[:, 3], o[:, 2])
        assert torch.equal(p[:, 4:8], o[:, 8:12]) and torch.equal(p[:, 8:12], o[:, 4:8])
        assert torch.equal(r.extra_p2, r.extra[:, [1, 0, 3, 2]])
        ndone += int(r.done.sum())
    s = b.get_state()
    assert np.array_equal(o[:, 4:12].sum((1, 2, 3)).cpu().numpy().astype(np.int32), s["nbullets"])   # merged: one bullet per tile+owner
    assert s["nbullets"].max() <= 12                                        # SURVEY 8a R0 bound on 15x10
    assert ndone > 0
    # determinism: a second run from the same seeds gives the same final state
    b2 = make(n, opponent="dummy", difficulty="hard", opp_seed=1)
    b2.seed(seeds); b2.reset()
    for t in range(T):
        b2.step(b2.uniform_actions(1234, t), None, auto_reset=True, obs=False)
    s2 = b2.get_state()
    for k in s:
        assert bits_eq(s[k], s2[k]), k
    # against the oracle at full size (state only; OpenMP oracle)
    from oracle import oracle as orc
    ob = orc.OracleBatch(n)
    ob.seed_reset(seeds)
    L = orc.lib()
    a = np.empty(n, np.uint8)
    for t in range(T):
        a[:] = b2.uniform_actions(1234, t).cpu().numpy()
        ob.step(a, None, hard=True, opp_seed=1, want_obs=False, nthreads=16)
    gw = b.GW
    grid = np.zeros((n, gw), np.uint32)
    for i in range(0, n, 37):
        e = ob.env(i)
        assert np.array_equal(e.grid_bits(), s["grid"][i]) and np.array_equal(e.players(), s["players"][i]), i
        assert e.s.step_count == s["step_count"][i] and e.s.episode_return == s["episode_return"][i]
    b.check()


# ------------------------------------------------------------------ ragged sizes, other K, overflow
@pytest.mark.parametrize("n,W,H,K", [(1, 15, 10, 16), (3, 15, 10, 16), (5, 32, 32, 16), (7, 9, 4, 12), (130, 15, 10, 32),
                                     (6, 64, 3, 64), (9, 3, 64, 16)])
def test_ragged_batches_and_other_bullet_capacities(torch_cuda, n, W, H, K):
    """N not a multiple of the 4 envs per block, K != 16, extreme aspect ratios (generic dims path)."""
    from oracle import oracle as orc
    rng = np.random.RandomState(n * 31 + W)
    seeds = 77 + np.arange(n, dtype=np.int64)
    b = make(n, W, H, max_bullets=K, opponent="selfplay", max_episode_steps=40)
    b.seed(seeds); b.reset()
    ob = orc.OracleBatch(n, W, H, max_episode_steps=40)
    ob.seed_reset(seeds)
    for t in range(80):
        a = [np.where(rng.random_sample(n) < 0.6, rng.randint(5, 13, n), rng.randint(0, 13, n)).astype(np.uint8) for _ in range(2)]
        r = b.step(a[0], a[1], auto_reset=True, obs_p2=True, reward64=True)
        o_obs, o_extra, _, o_r64, o_done, o_info = ob.step(a[0], a[1])
        _compare_step(b, ob, r, o_obs, o_extra, o_r64, o_done, o_info, t)
        g2, x2 = ob.observe(1)
        assert bits_eq(r.obs_p2.cpu().numpy(), g2) and bits_eq(r.extra_p2.cpu().numpy(), x2)
    b.check()


def test_bullet_overflow_is_reported(torch_cuda):
    """K = 2 slots: the third live bullet raises the sticky overflow error instead of corrupting state."""
    from inversus_reinforcement_learning_amd import _lib as L
    b = make(4, 15, 10, max_bullets=2, opponent="selfplay")
    b.seed(np.arange(4)); b.reset()
    p = b.get_state(("players",))["players"]
    p[:, 0, :2] = (1, 5); p[:, 1, :2] = (13, 8)
    b.set_state(players=p)
    b.step(np.full(4, 10, np.uint8), np.zeros(4, np.uint8), obs=False)      # CHARGE RIGHT: 3 bullets into 2 slots
    with pytest.raises(L.InversusError):
        b.check()
    assert np.all(b.get_state(("nbullets",))["nbullets"] <= 2)


def test_stepping_after_round_over_matches_reference_semantics(torch_cuda):
    """Without auto-reset a finished env keeps returning done and the terminal bonus (env_wrappers.py:408-422)."""
    from oracle import oracle as orc
    b = make(2, opponent="selfplay")
    b.seed(np.array([5, 6])); b.reset()
    envs = [orc.OracleEnv(15, 10, seed=s) for s in (5, 6)]
    p = b.get_state(("players",))["players"]
    p[:, 1, 4] = 0                                    # P2 dead in both envs
    b.set_state(players=p)
    for e in envs:
        e.set_player(1, alive=0)
    for t in range(3):
        r = b.step(np.array([1, 6], np.uint8), np.zeros(2, np.uint8), reward64=True)
        for i, e in enumerate(envs):
            ro, do, info = e.step([1, 6][i], 0)
            assert r.reward64[i].item() == ro and bool(r.done[i]) == do == True
            assert bool(r.info_flags[i] & 4) == bool(info.win)


def test_step_is_hip_graph_capturable(torch_cuda):
    """inv_step does no allocation / synchronisation, so a chain of steps can be captured in a HIP
    graph and replayed (launch-bound inner loops); the replay equals eager stepping."""
    torch = torch_cuda
    n, T = 1024, 6
    seeds = 300 + np.arange(n, dtype=np.int64)
    eager = make(n, opponent="dummy", difficulty="hard", opp_seed=4)
    graphed = make(n, opponent="dummy", difficulty="hard", opp_seed=4)
    for b in (eager, graphed):
        b.seed(seeds); b.reset()
    a1 = torch.stack([eager.uniform_actions(9, t) for t in range(T)])
    out = graphed._alloc_out((), True, False, False, None)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for t in range(T):
            graphed.step(a1[t], None, auto_reset=True, out=out)
    for rep in range(2):                                   # two replays == 2T eager steps
        g.replay()
        for t in range(T):
            r = eager.step(a1[t], None, auto_reset=True)
        torch.cuda.synchronize()
        assert torch.equal(out.obs, r.obs) and torch.equal(out.reward, r.reward) and torch.equal(out.done, r.done)
    se, sg = eager.get_state(), graphed.get_state()
    for k in se:
        assert bits_eq(se[k], sg[k]), k


@pytest.mark.parametrize("W,H,K", [(15, 10, 16), (32, 32, 16), (6, 5, 24)])
def test_arbitrary_states_vs_oracle(torch_cuda, W, H, K):
    """One full step from ARBITRARY (not just reachable) states: random grids, players anywhere (also on top of
    each other / on their own colour), any ammo / reload / alive / prev_alive combination, up to K-6 bullets with
    random positions, directions and owners (many on the same tile), random step counters around the timeout."""
    from oracle import oracle as orc
    from oracle.oracle import pack_bullet
    n = 2048
    rng = np.random.RandomState(W * 100 + K)
    GW = (W * H + 31) // 32
    b = make(n, W, H, max_bullets=K, opponent="selfplay", max_episode_steps=50)
    b.seed(np.arange(n, dtype=np.int64)); b.reset()
    ob = orc.OracleBatch(n, W, H, max_episode_steps=50)
    ob.seed_reset(np.arange(n, dtype=np.int64))
    for rep in range(6):
        dens = rng.choice([0.05, 0.5, 0.95], size=(n, 1))
        bits = (rng.random_sample((n, GW * 32)) < dens).astype(np.uint32)
        bits[:, W * H:] = 0
        grid = (bits.reshape(n, GW, 32) << np.arange(32, dtype=np.uint32)).sum(axis=2).astype(np.uint32)
        players = np.zeros((n, 2, 5), np.int32)
        players[:, :, 0] = rng.randint(0, W, (n, 2)); players[:, :, 1] = rng.randint(0, H, (n, 2))
        same = rng.random_sample(n) < 0.1
        players[same, 1, :2] = players[same, 0, :2]
        players[:, :, 2] = rng.randint(0, 7, (n, 2)); players[:, :, 3] = rng.randint(0, 30, (n, 2))
        players[:, :, 4] = (rng.random_sample((n, 2)) < 0.85)
        nb = rng.randint(0, K - 5, n).astype(np.int32)
        cx, cy = rng.randint(0, W, n), rng.randint(0, H, n)                     # cluster centre: forces shared target tiles
        bx = np.clip(cx[:, None] + rng.randint(-2, 3, (n, K)), 0, W - 1)
        by = np.clip(cy[:, None] + rng.randint(-2, 3, (n, K)), 0, H - 1)
        bd, bo = rng.randint(0, 4, (n, K)), rng.randint(0, 2, (n, K))
        bullets = (bx | (by << 6) | (bd << 12) | (bo << 14)).astype(np.uint16)
        bullets[np.arange(K)[None, :] >= nb[:, None]] = 0
        steps = rng.randint(0, 52, n).astype(np.int32)
        epret = rng.randn(n)
        prev = (rng.random_sample((n, 2)) < 0.9).astype(np.uint8)
        b.set_state(grid=grid, players=players, nbullets=nb, bullets=bullets, step_count=steps, episode_return=epret, prev_alive=prev)
        for i in range(n):
            e = ob.env(i)
            e.set_grid_bits(grid[i])
            for pid in range(2):
                e.set_player(pid, *[int(v) for v in players[i, pid]])
            e.set_bullets([(int(bx[i, j]), int(by[i, j]), int(bd[i, j]), int(bo[i, j])) for j in range(nb[i])])
            e.s.step_count = int(steps[i]); e.s.episode_return = float(epret[i])
            e.s.prev_alive[0], e.s.prev_alive[1] = int(prev[i, 0]), int(prev[i, 1])
        a1 = rng.randint(0, 13, n).astype(np.uint8); a2 = rng.randint(0, 13, n).astype(np.uint8)
        r = b.step(a1, a2, auto_reset=(rep % 2 == 0), obs_p2=True, reward64=True)
        o_obs, o_extra, _, o_r64, o_done, o_info = ob.step(a1, a2, auto_reset=(rep % 2 == 0), nthreads=8)
        _compare_step(b, ob, r, o_obs, o_extra, o_r64, o_done, o_info, rep)
        g2, x2 = ob.observe(1)
        assert bits_eq(r.obs_p2.cpu().numpy(), g2) and bits_eq(r.extra_p2.cpu().numpy(), x2)
    b.check()


def test_arbitrary_states_golden(torch_cuda):
    """G7: one step from 1 500 arbitrary states set the way the reference's tests poke them, against the
    reference's own outputs (all cases as one device batch)."""
    g = load("g7_arbitrary_15x10.npz")
    n = len(g["a1"])
    b = make(n, 15, 10, opponent="selfplay", max_episode_steps=int(g["max_steps"]))
    b.seed(np.arange(n, dtype=np.int64)); b.reset()
    b.set_state(grid=g["grid"], players=g["players"], nbullets=g["nbullets"].astype(np.int32), bullets=g["bullets"],
                step_count=g["steps"].astype(np.int32), episode_return=g["epret"], prev_alive=g["prev"].astype(np.uint8))
    r = b.step(g["a1"].astype(np.uint8), g["a2"].astype(np.uint8), obs_p2=True, reward64=True)
    s = b.get_state()
    assert np.array_equal(s["grid"], g["o_grid"]) and np.array_equal(s["players"], g["o_players"])
    assert np.array_equal(s["nbullets"], g["o_nbullets"]) and np.array_equal(s["bullets"], g["o_bullets"])
    assert np.array_equal(obs_bits(r.obs), g["o_obs"]) and np.array_equal(obs_bits(r.obs_p2), g["o_obs_p2"])
    assert bits_eq(r.extra.cpu().numpy(), g["o_extra"])
    assert bits_eq(r.reward64.cpu().numpy(), g["o_reward"]) and np.array_equal(r.done.cpu().numpy(), g["o_done"].astype(np.uint8))
    want = g["o_info"].astype(np.uint8)
    assert np.array_equal(r.info_flags.cpu().numpy(), want[:, 0] | (want[:, 1] << 1) | (want[:, 2] << 2) | (want[:, 3] << 3))
    assert np.array_equal(r.episode_steps.cpu().numpy(), g["o_steps"]) and bits_eq(r.episode_return.cpu().numpy(), g["o_epret"].astype(np.float64))
    b.check()


def test_action_buffers_ending_exactly_at_an_allocation_end(torch_cuda):
    """The action bytes are fetched as aligned dwords through the scalar cache: nothing beyond the 4-byte-rounded end of a1 / a2
    may be read (a version of the four-envs-per-wave kernel always fetched two dwords and faulted when the buffer ended at the
    end of a mapped segment).  a1 / a2 are the last bytes of 2 MiB allocations (torch hands those out as whole segments)."""
    torch = torch_cuda
    from oracle import oracle as orc
    n, T = 16384, 3
    big1 = torch.zeros(2 * 1024 * 1024, dtype=torch.uint8, device="cuda")
    big2 = torch.zeros(2 * 1024 * 1024, dtype=torch.uint8, device="cuda")
    a1 = big1[-T * n:].view(T, n)
    a2 = big2[-T * n:].view(T, n)
    g = torch.Generator(device="cuda").manual_seed(3)
    a1.copy_(torch.randint(0, 13, (T, n), device="cuda", generator=g, dtype=torch.uint8))
    a2.copy_(torch.randint(0, 13, (T, n), device="cuda", generator=g, dtype=torch.uint8))
    b = make(n, opponent="selfplay")
    seeds = np.arange(n, dtype=np.int64)
    b.seed(seeds); b.reset()
    r = b.rollout(a1, a2, obs=False)                                     # multi-step kernel: t * N + env indexing up to the last byte
    last = b.step(a1[T - 1], a2[T - 1], obs=False, auto_reset=True)      # single-step kernel on the final row of the buffer
    ob = orc.OracleBatch(n)
    ob.seed_reset(seeds)
    for t in list(range(T)) + [T - 1]:                                   # the rollout's T steps, then the extra single step
        _, _, o_r32, _, o_done, _ = ob.step(a1[t].cpu().numpy(), a2[t].cpu().numpy(), want_obs=False, nthreads=8)
        if t < T - 1 or t == T - 1 and last is not None:
            want_r, want_d = o_r32, o_done
    assert bits_eq(last.reward.cpu().numpy(), want_r) and np.array_equal(last.done.cpu().numpy(), want_d)
    o = ob.state(16)
    s = b.get_state()
    for k in ("grid", "players", "nbullets", "bullets", "step_count"):
        assert np.array_equal(s[k], o[k]), k
    b.check()


# ------------------------------------------------------------------ observation store policy (round 3)
@pytest.mark.parametrize("W,H,n", [(15, 10, 64), (15, 10, 62), (32, 32, 16), (8, 8, 16), (9, 7, 16)])
def test_observation_store_policies_and_unaligned_buffers_give_identical_bytes(torch_cuda, W, H, n):
    """inv_set_obs_store_policy is a speed knob: write-through stores (chosen by the library where every store covers whole
    128-B lines: aligned pointers and 4-env groups on 15x10 / line-multiple regions) and plain stores produce the same
    observations, also when the caller's buffer is NOT 128-B aligned (the library must fall back to plain stores) and when
    num_envs is not a multiple of four (the last wave of the four-env kernel writes region by region)."""
    torch = torch_cuda
    from inversus_reinforcement_learning_amd.device_env import StepResult
    outs = {}
    for tag, store, shift in (("auto", "auto", 0), ("plain", "plain", 0), ("auto, buffer + 16 B", "auto", 4), ("auto, buffer + 64 B", "auto", 16)):
        b = make(n, W, H, opponent="selfplay", obs_store=store)
        b.seed(np.arange(n, dtype=np.int64) + 11)
        b.reset()
        big = torch.full((2, n * 12 * H * W + 64), -7.0, device="cuda")
        views = [big[q, shift:shift + n * 12 * H * W].view(n, 12, H, W) for q in range(2)]
        assert views[0].data_ptr() % 128 == (4 * shift) % 128
        r0 = b._alloc_out((), True, True, False, None)
        o = StepResult(**{k: getattr(r0, k) for k in StepResult.__slots__})
        o.obs, o.obs_p2 = views
        rng = np.random.RandomState(3)
        seen = []
        for t in range(24):
            a1, a2 = rng.randint(0, 13, n).astype(np.uint8), rng.randint(0, 13, n).astype(np.uint8)
            b.step(a1, a2, auto_reset=True, out=o)
            seen.append((o.obs.cpu().numpy().copy(), o.obs_p2.cpu().numpy().copy(), o.reward.cpu().numpy().copy()))
        b.check()
        # nothing outside the views was touched
        assert float(big[:, :shift].sum()) == -7.0 * 2 * shift and bool((big[:, shift + n * 12 * H * W:] == -7.0).all())
        outs[tag] = seen
    ref = outs["auto"]
    for tag, seen in outs.items():
        for t, (x, y) in enumerate(zip(ref, seen)):
            assert all(bits_eq(p, q) for p, q in zip(x, y)), (tag, t)
    # and the rendered observation (inv_render_obs) through both policies
    for store in ("auto", "plain"):
        b = make(n, W, H, opponent="selfplay", obs_store=store)
        b.seed(np.arange(n, dtype=np.int64) + 11)
        b.reset()
        g0, _ = b.render(0)
        outs[store + " render"] = g0.cpu().numpy()
    assert bits_eq(outs["auto render"], outs["plain render"])


def test_env_shards_on_their_own_streams_equal_one_batch(torch_cuda):
    """EnvShards: the same envs cut into independent shards, each stepped by a free-running chain of launches on its own HIP stream
    (no synchronisation between the shards until the end), produce what one batch produces -- observations, rewards, dones, state."""
    torch = torch_cuda
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch, EnvShards
    n, T = 1000, 40                                           # 1000 = 334 + 333 + 333: ragged shards, shard sizes not multiples of 4
    one = DeviceEnvBatch(n, opponent="dummy", difficulty="hard", opp_seed=9)
    one.seed(17 + np.arange(n, dtype=np.int64)); one.reset()
    shards = EnvShards(n, 3, seed=17, opponent="dummy", difficulty="hard", opp_seed=9)
    assert [sh.count for sh in shards] == [334, 333, 333] and [sh.offset for sh in shards] == [0, 334, 667]
    acts = [one.uniform_actions(3, t) for t in range(T)]
    torch.cuda.synchronize()
    want = []
    for t in range(T):
        r = one.step(acts[t], None, auto_reset=True, reward64=True)
        want.append((r.obs.clone(), r.reward64.clone(), r.done.clone()))
    got = [[] for _ in shards]
    for k, sh in enumerate(shards):                           # each shard runs its WHOLE chain before the next is even queued ...
        with torch.cuda.stream(sh.stream):
            for t in range(T):
                r = sh.batch.step(acts[t][sh.offset:sh.offset + sh.count].contiguous(), None, auto_reset=True, reward64=True)
                got[k].append((r.obs, r.reward64, r.done))
    shards.synchronize()                                      # ... and nothing waited for anything until here
    shards.check()
    for t in range(T):
        for i, name in enumerate(("obs", "reward64", "done")):
            cat = torch.cat([got[k][t][i] for k in range(len(shards))])
            assert bits_eq(cat.cpu().numpy(), want[t][i].cpu().numpy()), (t, name)
    s1 = one.get_state()
    for sh in shards:
        with torch.cuda.stream(sh.stream):
            s2 = sh.batch.get_state()
        for k2 in ("grid", "players", "bullets", "nbullets", "step_count"):
            assert np.array_equal(s2[k2], s1[k2][sh.offset:sh.offset + sh.count]), k2
