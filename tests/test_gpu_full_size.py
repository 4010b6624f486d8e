"""Parity at the sizes BASELINE.json quotes (16 384 envs per GPU), through the C ABI against the CPU oracle:

  (a) the BENCHED workload -- vs_dummy EASY, 15 x 10, 16 384 envs, 64 vector steps: observation (f32 planes, every env),
      f32 extras, f64 reward bits, done, and at the end grid / players / ordered bullets of EVERY env;
  (b) config C3 at size -- self-play collector with packed observations, fused act + fused update, N = 16 384, T = 128,
      one update: buffer sizes, 256 sampled env columns replayed through the oracle, finite statistics;
  (c) config C5 at size -- 32 x 32 board, 16 384 envs, 32 steps vs the HARD device dummy: four 64-env blocks (1/64 of the
      batch) replayed by the oracle incl. observations, plus invariants over all envs;
  (d) the adversarial 32 x 32 trace of SURVEY 8(d): both players fire RIGHT from x = 0 every tick -- 14 live bullets,
      equal to the oracle at every step, no overflow flag.
Bar: bit-exact."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("both_step_kernels")]


def bits_eq(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def test_benched_workload_easy_16384_every_env_vs_oracle():
    import torch
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    from oracle import oracle as orc
    n, T = 16384, 64
    seeds = np.arange(n, dtype=np.int64)
    b = DeviceEnvBatch(n, opponent="dummy", difficulty="easy", opp_seed=1)          # bench.py's configuration
    b.seed(seeds); b.reset()
    ob = orc.OracleBatch(n)
    ob.seed_reset(seeds)
    bufs = ob.alloc_out()
    out = b._alloc_out((), True, False, True, None)
    ndone = 0
    for t in range(T):
        a1 = b.uniform_actions(1234, t)                                           # the bench's action stream
        r = b.step(a1, None, auto_reset=True, out=out)
        o_obs, o_extra, o_r32, o_r64, o_done, _ = ob.step(a1.cpu().numpy(), None, hard=False, opp_seed=1, nthreads=16, out=bufs)
        assert torch.equal(r.obs, torch.from_numpy(o_obs).cuda()), f"observation differs at step {t}"
        assert bits_eq(r.extra.cpu().numpy(), o_extra), t
        assert bits_eq(r.reward64.cpu().numpy(), o_r64), f"f64 reward bits differ at step {t}"
        assert bits_eq(r.reward.cpu().numpy(), o_r32), t
        assert np.array_equal(r.done.cpu().numpy(), o_done), t
        ndone += int(o_done.sum())
    s = b.get_state()
    o = ob.state(16)
    for k in ("grid", "players", "nbullets", "bullets", "step_count", "prev_alive"):                  # bullets: ordered lists
        assert np.array_equal(s[k], o[k]), k
    assert bits_eq(s["episode_return"], o["episode_return"])
    assert ndone > 0
    b.check()


def test_c3_selfplay_collector_and_fused_update_at_size():
    import torch
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
    from inversus_reinforcement_learning_amd.training import RolloutCollector
    from oracle import oracle as orc
    n, T = 16384, 128
    seeds = 70000 + np.arange(n, dtype=np.int64)
    b = DeviceEnvBatch(n, opponent="selfplay")
    b.seed(seeds); b.reset()
    torch.manual_seed(0)
    agent = PPOAgent(InversusCNNPolicy(12, 10, 15, 4), fused_act=True, fused_update=True, batch_size=n, epochs=1)
    opp = FusedPolicy(InversusCNNPolicy(12, 10, 15, 4).cuda().eval())
    col = RolloutCollector(b, agent, opp, record_opponent_actions=True)
    assert col.packed
    assert col.collect(T) == T * n
    # rollout buffer: packed observations, 57 words = 228 B per env-step and perspective (0.48 GB for the P1 side)
    assert len(agent.obs_grid_buffer) == T and all(g.dtype == torch.int32 and tuple(g.shape) == (n, 57) for g in agent.obs_grid_buffer)
    assert sum(g.numel() * 4 for g in agent.obs_grid_buffer) == T * n * 228 == 478_150_656
    # replay 256 sampled env columns (4 blocks of 64) through the oracle: packed P1 obs carried forward, rewards, dones
    for off in (0, 5000, 11111, n - 64):
        cols = slice(off, off + 64)
        ob = orc.OracleBatch(64)
        ob.seed_reset(seeds[cols])
        for t in range(T):
            a1 = agent.action_buffer[t][cols].cpu().numpy().astype(np.uint8)
            a2 = col.opponent_actions[t][cols].cpu().numpy().astype(np.uint8)
            o_obs, o_extra, o_r32, _, o_done, _ = ob.step(a1, a2)
            assert bits_eq(agent.reward_buffer[t][cols].cpu().numpy(), o_r32), (off, t)
            assert np.array_equal(agent.done_buffer[t][cols].cpu().numpy(), o_done.astype(bool)), (off, t)
            nxt = (agent.obs_grid_buffer[t + 1] if t + 1 < T else col.obs[0])[cols].cpu().numpy().view(np.uint32)
            want = np.packbits(o_obs.reshape(64, -1) != 0, axis=1, bitorder="little")
            want = np.pad(want, ((0, 0), (0, 57 * 4 - want.shape[1]))).view(np.uint32)
            assert np.array_equal(nxt, want), (off, t)
    assert col.episodes_finished == int(torch.stack(agent.done_buffer).sum())
    stats = agent.update()
    assert all(np.isfinite(stats[k]) for k in ("policy_loss", "value_loss", "entropy")) and 2.0 < stats["entropy"] <= np.log(13) + 1e-3
    assert all(torch.isfinite(p).all() for p in agent.policy.parameters())
    b.check()


def test_c5_selfplay_collector_and_fused_update_at_size():
    """BASELINE config C5 with the policy in the loop AND the fused update (round 3) at 16 384 envs: 4 self-play vector steps on the
    32 x 32 board with packed observations (384 words per env-step), two env columns blocks replayed by the oracle from the
    recorded actions, then one PPO epoch through the banded training kernels (4 minibatches of 16 384 samples)."""
    import torch
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
    from inversus_reinforcement_learning_amd.training import RolloutCollector
    from oracle import oracle as orc
    n, T, W, H = 16384, 4, 32, 32
    seeds = 90000 + np.arange(n, dtype=np.int64)
    b = DeviceEnvBatch(n, W, H, opponent="selfplay")
    b.seed(seeds); b.reset()
    torch.manual_seed(0)
    agent = PPOAgent(InversusCNNPolicy(12, H, W, 4), fused_act=True, fused_update=True, batch_size=n, epochs=1)
    assert agent.direct_grads and agent.fused.banded
    opp = FusedPolicy(InversusCNNPolicy(12, H, W, 4).cuda().eval(), actor_only=True)
    col = RolloutCollector(b, agent, opp, record_opponent_actions=True)
    assert col.packed and col.collect(T) == T * n
    assert all(g.dtype == torch.int32 and tuple(g.shape) == (n, 384) for g in agent.obs_grid_buffer)
    for off in (0, n - 64):
        cols = slice(off, off + 64)
        ob = orc.OracleBatch(64, W, H)
        ob.seed_reset(seeds[cols])
        for t in range(T):
            a1 = agent.action_buffer[t][cols].cpu().numpy().astype(np.uint8)
            a2 = col.opponent_actions[t][cols].cpu().numpy().astype(np.uint8)
            o_obs, _, o_r32, _, o_done, _ = ob.step(a1, a2)
            assert bits_eq(agent.reward_buffer[t][cols].cpu().numpy(), o_r32), (off, t)
            assert np.array_equal(agent.done_buffer[t][cols].cpu().numpy(), o_done.astype(bool)), (off, t)
            nxt = (agent.obs_grid_buffer[t + 1] if t + 1 < T else col.obs[0])[cols].cpu().numpy().view(np.uint32)
            want = np.packbits(o_obs.reshape(64, -1) != 0, axis=1, bitorder="little").view(np.uint32)
            assert np.array_equal(nxt, want), (off, t)
    before = agent.flat_adam.flat.clone()
    stats = agent.update()
    # (a random-init 32 x 32 policy is less uniform than the 15 x 10 one: its logits sum over 131 072 features)
    assert all(np.isfinite(stats[k]) for k in ("policy_loss", "value_loss", "entropy")) and 0.5 < stats["entropy"] <= np.log(13) + 1e-3
    assert bool(torch.isfinite(agent.flat_adam.flat).all()) and float((agent.flat_adam.flat - before).abs().max()) > 0
    b.check()


def test_c5_32x32_hard_16384_blocks_vs_oracle():
    import torch
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    from oracle import oracle as orc
    n, T, W, H = 16384, 32, 32, 32
    seeds = 300000 + np.arange(n, dtype=np.int64)
    b = DeviceEnvBatch(n, W, H, opponent="dummy", difficulty="hard", opp_seed=9)
    b.seed(seeds); b.reset()
    blocks = (0, 4096 + 17, 9999, n - 64)
    obs_ = {off: orc.OracleBatch(64, W, H) for off in blocks}
    for off, ob in obs_.items():
        ob.seed_reset(seeds[off:off + 64])
    out = b._alloc_out((), True, False, True, None)
    for t in range(T):
        a1 = b.uniform_actions(77, t)
        r = b.step(a1, None, auto_reset=True, out=out)
        a1h = a1.cpu().numpy()
        o = r.obs
        assert torch.all(o[:, 0] + o[:, 1] == 1) and torch.all((o == 0) | (o == 1))
        for off, ob in obs_.items():
            cols = slice(off, off + 64)
            o_obs, o_extra, _, o_r64, o_done, _ = ob.step(a1h[cols], None, hard=True, opp_seed=9, env_offset=off)
            assert bits_eq(o[cols].cpu().numpy(), o_obs), (off, t)
            assert bits_eq(r.extra[cols].cpu().numpy(), o_extra) and bits_eq(r.reward64[cols].cpu().numpy(), o_r64), (off, t)
            assert np.array_equal(r.done[cols].cpu().numpy(), o_done), (off, t)
    s = b.get_state()
    for off, ob in obs_.items():
        o = ob.state(16)
        for k in ("grid", "players", "nbullets", "bullets", "step_count"):
            assert np.array_equal(s[k][off:off + 64], o[k]), (off, k)
    b.check()


def test_adversarial_32x32_trace_reaches_14_live_bullets():
    """both players at x = 0 firing RIGHT every tick on the 32 x 32 board -- the worst case for the ordered
    bullet list (bound 2 * (6 + floor(30 / 30)) = 14 live bullets, K = 16)."""
    import torch
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    from oracle import oracle as orc
    n, W, H, T = 256, 32, 32, 96
    seeds = np.arange(n, dtype=np.int64)
    b = DeviceEnvBatch(n, W, H, opponent="selfplay", max_episode_steps=10 ** 6)
    b.seed(seeds); b.reset()
    st = b.get_state(("players",))["players"]
    st[:, :, 0] = 0; st[:, 0, 1] = 3; st[:, 1, 1] = 20
    b.set_state(players=st)
    ob = orc.OracleBatch(n, W, H, max_episode_steps=10 ** 6)
    ob.seed_reset(seeds)
    for i in range(n):
        e = ob.env(i)
        e.set_player(0, x=0, y=3); e.set_player(1, x=0, y=20)
    fire = np.full(n, 6, np.uint8)                                       # SHOOT RIGHT
    fire_d = torch.from_numpy(fire).cuda()
    mx = 0
    for t in range(T):
        r = b.step(fire_d, fire_d, reward64=True)
        o_obs, o_extra, _, o_r64, o_done, _ = ob.step(fire, fire, auto_reset=False)
        s = b.get_state(("nbullets", "bullets", "grid", "players"))
        o = ob.state(16)
        for k in ("nbullets", "bullets", "grid", "players"):
            assert np.array_equal(s[k], o[k]), (t, k)
        assert bits_eq(r.obs.cpu().numpy(), o_obs) and bits_eq(r.reward64.cpu().numpy(), o_r64), t
        mx = max(mx, int(o["nbullets"].max()))
    assert mx == 14, mx
    b.check()                                                            # no overflow flag


def test_c4_total_size_131072_envs_equals_eight_shards_of_16384():
    """BASELINE config C4's env path at its TOTAL size on one GPU: one batch of 131 072 self-play envs (what the 8-GPU run
    holds in total) equals eight shards of 16 384 with env_offset = 16384 * rank (what each rank holds) -- packed P1 / P2
    observations, rewards, dones and final state of every env -- and four 64-env blocks equal the oracle.  The 8-GPU job
    itself needs the driver's node; its only collective is the gradient all-reduce (tests/test_policy_dist_cpu.py)."""
    import torch
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    from oracle import oracle as orc
    n_total, n_shard, T = 131072, 16384, 12
    base = 5_000_000
    g = torch.Generator(device="cuda").manual_seed(8)
    a1 = torch.randint(0, 13, (T, n_total), device="cuda", generator=g, dtype=torch.uint8)
    a2 = torch.randint(0, 13, (T, n_total), device="cuda", generator=g, dtype=torch.uint8)
    big = DeviceEnvBatch(n_total, opponent="selfplay", max_episode_steps=7)
    big.seed(base + np.arange(n_total, dtype=np.int64)); big.reset()
    rb = big.rollout(a1, a2, obs_p2=True, bits=True)
    for rank in (0, 3, 7):
        lo = rank * n_shard
        sh = DeviceEnvBatch(n_shard, opponent="selfplay", max_episode_steps=7, env_offset=lo)
        sh.seed(base + lo + np.arange(n_shard, dtype=np.int64)); sh.reset()
        rs = sh.rollout(a1[:, lo:lo + n_shard].contiguous(), a2[:, lo:lo + n_shard].contiguous(), obs_p2=True, bits=True)
        for k in ("obs", "obs_p2", "extra", "reward", "done", "info_flags", "episode_steps", "episode_return"):
            assert torch.equal(getattr(rs, k), getattr(rb, k)[:, lo:lo + n_shard]), (rank, k)
        ss, sb = sh.get_state(), big.get_state()
        for k in ss:
            assert bits_eq(ss[k], sb[k][lo:lo + n_shard]), (rank, k)
    a1h, a2h = a1.cpu().numpy(), a2.cpu().numpy()
    for off in (0, 40000, 99999, n_total - 64):
        ob = orc.OracleBatch(64, max_episode_steps=7)
        ob.seed_reset(base + off + np.arange(64, dtype=np.int64))
        for t in range(T):
            o_obs, _, o_r32, _, o_done, _ = ob.step(a1h[t, off:off + 64], a2h[t, off:off + 64])
            assert bits_eq(rb.reward[t, off:off + 64].cpu().numpy(), o_r32) and np.array_equal(rb.done[t, off:off + 64].cpu().numpy(), o_done), (off, t)
            want = np.packbits(o_obs.reshape(64, -1) != 0, axis=1, bitorder="little")
            want = np.pad(want, ((0, 0), (0, 57 * 4 - want.shape[1]))).view(np.uint32)
            assert np.array_equal(rb.obs[t, off:off + 64].cpu().numpy().view(np.uint32), want), (off, t)
    big.check()
