"""PPOAgent.update against the reference's update (inversus_rl/ppo_agent.py:159-247), CPU f32.

tests/golden/g8_ppo_update.npz was produced by the reference PPOAgent on a rollout collected by the reference's own
loop (tests/golden/gen_golden.py::gen_ppo_update): the stored buffer, the per-epoch permutations `np.random.shuffle`
produced, and what the reference computed (normalised advantages, per-minibatch losses, entropy, pre-clip gradient
norm, returned means, per-tensor parameter sums after the update).  The agent under test gets the same buffer, the same
initial weights (torch.manual_seed(0); G5 proves equal construction) and the same permutations.

The GAE of this CPU test comes from the oracle (test infrastructure); the HIP GAE is checked against the same fixture
in tests/test_gpu_ppo_update_golden.py.  Two modes.  "numpy_f32": the advantage mean / std are computed the way the reference does (NumPy f32 `.mean()` /
`.std()`, ppo_agent.py:173) and injected -- then every recorded quantity must agree to f32 round-off (losses 1e-6 abs,
gradient norm 1e-6 rel; on the generating host they are bit-equal).  "shipped": the package's own statistics (one f64
(sum, sum of squares, count) reduction, the form that all-reduces over ranks) differ from NumPy's by 1 ulp (1.2e-7) in
the normalised advantages; Adam's sign-like first steps amplify that over successive minibatches (measured: 0 / 0 /
1e-7 / 2e-6 / ... / 6e-5 relative on the gradient norm of minibatches 1..8), so the bounds are 1e-5 on the first three
minibatches and 5e-4 afterwards.  Parameter |sums| 1e-6 rel (numpy_f32) or 2e-4 rel (shipped: LayerNorm biases start at zero, so their |sum| IS the sum of
sign-like Adam steps), parameter deltas 2e-3 rel in both modes."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_g8():
    with np.load(os.path.join(GOLDEN, "g8_ppo_update.npz")) as z:
        return {k: z[k] for k in z.files}


def fill_agent(agent, g, device, torch):
    """The reference's buffer (one transition per store_step call, env-minor) as T vector steps of N envs."""
    N, T = int(g["N"]), int(g["T"])
    grid = np.unpackbits(g["obs"], axis=1)[:, :1800].astype(np.float32).reshape(T, N, 12, 10, 15)
    t = lambda a, dt: torch.as_tensor(a, dtype=dt, device=device)
    for k in range(T):
        s = slice(k * N, (k + 1) * N)
        agent.store_batch(t(grid[k], torch.float32), t(g["extra"][s], torch.float32), t(g["actions"][s], torch.int64),
                          t(g["log_probs"][s], torch.float32), t(g["values"][s], torch.float32),
                          t(g["rewards"][s], torch.float32), t(g["dones"][s].astype(bool), torch.bool))


def check_case(g, case, stats, policy, tol):
    mb = stats["minibatches"]
    np.testing.assert_allclose(stats["advantages"], g[f"{case}_adv_norm"], atol=tol["adv"], rtol=0)
    np.testing.assert_allclose(mb["policy_loss"], g[f"{case}_mb_policy_loss"], atol=tol["loss"], rtol=0)
    np.testing.assert_allclose(mb["value_loss"], g[f"{case}_mb_value_loss"], atol=tol["loss"] * 10, rtol=tol["loss"])
    np.testing.assert_allclose(mb["entropy"], g[f"{case}_mb_entropy"], atol=tol["loss"], rtol=0)
    np.testing.assert_allclose(mb["grad_norm"][:3], g[f"{case}_mb_grad_norm"][:3], rtol=tol["gn"], atol=0)
    np.testing.assert_allclose(mb["grad_norm"], g[f"{case}_mb_grad_norm"], rtol=tol.get("gn_late", tol["gn"]), atol=0)
    ref = g[f"{case}_stats"]
    np.testing.assert_allclose([stats["policy_loss"], stats["value_loss"], stats["entropy"]], ref, atol=tol["loss"] * 10, rtol=tol["loss"])
    sd = policy.state_dict()
    assert list(sd.keys()) == list(g["keys"])
    np.testing.assert_allclose([float(v.double().abs().sum()) for v in sd.values()], g[f"{case}_abs_sums"], rtol=tol["sums"], atol=0)


@pytest.mark.parametrize("stats_mode", ["numpy_f32", "shipped"])
@pytest.mark.parametrize("case", ["e1_full", "e2_b32"])
def test_update_matches_the_reference_update_cpu(case, stats_mode, monkeypatch):
    import torch
    from inversus_reinforcement_learning_amd import parallel
    if stats_mode == "numpy_f32":
        def numpy_mean_std(x):
            a = x.detach().cpu().numpy()
            return torch.tensor(a.mean()), torch.tensor(a.std())
        monkeypatch.setattr(parallel, "global_mean_std", numpy_mean_std)
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
    from oracle import oracle as orc
    g = load_g8()
    N, T = int(g["N"]), int(g["T"])
    torch.manual_seed(0)
    policy = InversusCNNPolicy(12, 10, 15, 4)
    sd0 = {k: v.clone() for k, v in policy.state_dict().items()}
    agent = PPOAgent(policy, epochs=int(g[f"{case}_epochs"]), batch_size=int(g[f"{case}_batch"]), device="cpu",
                     gae_layout="reference_flat")
    fill_agent(agent, g, "cpu", torch)
    # GAE of the flat env-interleaved buffer: the oracle here (bit-equal to the fixture), the HIP kernel in the GPU test
    adv, ret = orc.gae_flat(g["rewards"].reshape(T, N), g["values"].reshape(T, N), g["dones"].reshape(T, N))
    assert np.array_equal(adv.reshape(-1).view(np.uint32), g["adv_flat"].view(np.uint32))
    assert np.array_equal(ret.reshape(-1).view(np.uint32), g["ret_flat"].view(np.uint32))
    stats = agent.update(advantages=adv.reshape(-1), returns=ret.reshape(-1), permutations=g[f"{case}_perm"], trace=True)
    tol = dict(adv=1e-7, loss=1e-6, gn=1e-6, sums=1e-6) if stats_mode == "numpy_f32" else \
        dict(adv=2e-6, loss=1e-5, gn=1e-5, gn_late=5e-4, sums=2e-4)
    check_case(g, case, stats, policy, tol)
    # the parameter deltas themselves (Adam, lr 1e-4, clip 0.5): per tensor, sum |delta| and sum delta
    sd1 = policy.state_dict()
    d_abs = np.array([float((sd1[k].double() - sd0[k].double()).abs().sum()) for k in sd1])
    np.testing.assert_allclose(d_abs, g[f"{case}_delta_abs_sums"], rtol=2e-3, atol=1e-9)
    assert agent.obs_grid_buffer == []                                   # buffers cleared (ppo_agent.py:237)


def test_update_rejects_bad_permutations():
    import torch
    from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
    g = load_g8()
    agent = PPOAgent(torch.nn.Linear(1, 1), epochs=2, batch_size=32, device="cpu")
    fill_agent(agent, g, "cpu", torch)
    z = np.zeros(128, np.float32)
    with pytest.raises(ValueError):
        agent.update(advantages=z, returns=z, permutations=[np.arange(128)])            # one per epoch
    with pytest.raises(ValueError):
        agent.update(advantages=z, returns=z, permutations=[np.arange(64), np.arange(64)])
