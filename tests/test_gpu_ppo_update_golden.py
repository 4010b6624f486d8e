"""PPOAgent.update on the GPU against the reference's update (tests/golden/g8_ppo_update.npz, see
tests/test_ppo_update_golden_cpu.py for the fixture): the f32 path with the HIP GAE kernel (reference_flat layout,
bit-equal advantages), then the fused bf16 path bounded against the reference numbers as well.

Tolerances: the f32 GPU path runs MIOpen / hipBLASLt kernels whose summation order differs from the CPU reference:
first minibatch losses 2e-4 abs (value loss 2e-4 rel), gradient norm 1e-3 rel; later minibatches inherit Adam's
amplification of that noise (CPU test docstring): 2e-2 rel on the gradient norm, 5e-3 abs on the losses.
The fused path computes the trunk in bf16 (8 significant bits) -- a class bound, like tests/test_gpu_fused_train.py, not
parity.  First minibatch (same weights as the reference, so only the bf16 forward / backward differs; measured on MI355X:
policy loss 1e-4, value loss 9e-4 abs, entropy 4e-6, gradient norm 0.3 %): policy loss and entropy 2e-3 abs, value loss
1e-3 rel, gradient norm 3 %.  Later minibatches start from weights that already differ by bf16-sized gradient noise times
Adam's sign-like steps, on 32-sample minibatches (measured up to 0.031 on the policy loss): policy loss 0.1 abs, entropy
2e-2 abs, value loss 5 %, final parameter |sums| 2 %."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_ppo_update_golden_cpu import fill_agent, load_g8          # noqa: E402

pytestmark = pytest.mark.gpu


def _run(case, g, **agent_kw):
    import torch
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
    torch.manual_seed(0)
    policy = InversusCNNPolicy(12, 10, 15, 4)
    agent = PPOAgent(policy, epochs=int(g[f"{case}_epochs"]), batch_size=int(g[f"{case}_batch"]), device="cuda",
                     gae_layout="reference_flat", **agent_kw)
    fill_agent(agent, g, "cuda", torch)
    adv, ret = agent.compute_advantages()                                # HIP scan over the env-interleaved buffer
    assert np.array_equal(adv.view(np.uint32), g["adv_flat"].view(np.uint32))
    assert np.array_equal(ret.view(np.uint32), g["ret_flat"].view(np.uint32))
    stats = agent.update(permutations=g[f"{case}_perm"], trace=True)
    return stats, policy


@pytest.mark.parametrize("case", ["e1_full", "e2_b32"])
def test_f32_gpu_update_matches_the_reference_update(case):
    g = load_g8()
    stats, policy = _run(case, g)
    mb = stats["minibatches"]
    np.testing.assert_allclose(stats["advantages"], g[f"{case}_adv_norm"], atol=2e-6, rtol=0)
    dev = {k: np.abs(mb[k] - g[f"{case}_mb_{k}"]) for k in ("policy_loss", "value_loss", "entropy", "grad_norm")}
    print(case, {k: float(v.max()) for k, v in dev.items()}, "first", {k: float(v[0]) for k, v in dev.items()})
    assert dev["policy_loss"][0] < 2e-4 and dev["entropy"][0] < 2e-4
    assert dev["value_loss"][0] < 2e-4 * g[f"{case}_mb_value_loss"][0] + 2e-4
    assert dev["grad_norm"][0] < 1e-3 * g[f"{case}_mb_grad_norm"][0]
    assert dev["policy_loss"].max() < 5e-3 and dev["entropy"].max() < 5e-3
    np.testing.assert_allclose(mb["value_loss"], g[f"{case}_mb_value_loss"], rtol=5e-3)
    np.testing.assert_allclose(mb["grad_norm"], g[f"{case}_mb_grad_norm"], rtol=2e-2)
    np.testing.assert_allclose([stats["policy_loss"], stats["value_loss"], stats["entropy"]], g[f"{case}_stats"], rtol=5e-3, atol=5e-3)
    sd = policy.state_dict()
    np.testing.assert_allclose([float(v.double().abs().sum()) for v in sd.values()], g[f"{case}_abs_sums"], rtol=1e-3)


@pytest.mark.parametrize("case", ["e1_full", "e2_b32"])
def test_fused_update_stays_within_bf16_of_the_reference_update(case):
    g = load_g8()
    stats, policy = _run(case, g, fused_update=True)
    mb = stats["minibatches"]
    dev = {k: np.abs(mb[k] - g[f"{case}_mb_{k}"]) for k in ("policy_loss", "value_loss", "entropy", "grad_norm")}
    print(case, "fused", {k: float(v.max()) for k, v in dev.items()}, "first", {k: float(v[0]) for k, v in dev.items()})
    assert dev["policy_loss"][0] < 2e-3 and dev["entropy"][0] < 2e-3
    assert dev["value_loss"][0] < 1e-3 * g[f"{case}_mb_value_loss"][0] + 1e-3
    assert dev["grad_norm"][0] < 0.03 * g[f"{case}_mb_grad_norm"][0]
    np.testing.assert_allclose(mb["policy_loss"], g[f"{case}_mb_policy_loss"], atol=0.1)
    np.testing.assert_allclose(mb["entropy"], g[f"{case}_mb_entropy"], atol=2e-2)
    np.testing.assert_allclose(mb["value_loss"], g[f"{case}_mb_value_loss"], rtol=5e-2)
    sd = policy.state_dict()
    np.testing.assert_allclose([float(v.double().abs().sum()) for v in sd.values()], g[f"{case}_abs_sums"], rtol=2e-2)
