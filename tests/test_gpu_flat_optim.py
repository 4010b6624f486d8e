"""csrc/inv_optim.hip against the torch expressions it replaces (inversus_rl/ppo_agent.py:211-232 runs them under autograd /
torch.optim): the PPO minibatch loss with its backward, and clip_grad_norm_ + Adam over one flat buffer.  f32 on both
sides; tolerance 1e-6 relative (2e-6 on the losses: block-wise summation order differs from torch's)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def torch_loss(torch, logits, values, actions, old_logp, adv, ret, clip, vc, ec):
    dist = torch.distributions.Categorical(logits=logits)
    new_logp = dist.log_prob(actions)
    entropy = dist.entropy().mean()
    ratio = torch.exp(new_logp - old_logp)
    s1, s2 = ratio * adv, torch.clamp(ratio, 1.0 - clip, 1.0 + clip) * adv
    pol = -torch.min(s1, s2).mean()
    val = torch.nn.functional.mse_loss(values, ret)
    return pol + vc * val - ec * entropy, pol, val, entropy


@pytest.mark.parametrize("B", [1, 37, 512, 16384])
def test_ppo_loss_kernel_matches_autograd(B):
    import torch
    from inversus_reinforcement_learning_amd.flat_optim import ppo_loss
    g = torch.Generator(device="cuda").manual_seed(B)
    logits = (3.0 * torch.randn(B, 13, device="cuda", generator=g)).requires_grad_()
    values = torch.randn(B, device="cuda", generator=g).requires_grad_()
    actions = torch.randint(0, 13, (B,), device="cuda", generator=g)
    with torch.no_grad():
        lp = torch.log_softmax(logits, 1).gather(1, actions[:, None])[:, 0]
    # old log-probs around the new ones so that every branch occurs: inside the clip range, clipped high / low with either sign of A
    old_logp = lp + 0.25 * torch.randn(B, device="cuda", generator=g)
    old_logp[::7] = lp[::7]                                             # ratio == 1 exactly (the state right after a rollout)
    adv = torch.randn(B, device="cuda", generator=g)
    adv[::11] = 0.0
    ret = torch.randn(B, device="cuda", generator=g)
    want, pol, val, ent = torch_loss(torch, logits, values, actions, old_logp, adv, ret, 0.2, 0.1, 0.02)
    want.backward()
    dl_want, dv_want = logits.grad.clone(), values.grad.clone()
    logits.grad = values.grad = None
    loss, means = ppo_loss(logits, values, actions, old_logp, adv, ret, 0.2, 0.1, 0.02)
    (2.5 * loss).backward()                                              # an upstream factor must scale the stored gradients
    np.testing.assert_allclose(means.cpu().numpy(), [pol.item(), val.item(), ent.item()], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(loss.item(), want.item(), rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(logits.grad.cpu().numpy(), 2.5 * dl_want.cpu().numpy(), rtol=2e-5, atol=1e-5 / B)       # atol: 1e-5 of the 1/B gradient scale (entries where the policy and entropy terms cancel)
    np.testing.assert_allclose(values.grad.cpu().numpy(), 2.5 * dv_want.cpu().numpy(), rtol=1e-6, atol=1e-9)


def test_clip_adam_matches_torch_clip_grad_norm_and_adam():
    import torch
    from inversus_reinforcement_learning_amd.flat_optim import FlatAdam
    torch.manual_seed(0)
    shapes = [(32, 12, 3, 3), (32,), (257, 19), (3,), (128, 128, 3, 3), (1,)]         # odd sizes: the flat length is not a multiple of 4
    ref = [torch.nn.Parameter(torch.randn(s, device="cuda") * 0.1) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    opt = torch.optim.Adam(ref, lr=1e-3)
    fa = FlatAdam(mine, lr=1e-3, max_grad_norm=0.5)
    assert fa.n == sum(p.numel() for p in ref) and fa.n % 4 != 0
    for step in range(6):
        scale = [5.0, 1e-3, 0.3, 2.0, 1e-2, 1.0][step]                  # norms above and below the clip threshold
        grads = [scale * torch.randn_like(p) for p in ref]
        opt.zero_grad()
        for p, q, gr in zip(ref, mine, grads):
            p.grad = gr.clone()
            q.grad.copy_(gr)                                             # the flat bucket is the .grad storage
        norm_want = torch.nn.utils.clip_grad_norm_(ref, 0.5)
        opt.step()
        norm = fa.step()
        np.testing.assert_allclose(float(norm), float(norm_want), rtol=1e-6)
        for p, q in zip(ref, mine):
            np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=1e-6, atol=1e-7)
    # grad_scale = 1 / world_size applied inside the kernel == averaging the summed gradient first
    q2 = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    fb = FlatAdam(q2, lr=1e-3, max_grad_norm=0.5)
    fb.exp_avg.copy_(fa.exp_avg); fb.exp_avg_sq.copy_(fa.exp_avg_sq); fb.steps = fa.steps
    gsum = torch.randn(fa.n, device="cuda")
    fa.grad.copy_(gsum / 4); fb.grad.copy_(gsum)
    na, nb = fa.step(), fb.step(grad_scale=0.25)
    np.testing.assert_allclose(float(na), float(nb), rtol=1e-6)
    np.testing.assert_allclose(fb.flat.cpu().numpy(), fa.flat.cpu().numpy(), rtol=1e-6, atol=1e-7)


def test_flat_adam_detects_detached_buffers():
    import torch
    from inversus_reinforcement_learning_amd.flat_optim import FlatAdam
    lin = torch.nn.Linear(8, 4).cuda()
    fa = FlatAdam(lin.parameters())
    lin.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError):
        fa.step()


def test_agent_with_fused_update_owns_flat_parameters_and_actor_sees_updates():
    import torch
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
    from inversus_reinforcement_learning_amd.training import RolloutCollector
    torch.manual_seed(0)
    b = DeviceEnvBatch(256, opponent="dummy", difficulty="hard", opp_seed=3)
    b.seed(np.arange(256, dtype=np.int64)); b.reset()
    agent = PPOAgent(InversusCNNPolicy(12, 10, 15, 4), fused_act=True, fused_update=True, batch_size=1024, epochs=2)
    assert agent.flat_adam is not None and agent.flat_adam.n == 10249582
    sd0 = {k: v.clone() for k, v in agent.policy.state_dict().items()}
    col = RolloutCollector(b, agent)
    col.collect(16)
    stats = agent.update(trace=True)
    assert all(np.isfinite(stats[k]) for k in ("policy_loss", "value_loss", "entropy"))
    assert np.all(np.isfinite(stats["minibatches"]["grad_norm"])) and np.all(stats["minibatches"]["grad_norm"] > 0)
    assert agent.flat_adam.steps == 2 * 4
    changed = [k for k, v in agent.policy.state_dict().items() if not torch.equal(v, sd0[k])]
    assert len(changed) == len(sd0)                                      # every tensor moved (views of the flat buffer)
    # checkpoints still round-trip through the reference's format (plain state_dict)
    pol2 = InversusCNNPolicy(12, 10, 15, 4)
    pol2.load_state_dict({k: v.cpu() for k, v in agent.policy.state_dict().items()})
    agent.policy.load_state_dict(pol2.state_dict())                      # in-place copy: the aliasing must survive
    agent.flat_adam._check_aliases()


def test_sample_actions_is_categorical_sampling_with_the_loss_kernels_log_probs():
    """inv_sample_actions: the sampled distribution is softmax(logits) (chi-square over 2 M draws of one row), log_prob is the
    log-softmax entry of the sampled action (to f32 round-off of torch's, and bit-equal to what inv_ppo_loss computes: ratio 1),
    torch.manual_seed controls the stream, both action forms agree."""
    import torch
    from inversus_reinforcement_learning_amd.flat_optim import ppo_loss, sample_actions
    torch.manual_seed(3)
    row = torch.tensor([0.3, -1.2, 2.0, 0.0, -3.0, 1.1, 0.7, -0.4, 0.9, -2.2, 0.1, 1.6, -0.8], device="cuda")
    n = 1 << 21
    logits = row.repeat(n, 1)
    a, a8, lp = sample_actions(logits, want_u8=True)
    assert a.dtype == torch.int64 and a8.dtype == torch.uint8 and torch.equal(a, a8.long()) and int(a.min()) >= 0 and int(a.max()) <= 12
    p = torch.softmax(row.double(), 0)
    counts = torch.bincount(a, minlength=13).double()
    chi2 = float((((counts - n * p) ** 2) / (n * p)).sum())
    assert chi2 < 45.0, chi2                                               # 12 degrees of freedom: P(chi2 > 45) ~ 1e-5
    want_lp = torch.log_softmax(logits[:4096], -1).gather(1, a[:4096, None]).squeeze(1)
    assert float((lp[:4096] - want_lp).abs().max()) < 1e-6
    # varied logits: the same log-probs as the loss kernel -> policy term of the loss = -mean(A) exactly (ratio == 1)
    g = torch.Generator(device="cuda").manual_seed(0)
    lg = torch.randn(5000, 13, device="cuda", generator=g) * 2
    torch.manual_seed(11); a1, _, lp1 = sample_actions(lg)
    torch.manual_seed(11); a2, _, lp2 = sample_actions(lg)
    torch.manual_seed(12); a3, _, _ = sample_actions(lg)
    assert torch.equal(a1, a2) and torch.equal(lp1, lp2) and not torch.equal(a1, a3)
    adv = torch.randn(5000, device="cuda", generator=g)
    _, means = ppo_loss(lg, torch.zeros(5000, device="cuda"), a1, lp1, adv, torch.zeros(5000, device="cuda"), 0.2, 0.1, 0.02)
    assert abs(float(means[0]) + float(adv.double().mean())) < 1e-6
    # every row gets its own uniform: a batch of identical rows does not produce identical actions
    assert len(torch.unique(a[:64])) > 1


def test_every_rank_gets_a_sampling_stream_of_its_own():
    """inv_sample_actions keys Philox by torch's device generator and counts by the LOCAL sample index, and every
    process starts that generator from the same default seed -- so env i of every shard drew the same uniform at every step.
    training._make now seeds the generator per rank (parallel.seed_device_generator): same (seed, rank) = same stream, another
    rank = another stream."""
    import torch
    from inversus_reinforcement_learning_amd.flat_optim import sample_actions
    from inversus_reinforcement_learning_amd.parallel import seed_device_generator
    logits = torch.zeros(8192, 13, device="cuda")

    def stream(base, rank):
        seed_device_generator(base, rank)
        return torch.stack([sample_actions(logits)[0] for _ in range(3)])

    a0, a1, a0_again, b0 = stream(7, 0), stream(7, 1), stream(7, 0), stream(8, 0)
    assert torch.equal(a0, a0_again)
    assert float((a0 != a1).float().mean()) > 0.85 and float((a0 != b0).float().mean()) > 0.85      # 12/13 expected
    assert seed_device_generator(7, 0) != seed_device_generator(7, 1)


def test_agent_optimizer_is_a_handle_on_the_optimizer_that_steps():
    """with fused_update the agent used to keep a torch Adam that was never stepped -- changing its learning rate
    had no effect and checkpointing it lost Adam's state.  agent.optimizer is now a handle on FlatAdam."""
    import torch
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
    torch.manual_seed(0)
    agent = PPOAgent(InversusCNNPolicy(12, 10, 15, 4), lr=3e-4, fused_update=True, batch_size=64, epochs=1)
    opt = agent.optimizer
    assert opt.param_groups[0]["lr"] == pytest.approx(3e-4) and agent.flat_adam.lr == pytest.approx(3e-4)
    for g in opt.param_groups:                                            # the usual scheduler idiom
        g["lr"] = 1e-5
    assert agent.flat_adam.lr == pytest.approx(1e-5)
    # the group is a real dict -- "lr" shows up wherever common optimizer code looks for it, and the dict is the source
    # of truth for the learning rate FlatAdam steps with
    g = opt.param_groups[0]
    assert "lr" in g and "lr" in g.keys() and dict(g)["lr"] == pytest.approx(1e-5) and g.copy()["lr"] == pytest.approx(1e-5)
    assert {k: v for k, v in g.items() if k != "params"}["lr"] == pytest.approx(1e-5)
    assert g.setdefault("initial_lr", g["lr"]) == pytest.approx(1e-5) and g.setdefault("lr", 9.0) == pytest.approx(1e-5)
    g.update(lr=2e-5)
    assert agent.flat_adam.lr == pytest.approx(2e-5)
    with pytest.raises(RuntimeError):
        opt.step()
    agent.flat_adam.grad.normal_()
    agent.flat_adam.step()
    sd = opt.state_dict()
    assert sd["steps"] == 1 and float(sd["exp_avg"].abs().sum()) > 0
    before = agent.flat_adam.flat.clone()
    agent.flat_adam.grad.normal_()
    agent.flat_adam.step()
    after_two = agent.flat_adam.flat.clone()
    # restore the state after step 1 and the parameters, replay step 2 with the same gradient: same result
    g2 = agent.flat_adam.grad.clone()
    opt.load_state_dict(sd)
    agent.flat_adam.flat.copy_(before)
    agent.flat_adam.grad.copy_(g2)
    agent.flat_adam.step()
    assert agent.flat_adam.steps == 2 and torch.equal(agent.flat_adam.flat, after_two)
    # a state saved for other parameter shapes / another order is refused (the flat moments would land on the wrong weights)
    bad = dict(sd, param_shapes=list(reversed(sd["param_shapes"])))
    with pytest.raises(ValueError):
        opt.load_state_dict(bad)
