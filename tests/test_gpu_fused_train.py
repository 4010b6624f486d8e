"""GPU tests of the fused trunk training path (csrc/inv_policy.hip: k_policy_trunk<SAVE>, k_policy_trunk_bwd;
fused_train.py) against torch autograd on the f32 module (the reference's update runs autograd over
inversus_rl/policies.py:91-106).  bf16 operands / f32 accumulation: tolerances are relative L2 errors per tensor."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _policy(seed=0, perturb=True):
    import torch
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    torch.manual_seed(seed)
    pol = InversusCNNPolicy(12, 10, 15, 4).cuda()
    if perturb:
        with torch.no_grad():                                  # non-trivial LayerNorm affine and biases
            for nm in (pol.norm1, pol.norm2, pol.norm3, pol.norm4):
                nm.weight.add_(0.2 * torch.randn_like(nm.weight)); nm.bias.add_(0.2 * torch.randn_like(nm.bias))
    return pol


def _obs(n, seed=0):
    """Real observations: a few random self-play steps from seeded resets (packed and f32 of the same states)."""
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    b = DeviceEnvBatch(n, opponent="selfplay")
    b.seed(seed + np.arange(n, dtype=np.int64)); b.reset()
    rng = np.random.RandomState(seed)
    for _ in range(5):
        b.step(rng.randint(0, 13, n).astype(np.uint8), rng.randint(0, 13, n).astype(np.uint8), auto_reset=True, obs=False)
    gf, x = b.render(0)
    gb, _ = b.render(0, bits=True)
    return gf, gb, x


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


@pytest.mark.parametrize("n", [3, 700])
def test_training_forward_matches_inference_trunk_and_saves_consistent_tensors(n):
    import torch
    from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
    from inversus_reinforcement_learning_amd.fused_train import FusedTrainPolicy
    pol = _policy()
    gf, gb, x = _obs(n)
    ft, fi = FusedTrainPolicy(pol), FusedPolicy(pol)
    with torch.no_grad():
        f_train = ft.features(gf)
        assert torch.equal(f_train, ft.features(gb))           # packed observations: same operands
        f_inf = fi.features(gf)
        want = pol.trunk(gf).reshape(n, 128, 150).permute(0, 2, 1).reshape(n, 19200)
    # the training forward rounds x^ to bf16 before the affine (so that the backward sees the same ReLU mask);
    # otherwise the same pipeline as the inference kernel
    assert (f_train.float() - f_inf.float()).abs().max() < 0.2 and (f_train.float() - f_inf.float()).abs().mean() < 0.004
    assert (f_train.float() - want).abs().max() < 0.25 and (f_train.float() - want).abs().mean() < 0.008
    lt, vt = ft(gf, x)
    with torch.no_grad():
        lm, vm = pol(gf, x)
    assert (lt - lm).abs().max() < 0.05 and (vt - vm).abs().max() < 0.05
    # saved == NULL (the learner's rollout forward): the inference epilogue with the training forward's numerics, bit for bit
    fr = FusedPolicy(pol, train_numerics=True)
    with torch.no_grad():
        assert torch.equal(fr.features(gf), f_train) and torch.equal(fr.features(gb), f_train)


def test_rollout_forward_with_train_numerics_is_bitwise_the_training_forward():
    """inv_policy_trunk_train_fwd(saved = NULL) is another instantiation of the kernel (inference epilogue + x^ rounded to bf16).
    Both must fuse their multiply-adds the same way: with contraction left to the compiler the LayerNorm statistics differed in
    the last bit and ~4e-7 of the features by one bf16 ulp (seen at n = 4096, not at n = 3)."""
    import torch
    from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
    from inversus_reinforcement_learning_amd.fused_train import FusedTrainPolicy
    n = 4096
    pol = _policy(3)
    gf, gb, _ = _obs(n, seed=7)
    ft, fr = FusedTrainPolicy(pol), FusedPolicy(pol, train_numerics=True)
    with torch.no_grad():
        want = ft.features(gb)
        assert torch.equal(fr.features(gb), want) and torch.equal(fr.features(gf), want) and torch.equal(ft.features(gf), want)


@pytest.mark.parametrize("n,packed", [(64, False), (333, True), (1300, True)])     # 1300 > 512 resident blocks: persistent loops
def test_parameter_gradients_match_f32_autograd(n, packed):
    import torch
    from inversus_reinforcement_learning_amd.fused_train import FusedTrainPolicy
    pol = _policy(1)
    gf, gb, x = _obs(n, 7)
    torch.manual_seed(2)
    r1, r2 = torch.randn(n, 13, device="cuda"), torch.randn(n, 1, device="cuda")
    def grads(fwd):
        pol.zero_grad()
        lo, v = fwd()
        ((lo.float() * r1).sum() + (v.float() * r2).sum()).backward()
        return {k: p.grad.clone() for k, p in pol.named_parameters()}

    def autocast_fwd():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return pol(gf, x)

    want = grads(lambda: pol(gf, x))                            # f32 autograd: the reference's update path
    scale = {k: rel(g, want[k]) for k, g in grads(autocast_fwd).items()}   # what bf16 operands cost torch itself
    got = grads(lambda: FusedTrainPolicy(pol)(gb if packed else gf, x))
    errs = {k: rel(g, want[k]) for k, g in got.items()}
    assert set(errs) == set(want) and all(np.isfinite(list(errs.values())))
    # bf16 features flip a few ReLU masks in the heads, which dominates both errors (7-10 % on this random upstream
    # gradient); the fused path must stay in the class of torch's own bf16 autocast, tensor by tensor
    for k in errs:
        assert errs[k] <= max(2.0 * scale[k], 0.03), (k, errs[k], scale[k])
        assert errs[k] < 0.2, (k, errs[k])


@pytest.mark.parametrize("n", [1, 37, 1100])
def test_wgrad_kernel_matches_im2col_gemm(n):
    """inv_policy_wgrad (MFMA over pixels, transposing LDS reads, f32 atomics) vs the same contraction done by torch
    in f32 on the same bf16 operands: only the summation order differs."""
    import torch
    from inversus_reinforcement_learning_amd import fused_train as FT
    torch.manual_seed(n)
    for layer in range(4):
        cin, cinp, cout = FT._CIN[layer], FT._CINP[layer], FT._COUT[layer]
        x = torch.zeros(n, 150, cinp, device="cuda")
        x[:, :, :cin] = torch.randn(n, 150, cin, device="cuda")
        x = x.to(torch.bfloat16)
        dz = (torch.randn(n, 150, cout, device="cuda") * 0.1).to(torch.bfloat16)
        got, got_b = FT._wgrad(layer, x, dz)
        xf = torch.nn.functional.pad(x[..., :cin].float().reshape(n, 10, 15, cin), (0, 0, 1, 1, 1, 1))
        cols = xf.unfold(1, 3, 1).unfold(2, 3, 1).permute(0, 1, 2, 4, 5, 3).reshape(n * 150, 9 * cin)
        want = (dz.float().reshape(n * 150, cout).t() @ cols).reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
        assert tuple(got.shape) == (cout, cin, 3, 3)
        assert rel(got, want) < 2e-5, (layer, rel(got, want))
        assert float((got - want).abs().max()) < 1e-3 * max(1.0, float(want.abs().max())), layer
        want_b = dz.float().sum(dim=(0, 1))
        assert float((got_b - want_b).abs().max()) < 1e-3 * max(1.0, float(want_b.abs().max())), layer


def test_param_grad_reductions_match_torch():
    import ctypes as C
    import torch
    from inversus_reinforcement_learning_amd import _lib as L
    lib = L.load()
    torch.manual_seed(0)
    for n, c in ((1, 32), (77, 64), (900, 128)):
        dy, xh = [(torch.randn(n, 150, c, device="cuda") * s).to(torch.bfloat16) for s in (0.1, 1.0)]
        slices = min(n, 16)
        part = torch.full((slices, 2, 150, c), float("nan"), device="cuda")
        rc = lib.inv_policy_param_grads(*[C.c_void_p(t.data_ptr()) for t in (dy, xh, part)], slices, 150 * c, n,
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        red = part.sum(0)
        want = torch.stack([(dy.float() * xh.float()).sum(0), dy.float().sum(0)])
        assert rel(red, want) < 1e-5 and float((red - want).abs().max()) < 1e-3


def test_ppo_update_with_fused_trunk_tracks_torch_update():
    """One PPO epoch through PPOAgent(fused_update=True) moves the parameters like the torch update does."""
    import torch
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
    from inversus_reinforcement_learning_amd.training import RolloutCollector
    n, T = 128, 8
    deltas = []
    for fused in (True, False):
        torch.manual_seed(3)
        b = DeviceEnvBatch(n, opponent="dummy", difficulty="hard", opp_seed=1, max_episode_steps=30)
        b.seed(np.arange(n, dtype=np.int64)); b.reset()
        agent = PPOAgent(InversusCNNPolicy(12, 10, 15, 4), lr=1e-4, batch_size=n * T, epochs=1, fused_act=True, fused_update=fused)
        before = torch.cat([p.detach().reshape(-1).clone() for p in agent.policy.parameters()])
        RolloutCollector(b, agent).collect(T)
        stats = agent.update()
        assert all(np.isfinite(list(stats.values())))
        deltas.append(torch.cat([p.detach().reshape(-1) for p in agent.policy.parameters()]) - before)
    # Adam's first step is lr * sign-like: compare directions
    cos = float(torch.nn.functional.cosine_similarity(deltas[0], deltas[1], dim=0))
    assert cos > 0.9, cos


def test_rollout_forward_uses_the_updates_numerics_when_the_update_is_fused():
    """PPOAgent(fused_act=True, fused_update=True): log-probs / values stored by act() are what the update's first
    forward recomputes on the same observations (PPO ratio starts at 1); with the inference epilogue they differ ~1e-2."""
    import torch
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
    from inversus_reinforcement_learning_amd.training import RolloutCollector
    n, T = 256, 6
    torch.manual_seed(0)
    b = DeviceEnvBatch(n, opponent="dummy", difficulty="hard", opp_seed=2)
    b.seed(np.arange(n, dtype=np.int64)); b.reset()
    pol = _policy(5)
    agent = PPOAgent(pol, fused_act=True, fused_update=True, batch_size=n * T, epochs=1)
    assert agent.fused.train_numerics
    RolloutCollector(b, agent).collect(T)
    grid, extra = torch.cat(agent.obs_grid_buffer), torch.cat(agent.obs_extra_buffer)
    acts, old_lp, old_v = torch.cat(agent.action_buffer), torch.cat(agent.log_prob_buffer), torch.cat(agent.value_buffer)
    with torch.no_grad():
        lo, v = agent.train_fwd(grid, extra)
        new_lp = torch.distributions.Categorical(logits=lo.float(), validate_args=False).log_prob(acts)
        # identical trunk features and identical head kernels; only the library GEMM of the first Linear runs at another batch
        # size (another tiling / summation order), which moves a bf16 rounding of `pre` here and there: mostly exact zeros
        assert float((new_lp - old_lp).abs().max()) < 2e-3 and float((v.float().squeeze(-1) - old_v).abs().max()) < 2e-3
        assert float((new_lp - old_lp).abs().mean()) < 1e-5
        # at the rollout's own batch size (what training uses: minibatch = num_envs) the stored numbers are reproduced exactly
        for t in range(T):
            lo_t, v_t = agent.train_fwd(agent.obs_grid_buffer[t], agent.obs_extra_buffer[t])
            lp_t = torch.distributions.Categorical(logits=lo_t.float(), validate_args=False).log_prob(agent.action_buffer[t])
            # stored log-probs come from inv_sample_actions (the loss kernel's log-softmax expressions), lp_t from torch's
            assert float((lp_t - agent.log_prob_buffer[t]).abs().max()) < 1e-6 and torch.equal(v_t.float().squeeze(-1), agent.value_buffer[t])
        lo_i, _ = FusedPolicy(pol)(grid, extra)                  # inference epilogue: close, but not the same numbers
        lp_i = torch.distributions.Categorical(logits=lo_i.float(), validate_args=False).log_prob(acts)
        assert 5e-4 < float((lp_i - old_lp).abs().max()) < 0.1


def test_pack_kernels_equal_the_torch_index_arithmetic():
    """inv_policy_pack_conv / pack_ln / obs_to_x0 against the torch formulations they replace (bit-equal: pure data movement + one rounding)."""
    import torch
    from inversus_reinforcement_learning_amd.device_env import unpack_obs
    from inversus_reinforcement_learning_amd.fused_policy import _pack_conv, pack_conv, pack_ln
    from inversus_reinforcement_learning_amd.fused_train import _pack_ln
    from inversus_reinforcement_learning_amd import _lib as L
    import ctypes as C
    torch.manual_seed(3)
    for cout, cin, pad in ((32, 12, 32), (64, 32, 32), (128, 64, 64), (128, 128, 128)):
        w = torch.randn(cout, cin, 3, 3, device="cuda")
        assert torch.equal(pack_conv(w, pad), _pack_conv(w, pad))
        if cin % 16 == 0:
            wt = w.flip(2, 3).permute(1, 0, 2, 3).contiguous()
            assert torch.equal(pack_conv(w, cout, transposed=True), _pack_conv(wt, cout))
        g, b = torch.randn(cout, 10, 15, device="cuda"), torch.randn(cout, 10, 15, device="cuda")
        assert torch.equal(pack_ln(g, b, cout), _pack_ln(g, b, cout))
    n = 77
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, 57), dtype=torch.int32, device="cuda")
    bits[:, 56] &= 0xFF
    planes = unpack_obs(bits)
    want = torch.zeros((n, 150, 16), dtype=torch.bfloat16, device="cuda")
    want[:, :, :12] = planes.reshape(n, 12, 150).permute(0, 2, 1)
    lib = L.load()
    for src, is_bits in ((bits, 1), (planes, 0)):
        x0 = torch.empty_like(want)
        L.raise_for(lib.inv_policy_obs_to_x0(C.c_void_p(src.data_ptr()), is_bits, C.c_void_p(x0.data_ptr()), n,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), lib)
        assert torch.equal(x0, want)


def test_direct_gradients_equal_the_autograd_path():
    """FusedTrainPolicy.loss_and_grads (no autograd engine, gradients copied into the flat bucket) against loss.backward() over
    the same kernels: the same numbers up to the f32 atomics of the weight-gradient kernels."""
    import torch
    from inversus_reinforcement_learning_amd.flat_optim import ppo_loss
    from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
    n = 700
    pol = _policy(7)
    agent = PPOAgent(pol, fused_act=True, fused_update=True, batch_size=n, epochs=1)
    assert agent.direct_grads
    gf, gb, x = _obs(n, seed=3)
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = torch.randint(0, 13, (n,), device="cuda", generator=g)
    old_lp = -2.5 + 0.1 * torch.randn(n, device="cuda", generator=g)
    adv, ret = torch.randn(n, device="cuda", generator=g), torch.randn(n, device="cuda", generator=g)
    flat = agent.flat_adam.grad
    flat.fill_(123.0)                                                     # the direct path overwrites every slot
    means_d = agent.train_fwd.loss_and_grads(gb, x, acts, old_lp, adv, ret, 0.2, 0.1, 0.02)
    direct = flat.clone()
    assert bool(torch.isfinite(direct).all()) and not bool((direct == 123.0).any())
    flat.zero_()
    lo, v = agent.train_fwd(gb, x)
    loss, means_a = ppo_loss(lo.float(), v.float().squeeze(-1), acts, old_lp, adv, ret, 0.2, 0.1, 0.02)
    loss.backward()
    agent.flat_adam._check_aliases()
    assert torch.equal(means_d, means_a)
    off = 0
    for q in agent.flat_adam.params:
        a, b = direct[off:off + q.numel()], flat[off:off + q.numel()]
        assert rel(a, b) < 1e-5, (tuple(q.shape), rel(a, b))
        off += q.numel()


def test_backward_in_two_launches_gives_the_one_launch_gradients(monkeypatch):
    """INVERSUS_BWD_PARTS=2 (fused_train.py: the backward kernel over two sample ranges, each range's LayerNorm reduction queued on
    the side stream right behind it) against the default single launch: the same 16 trunk gradients up to the f32 atomics of the
    reductions and of the weight-gradient kernels.  8 192 samples: the smallest batch the split applies to."""
    import torch
    from inversus_reinforcement_learning_amd.fused_train import FusedTrainPolicy, _trunk_params
    n = 8192
    pol = _policy(11)
    gf, gb, x = _obs(n, seed=5)
    ft = FusedTrainPolicy(pol)
    g = torch.Generator(device="cuda").manual_seed(2)
    feat = ft.features(gb)
    dfeat = (torch.randn(feat.shape, device="cuda", generator=g) * 1e-3).to(feat.dtype)
    monkeypatch.setenv("INVERSUS_BWD_PARTS", "1")
    one = torch.autograd.grad(ft.features(gb), _trunk_params(pol), dfeat)
    monkeypatch.setenv("INVERSUS_BWD_PARTS", "2")
    two = torch.autograd.grad(ft.features(gb), _trunk_params(pol), dfeat)
    torch.cuda.synchronize()
    for a, b, q in zip(two, one, _trunk_params(pol)):
        assert rel(a, b) < 1e-5, (tuple(q.shape), rel(a, b))


def test_graph_captured_minibatches_give_the_eager_minibatches_gradients():
    """PPOAgent(fused_update=True, graph_update=True): a full-size minibatch's forward + loss + backward is captured once in a HIP graph and
    replayed (index_select into static inputs, one graph launch, then clip + Adam outside the graph).  With lr = 0 the parameters never
    move, so every minibatch of an update is evaluated at the same weights in both agents: after 2 epochs x 4 minibatches (1 eager,
    1 captured + replayed, 6 replayed) from the same rollout and the same permutations, the flat gradient bucket (= the LAST minibatch's
    gradients, written by a REPLAY in one agent and eagerly in the other) agrees to the f32 atomics' reassociation, the loss statistics
    averaged over all eight minibatches agree, and both optimizers have stepped eight times.  (Comparing parameters after real Adam steps
    is not a test of the graph: the first Adam steps are sign-like, so a last-bit difference of a near-zero gradient element -- which two
    EAGER runs have as well -- becomes a difference of the step size.)"""
    import torch
    from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
    n, T, bs = 128, 8, 256
    torch.manual_seed(1)
    g = torch.Generator(device="cuda").manual_seed(2)
    grid = (torch.rand(T, n, 12, 10, 15, device="cuda", generator=g) < 0.2)
    perms = [torch.randperm(n * T, generator=torch.Generator().manual_seed(7 + e)).numpy() for e in range(2)]
    outs = []
    for graph in (False, True):
        pol = _policy(4)
        agent = PPOAgent(pol, fused_act=True, fused_update=True, batch_size=bs, epochs=2, lr=0.0, graph_update=graph)
        assert agent.graph_update == graph and agent.direct_grads
        gg = torch.Generator(device="cuda").manual_seed(3)
        for t in range(T):
            planes = grid[t].float()
            extra = torch.rand(n, 4, device="cuda", generator=gg)
            acts = torch.randint(0, 13, (n,), device="cuda", generator=gg)
            lp = -2.5 + 0.1 * torch.randn(n, device="cuda", generator=gg)
            val = 0.1 * torch.randn(n, device="cuda", generator=gg)
            rew = 0.05 * torch.randn(n, device="cuda", generator=gg)
            done = torch.rand(n, device="cuda", generator=gg) < 0.05
            agent.store_batch(planes, extra, acts, lp, val, rew, done)
        init = [p.detach().clone() for p in agent.policy.parameters()]
        stats = agent.update(permutations=perms)
        assert (agent._mb_graph is not None) == graph and agent.flat_adam.steps == 8
        assert all(torch.equal(a, b.detach()) for a, b in zip(init, agent.policy.parameters()))           # lr = 0
        outs.append((stats, agent.flat_adam.grad.detach().clone(), [p.grad.detach().clone() for p in agent.policy.parameters()]))
    (s0, f0, g0), (s1, f1, g1) = outs
    for k in ("policy_loss", "value_loss", "entropy"):
        assert abs(s0[k] - s1[k]) < 1e-5 * max(1.0, abs(s0[k])), (k, s0[k], s1[k])
    assert float(f0.norm()) > 0
    for a, b in zip(g0, g1):
        assert float((a.double() - b.double()).norm()) <= 1e-4 * float(a.double().norm()) + 1e-9, (tuple(a.shape), float((a - b).norm() / a.norm()))


@pytest.mark.parametrize("n", [1, 333, 4096])
def test_top_layer_sums_without_a_dy_tensor_are_bitwise_the_sums_from_the_stored_dy(n):
    """ABI 8: inv_policy_trunk_bwd with grads.dy[3] == NULL writes no dy for the top layer, inv_policy_param_grads_all2 recomputes it
    as [x^ g + b > 0] * dfeat.  With ONE sample slice every output element receives exactly one atomic add onto zero, so the four
    layers' sums of both routes can be compared bit for bit; dz does not depend on whether dy[3] is stored."""
    import ctypes as C
    import torch
    from inversus_reinforcement_learning_amd import _lib as L
    from inversus_reinforcement_learning_amd import fused_train as FT
    lib = L.load()
    pol = _policy(3).train()
    ft = FT.FusedTrainPolicy(pol)
    gf, _, _ = _obs(n, 5)
    feat = ft.features(gf)
    sv = feat.grad_fn.saved_tensors
    obs, rstd = sv[0], sv[1]
    xhat, act, lns, ws = sv[2:6], sv[6:9], sv[9:13], sv[13:17]
    df = (torch.randn_like(feat.float()) * 0.01).to(torch.bfloat16)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bp, keep = L.InvPolicyBwdParams(), []
    for i in range(1, 4):
        pk = FT.pack_conv(ws[i], FT._COUT[i], transposed=True); keep.append(pk); bp.wt[i - 1] = pk.data_ptr()
    for i in range(4):
        bp.ln[i] = lns[i].data_ptr()
    saved = L.InvPolicySaved()
    for i in range(4):
        saved.xhat[i] = xhat[i].data_ptr()
    for i in range(3):
        saved.act[i] = act[i].data_ptr()
    saved.rstd = rstd.data_ptr()
    res = []
    for stored in (True, False):
        dy = [torch.full_like(x, 7.0) if (i < 3 or stored) else None for i, x in enumerate(xhat)]
        dz = [torch.full_like(x, 7.0) for x in xhat]
        grads = L.InvPolicyGrads()
        for i in range(4):
            grads.dy[i] = dy[i].data_ptr() if dy[i] is not None else None
            grads.dz[i] = dz[i].data_ptr()
        L.raise_for(lib.inv_policy_trunk_bwd(C.byref(bp), C.byref(saved), C.c_void_p(df.data_ptr()), C.byref(grads), n, st), lib)
        lng = L.InvPolicyLnGrads()
        gw = [torch.zeros(c, 10, 15, device="cuda") for c in FT._COUT]
        gb = [torch.zeros(c, 10, 15, device="cuda") for c in FT._COUT]
        for i in range(4):
            lng.gw[i], lng.gb[i] = gw[i].data_ptr(), gb[i].data_ptr()
        if stored:
            L.raise_for(lib.inv_policy_param_grads_all(C.byref(grads), C.byref(saved), C.byref(lng), 1, n, st), lib)
        else:
            L.raise_for(lib.inv_policy_param_grads_all2(C.byref(grads), C.byref(saved), C.byref(lng), C.c_void_p(df.data_ptr()),
                                                         C.c_void_p(lns[3].data_ptr()), 1, n, st), lib)
        torch.cuda.synchronize()
        res.append((dz, gw, gb, dy))
    (dz_a, gw_a, gb_a, dy_a), (dz_b, gw_b, gb_b, _) = res
    for i in range(4):
        assert torch.equal(dz_a[i], dz_b[i]), i
        assert torch.equal(gw_a[i], gw_b[i]) and torch.equal(gb_a[i], gb_b[i]), i
    assert float(gw_a[3].abs().sum()) > 0 and float(gb_a[3].abs().sum()) > 0
    # the stored dy of the top layer IS the masked dfeat (what the recomputation assumes)
    keepm = dy_a[3] != 0
    assert torch.equal(dy_a[3][keepm], df.reshape_as(dy_a[3])[keepm]) and 0.2 < float(keepm.float().mean()) < 0.9
    # argument checks: dfeat and the packed parameters come together; without them dy[3] is required
    grads = L.InvPolicyGrads()
    for i in range(3):
        grads.dy[i] = dy_a[i].data_ptr()
    assert lib.inv_policy_param_grads_all2(C.byref(grads), C.byref(saved), C.byref(lng), C.c_void_p(df.data_ptr()), None, 1, n, st) == L.INV_ERR_BAD_ARG
    assert lib.inv_policy_param_grads_all(C.byref(grads), C.byref(saved), C.byref(lng), 1, n, st) == L.INV_ERR_BAD_ARG
