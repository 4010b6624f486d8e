"""GPU tests of the fused policy heads (csrc/inv_heads.hip, heads.py) against the f32 torch module
(reference: inversus_rl/policies.py:60-75, :102-106, run under autograd in ppo_agent.py:206-230).
bf16 operands / f32 accumulation: tolerances are relative L2 errors per tensor, written at each assert."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _policy(seed=0):
    import torch
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    torch.manual_seed(seed)
    pol = InversusCNNPolicy(12, 10, 15, 4).cuda()
    with torch.no_grad():                                     # biases are zero-initialised in places: make every term count
        for seq in (pol.fc_actor, pol.fc_critic):
            for i in (0, 2, 4):
                seq[i].bias.add_(0.1 * torch.randn_like(seq[i].bias))
    return pol


def _inputs(n, seed=0):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    feat = (torch.randn(n, 19200, device="cuda", generator=g).clamp_min(0) * 0.5).to(torch.bfloat16)     # post-ReLU features
    extra = torch.rand(n, 4, device="cuda", generator=g)
    return feat, extra


def _module_heads(pol, feat, extra):
    """The module's heads on the module's feature order (c * 150 + p) in f32."""
    import torch
    n = feat.shape[0]
    x = torch.cat([feat.float().reshape(n, 150, 128).permute(0, 2, 1).reshape(n, 19200), extra], dim=1)
    return pol.fc_actor(x), pol.fc_critic(x)


def _emulated_heads(pol, feat, extra):
    """The kernels' numerics in torch f32 on rounded values (bf16 weights; h, a2 rounded to bf16, straight-through; `pre` stays f32:
    inv_fc1_fwd hands f32 partial sums to inv_heads_fwd), so that the ReLU masks are the kernels' masks: what the gradients are
    compared with."""
    import torch
    q = lambda x: x + (x.to(torch.bfloat16).float() - x).detach()
    outs = []
    for seq in (pol.fc_actor, pol.fc_critic):
        w1 = seq[0].weight
        wf = q(w1[:, :19200].reshape(256, 128, 150).permute(0, 2, 1).reshape(256, 19200))
        pre = feat @ wf.t()
        h = q(torch.relu(pre + extra @ w1[:, 19200:].t() + seq[0].bias))
        a2 = q(torch.relu(h @ q(seq[2].weight).t() + seq[2].bias))
        outs.append(a2 @ q(seq[4].weight).t() + seq[4].bias)
    return outs


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


@pytest.mark.parametrize("n", [1, 31, 33, 700])
def test_forward_matches_the_module_heads(n):
    import torch
    from inversus_reinforcement_learning_amd import heads as HD
    pol = _policy()
    feat, extra = _inputs(n)
    ops = HD.HeadsOperands(HD.head_tensors(pol), 150)
    with torch.no_grad():
        logits, value = ops.forward(feat, extra)
        lm, vm = _module_heads(pol, feat, extra)
    assert logits.shape == (n, 13) and value.shape == (n, 1)
    # bf16 weights and bf16 h / a2 against f32: a few 1e-3 relative
    assert rel(logits, lm) < 1e-2 and rel(value, vm) < 1e-2
    assert float((logits - lm).abs().max()) < 0.02 and float((value - vm).abs().max()) < 0.02


def test_actor_only_equals_the_actor_of_both_heads_and_results_do_not_depend_on_the_batch():
    import torch
    from inversus_reinforcement_learning_amd import heads as HD
    pol = _policy(1)
    feat, extra = _inputs(515, 1)
    both = HD.HeadsOperands(HD.head_tensors(pol), 150)
    actor = HD.HeadsOperands(HD.head_tensors(pol, actor_only=True), 150)
    with torch.no_grad():
        l2, v2 = both.forward(feat, extra)
        l1, v1 = actor.forward(feat, extra)
        assert v1 is None and float((l1 - l2).abs().max()) < 2e-3               # the narrower first GEMM splits K differently: another f32 summation order
        # a sample's result does not depend on where it sits in a tile: 32-row tiles of another batch, fed the same `pre` rows
        # (inv_fc1_fwd's slabs added in slab order, as inv_heads_fwd adds them) as ONE slab
        import ctypes as C
        from inversus_reinforcement_learning_amd import _lib as L
        lib = L.load()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        slabs = int(lib.inv_fc1_fwd_slabs(515, 512, 19200))
        part = torch.empty((slabs, 515, 512), dtype=torch.float32, device="cuda")
        L.raise_for(lib.inv_fc1_fwd(C.c_void_p(feat.data_ptr()), C.c_void_p(both.wf.data_ptr()), 515, 512, 19200, C.c_void_p(part.data_ptr()), st), lib)
        pre = part[0].clone()
        for k in range(1, slabs):
            pre += part[k]
        sub = torch.arange(7, 515, 3, device="cuda")
        pre_s, ex_s = pre[sub].contiguous(), extra[sub].contiguous()
        lo = torch.empty((sub.numel(), 13), dtype=torch.float32, device="cuda"); va = torch.empty((sub.numel(), 1), dtype=torch.float32, device="cuda")
        rc = lib.inv_heads_fwd(C.byref(both.params), C.c_void_p(pre_s.data_ptr()), 1, C.c_void_p(ex_s.data_ptr()), C.c_void_p(lo.data_ptr()),
                               C.c_void_p(va.data_ptr()), None, None, 2, sub.numel(), st)
        L.raise_for(rc, lib)
        assert torch.equal(lo, l2[sub]) and torch.equal(va, v2[sub])


def test_act_and_update_batch_sizes_split_k_differently_and_the_logits_differ_by_f32_reassociation_only():
    """Advisor r4: the forward's split-K count comes from inv_fc1_fwd_slabs(B, nout, K) (B and the CU count), so the stored
    old_log_prob of a 16 384-row rollout forward (2 slabs here) and the update's recomputation on a smaller minibatch (up to 8 slabs)
    add the same f32 products in another order.  The documented tolerance (heads.py): logits / values equal to f32 reassociation --
    max |d| < 2e-3 on logits of O(1) ... O(10), mean |d| < 2e-5 -- so the PPO ratio of epoch 0 is 1 +- 2e-3 when the minibatch is
    smaller than the rollout batch, and exactly 1 when they are equal (test_gpu_fused_train)."""
    import torch
    from inversus_reinforcement_learning_amd import heads as HD, _lib as L
    lib = L.load()
    pol = _policy(3)
    B, mb = 16384, 512
    feat, extra = _inputs(B, 3)
    s_roll, s_mb = int(lib.inv_fc1_fwd_slabs(B, 512, 19200)), int(lib.inv_fc1_fwd_slabs(mb, 512, 19200))
    assert s_roll >= 1 and s_mb >= 1 and s_roll != s_mb                     # the case the finding is about (otherwise the test proves nothing)
    ops = HD.HeadsOperands(HD.head_tensors(pol), 150)
    with torch.no_grad():
        lo, va = ops.forward(feat, extra)
        dl, dv = [], []
        for k in range(0, B, 4096):                                         # a sample of minibatches out of the rollout batch
            lo_m, va_m = ops.forward(feat[k:k + mb].contiguous(), extra[k:k + mb].contiguous())
            dl.append((lo_m - lo[k:k + mb]).abs()); dv.append((va_m - va[k:k + mb]).abs())
        dl, dv = torch.cat(dl), torch.cat(dv)
        assert float(dl.max()) < 2e-3 and float(dv.max()) < 2e-3 and float(dl.mean()) < 2e-5 and float(dv.mean()) < 2e-5
        lo2, va2 = ops.forward(feat, extra)                                 # equal sizes: the same bits (no atomics anywhere on this path)
        assert torch.equal(lo2, lo) and torch.equal(va2, va)


def test_an_odd_pixel_count_is_refused_before_the_forward_runs():
    """Advisor r4: inv_fc1_dfeat / inv_fc1_dw need 128 P to be a multiple of 256 (an even pixel count); heads_train has no torch tail,
    so an odd-P board must fail BEFORE the forward, not with INV_ERR_BAD_ARG in the middle of backward."""
    import torch
    from inversus_reinforcement_learning_amd import heads as HD
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    pol = InversusCNNPolicy(12, 3, 5, 4).cuda()                             # 15 pixels: odd
    assert not HD.supported(pol)
    feat = torch.zeros((4, 128 * 15), dtype=torch.bfloat16, device="cuda", requires_grad=False)
    with pytest.raises(ValueError, match="even"):
        HD.heads_train(pol, feat, torch.zeros((4, 4), device="cuda"))
    assert HD.supported(InversusCNNPolicy(12, 4, 5, 4).cuda()) and HD.supported(_policy())


@pytest.mark.parametrize("n", [33, 1024])
def test_gradients_match_f32_autograd(n):
    import torch
    from inversus_reinforcement_learning_amd import heads as HD
    pol = _policy(2)
    feat, extra = _inputs(n, 2)
    g = torch.Generator(device="cuda").manual_seed(5)
    dl = torch.randn(n, 13, device="cuda", generator=g) / n
    dv = torch.randn(n, 1, device="cuda", generator=g) / n
    params = [t for h in HD.head_tensors(pol) for t in h]
    # reference: f32 autograd through the module's heads evaluated with the kernels' roundings (same ReLU masks) ...
    f32 = feat.float().requires_grad_(True)
    lm, vm = _emulated_heads(pol, f32, extra)
    want = torch.autograd.grad([lm, vm], [f32] + params, [dl, dv])
    # ... and through the plain f32 module (masks flip where a pre-activation is within a bf16 rounding of zero)
    f32b = feat.float().requires_grad_(True)
    lp, vp = _module_heads(pol, f32b, extra)
    plain = torch.autograd.grad([lp, vp], [f32b] + params, [dl, dv])
    fb = feat.clone().requires_grad_(True)
    lo, va = HD.heads_train(pol, fb, extra)
    got = torch.autograd.grad([lo, va], [fb] + params, [dl, dv])
    names = ["dfeat", "a.W1", "a.b1", "a.W2", "a.b2", "a.W3", "a.b3", "c.W1", "c.b1", "c.W2", "c.b2", "c.W3", "c.b3"]
    errs = {nm: rel(gq, w) for nm, gq, w in zip(names, got, want)}
    print(errs)
    for nm, gq, w in zip(names, got, want):
        assert gq.shape == w.shape, nm
    # bf16 d out / d a2 / d pre in the kernels' chain (the emulation keeps gradients in f32): < 1.5e-2 relative per tensor
    for nm, e in errs.items():
        assert e < 1.5e-2, (nm, e)
    for nm, gq, w in zip(names, got, plain):
        assert rel(gq, w) < 0.12, (nm, rel(gq, w))


def test_ragged_batch_leaves_rows_past_b_untouched_and_adds_into_the_gradient_buffers():
    import ctypes as C
    import torch
    from inversus_reinforcement_learning_amd import _lib as L
    from inversus_reinforcement_learning_amd import heads as HD
    pol = _policy(3)
    n = 45                                                        # 32 + 13: the second tile is ragged
    feat, extra = _inputs(n, 3)
    ops = HD.HeadsOperands(HD.head_tensors(pol), 150, need_t=True)
    logits, value, ex, h, a2 = ops.forward(feat, extra, save=True)
    lib = L.load()
    dl = torch.randn(n, 13, device="cuda"); dv = torch.randn(n, 1, device="cuda")
    small = torch.full((4624,), 1.0, dtype=torch.float32, device="cuda")          # pre-filled: the kernel ADDS
    gr = L.InvHeadsGrads()
    b = small.data_ptr()
    gr.db1, gr.dwe = b, b + 512 * 4
    gr.db2[0], gr.db2[1] = b + 2560 * 4, b + 2688 * 4
    gr.dw3[0], gr.dw3[1] = b + 2816 * 4, b + 4480 * 4
    gr.db3[0], gr.db3[1] = b + 4608 * 4, b + 4621 * 4
    dpre = torch.full((n + 8, 512), 7.0, dtype=torch.bfloat16, device="cuda"); da2 = torch.full((n + 8, 256), 7.0, dtype=torch.bfloat16, device="cuda")
    rc = lib.inv_heads_bwd(C.byref(ops.params), C.c_void_p(dl.data_ptr()), C.c_void_p(dv.data_ptr()), C.c_void_p(h.data_ptr()), C.c_void_p(a2.data_ptr()),
                           C.c_void_p(ex.data_ptr()), C.c_void_p(dpre.data_ptr()), None, 0, C.c_void_p(da2.data_ptr()), C.byref(gr), n, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    L.raise_for(rc, lib)
    # d pre^T with a leading dimension (a multiple of 8, >= B): the same numbers, transposed; a ragged last tile writes only its own
    # columns, the padding columns are the caller's (inv_fc1_dw wants them zero)
    n8 = 40
    dpre8 = torch.full((n8, 512), 7.0, dtype=torch.bfloat16, device="cuda"); dpt = torch.full((512, 48), 7.0, dtype=torch.bfloat16, device="cuda")
    args = lambda nn, t, ld: (C.byref(ops.params), C.c_void_p(dl.data_ptr()), C.c_void_p(dv.data_ptr()), C.c_void_p(h.data_ptr()), C.c_void_p(a2.data_ptr()),
                              C.c_void_p(ex.data_ptr()), C.c_void_p(dpre8.data_ptr()), C.c_void_p(t.data_ptr()), ld, C.c_void_p(da2.data_ptr()), C.byref(gr), nn,
                              C.c_void_p(torch.cuda.current_stream().cuda_stream))
    saved_small = small.clone()
    L.raise_for(lib.inv_heads_bwd(*args(n8, dpt, 48)), lib)
    assert torch.equal(dpt[:, :n8], dpre8.t()) and bool((dpt[:, n8:] == 7).all()) and torch.equal(dpre8, dpre[:n8])
    assert lib.inv_heads_bwd(*args(n, dpt, 40)) == L.INV_ERR_BAD_ARG             # leading dimension < B
    assert lib.inv_heads_bwd(*args(n8, dpt, 44)) == L.INV_ERR_BAD_ARG            # not a multiple of 8 (rows must stay 16-byte aligned)
    small.copy_(saved_small)
    assert bool((dpre[n:] == 7).all()) and bool((da2[n:] == 7).all())
    # d b2 = column sums of d a2, d b1 = column sums of d pre, on top of the 1.0 that was there
    assert torch.allclose(small[2560:2816] - 1.0, da2[:n].float().sum(0), rtol=1e-4, atol=1e-4)
    assert torch.allclose(small[:512] - 1.0, dpre[:n].float().sum(0), rtol=1e-4, atol=1e-4)
    assert torch.allclose((small[512:2560] - 1.0).view(512, 4), dpre[:n].float().t() @ ex, rtol=1e-4, atol=1e-4)
    assert torch.allclose((small[2816:4480] - 1.0).view(13, 128), dl.t() @ a2[:, :128].float(), rtol=1e-4, atol=1e-4)
    assert torch.allclose(small[4608:4621] - 1.0, dl.sum(0), rtol=1e-4, atol=1e-4) and torch.allclose(small[4621:4622] - 1.0, dv.sum(0), rtol=1e-4, atol=1e-4)
    assert float(small[4622]) == 1.0 and float(small[4623]) == 1.0


def test_pack_fc1_is_the_feature_order_permutation_and_transpose_is_a_transpose():
    import ctypes as C
    import torch
    from inversus_reinforcement_learning_amd import _lib as L
    from inversus_reinforcement_learning_amd import heads as HD
    pol = _policy(4)
    w = pol.fc_actor[0].weight.detach()
    wf, we = HD._pack_fc1([w], 150)
    want = w[:, :19200].reshape(256, 128, 150).permute(0, 2, 1).reshape(256, 19200).to(torch.bfloat16)
    assert torch.equal(wf, want) and torch.equal(we, w[:, 19200:])
    lib = L.load()                                               # (the gradient's way back is inv_fc1_dw's epilogue: tests/test_gpu_fc1.py)
    for rows, cols in ((256, 19200), (70, 130)):
        src = torch.randn(rows, cols, device="cuda").to(torch.bfloat16); dst = torch.empty((cols, rows), dtype=torch.bfloat16, device="cuda")
        L.raise_for(lib.inv_transpose_bf16(C.c_void_p(src.data_ptr()), rows, cols, C.c_void_p(dst.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)), lib)
        assert torch.equal(dst, src.t())


def test_bad_arguments_are_refused():
    import ctypes as C
    from inversus_reinforcement_learning_amd import _lib as L
    lib = L.load()
    p = L.InvHeadsParams()
    assert lib.inv_heads_fwd(C.byref(p), None, 1, None, None, None, None, None, 2, 4, None) == L.INV_ERR_BAD_ARG
    assert lib.inv_heads_fwd(C.byref(p), None, 1, None, None, None, None, None, 3, 4, None) == L.INV_ERR_BAD_ARG
    lst = L.InvHeadsPackList()
    assert lib.inv_heads_pack(C.byref(lst), 9, None) == L.INV_ERR_BAD_ARG and lib.inv_heads_pack(C.byref(lst), 1, None) == L.INV_ERR_BAD_ARG
    assert lib.inv_heads_pack(C.byref(lst), 0, None) == L.INV_OK


def test_other_head_sizes_keep_the_torch_tail_and_the_autograd_update():
    """hidden_dim != 256 has no heads kernel: FusedPolicy / FusedTrainPolicy keep their torch tails (still on the GPU, still the
    fused trunk) and PPOAgent's fused update goes through autograd instead of loss_and_grads."""
    import torch
    from inversus_reinforcement_learning_amd import heads as HD
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
    from inversus_reinforcement_learning_amd.training import RolloutCollector
    torch.manual_seed(0)
    pol = InversusCNNPolicy(12, 10, 15, 4, hidden_dim=128).cuda()
    assert not HD.supported(pol)
    n, T = 128, 4
    b = DeviceEnvBatch(n, opponent="dummy", difficulty="hard", opp_seed=1)
    b.seed(np.arange(n, dtype=np.int64)); b.reset()
    g, x = b.render(0)
    f = FusedPolicy(pol)
    assert f.heads is None
    with torch.no_grad():
        lo, v = f(g, x)
        lm, vm = pol(g, x)
    assert float((lo - lm).abs().max()) < 0.05 and float((v - vm).abs().max()) < 0.05
    agent = PPOAgent(pol, fused_act=True, fused_update=True, batch_size=n * T, epochs=1)
    assert not agent.direct_grads
    before = float(sum(p.detach().double().abs().sum() for p in pol.parameters()))
    RolloutCollector(b, agent).collect(T)
    stats = agent.update()
    after = float(sum(p.detach().double().abs().sum() for p in pol.parameters()))
    assert np.isfinite(stats["policy_loss"]) and np.isfinite(stats["entropy"]) and after != before
