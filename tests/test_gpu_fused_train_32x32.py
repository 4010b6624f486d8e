"""GPU tests of the BANDED trunk's training path (BASELINE config C5 with the fused update; csrc/inv_policy.hip: the SAVE variant
of k_policy_trunk_banded, k_band_ln_bwd, k_band_conv_t, k_band_act, k_policy_wgrad<.., WgGeoBand>, k_band_param_grads;
fused_train.py::_TrunkTrainBandedFn) against torch on the f32 module (the reference's update runs autograd over
inversus_rl/policies.py:91-106 at H = W = 32).  bf16 operands / f32 accumulation: relative L2 errors per tensor."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
H = W = 32
P = H * W


def _policy(seed=0):
    import torch
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    torch.manual_seed(seed)
    pol = InversusCNNPolicy(12, H, W, 4).cuda()
    with torch.no_grad():                                      # non-trivial LayerNorm affine
        for nm in (pol.norm1, pol.norm2, pol.norm3, pol.norm4):
            nm.weight.add_(0.2 * torch.randn_like(nm.weight)); nm.bias.add_(0.2 * torch.randn_like(nm.bias))
    return pol


def _obs(n, seed=0):
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    b = DeviceEnvBatch(n, W, H, opponent="selfplay")
    b.seed(seed + np.arange(n, dtype=np.int64)); b.reset()
    rng = np.random.RandomState(seed)
    for _ in range(12):
        b.step(rng.randint(0, 13, n).astype(np.uint8), rng.randint(0, 13, n).astype(np.uint8), auto_reset=True, obs=False)
    gf, x = b.render(0)
    gb, _ = b.render(0, bits=True)
    return gf, gb, x


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


def _st():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("n", [2, 9])
def test_band_wgrad_matches_im2col_gemm(n):
    """The weight-gradient kernel on band units (128 dz pixels against the 6 surrounding rows of x, rows outside the board staged
    as zeros) vs the same contraction by torch in f32 on the same bf16 operands."""
    import torch
    from inversus_reinforcement_learning_amd import _lib as L, fused_train as FT
    lib = L.load()
    torch.manual_seed(n)
    for layer in range(4):
        cin, cinp, cout = FT._CIN[layer], FT._CINP[layer], FT._COUT[layer]
        x = torch.zeros(n, P, cinp, device="cuda")
        x[:, :, :cin] = torch.randn(n, P, cin, device="cuda")
        x = x.to(torch.bfloat16)
        dz = (torch.randn(n, P, cout, device="cuda") * 0.1).to(torch.bfloat16)
        dw = torch.zeros(cout, 9, cinp, device="cuda")
        db = torch.zeros(cout, device="cuda")
        L.raise_for(lib.inv_policy_band_wgrad(layer, H, W, _p(x), _p(dz), _p(dw), _p(db), n, _st()), lib)
        got = dw[:, :, :cin].reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
        xf = torch.nn.functional.pad(x[..., :cin].float().reshape(n, H, W, cin), (0, 0, 1, 1, 1, 1))
        cols = xf.unfold(1, 3, 1).unfold(2, 3, 1).permute(0, 1, 2, 4, 5, 3).reshape(n * P, 9 * cin)
        want = (dz.float().reshape(n * P, cout).t() @ cols).reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
        assert rel(got, want) < 2e-5, (layer, rel(got, want))
        want_b = dz.float().sum(dim=(0, 1))
        assert float((db - want_b).abs().max()) < 1e-3 * max(1.0, float(want_b.abs().max())), layer
        if cinp > cin:
            assert float(dw[:, :, cin:].abs().max()) == 0.0                # the zero channels of x0 contribute nothing


@pytest.mark.parametrize("layer", [1, 2, 3])
def test_band_conv_t_matches_the_data_gradient_of_conv2d(layer):
    import torch
    from inversus_reinforcement_learning_amd import _lib as L, fused_train as FT
    from inversus_reinforcement_learning_amd.fused_policy import pack_conv
    lib = L.load()
    torch.manual_seed(layer)
    n, cin, cout = 3, FT._CIN[layer], FT._COUT[layer]
    w = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.1)
    dz = (torch.randn(n, P, cout, device="cuda") * 0.5).to(torch.bfloat16)
    wt = pack_conv(w, cout, transposed=True)
    zero = torch.zeros(128, device="cuda")
    da = torch.empty(n, P, cin, dtype=torch.bfloat16, device="cuda")
    L.raise_for(lib.inv_policy_band_conv_t(layer, H, W, _p(wt), _p(zero), _p(dz), _p(da), n, _st()), lib)
    g = dz.float().reshape(n, H, W, cout).permute(0, 3, 1, 2)
    want = torch.nn.grad.conv2d_input((n, cin, H, W), w.to(torch.bfloat16).float(), g, padding=1)
    if layer == 3:
        want = want + g                                                     # the skip connection around conv4
    want = want.permute(0, 2, 3, 1).reshape(n, P, cin)
    assert rel(da, want) < 6e-3, rel(da, want)                              # bf16 output rounding


@pytest.mark.parametrize("layer,n", [(1, 3), (2, 5), (3, 700)])
def test_band_conv_t_sums_leaves_the_data_gradient_and_the_layer_belows_two_sums(layer, n):
    """ABI 7: conv_t(l) with the LayerNorm-backward sums of layer l - 1 taken from its write-back pass.  The data gradient is the
    plain entry point's bit for bit; the sums are inv_policy_band_ln_sums' on that tensor up to the order of the f32 additions.
    n = 700 > the grid: a workgroup handles several samples, so its accumulators and the LDS reduction are reused."""
    import torch
    from inversus_reinforcement_learning_amd import _lib as L, fused_train as FT
    from inversus_reinforcement_learning_amd.fused_policy import pack_conv, pack_ln
    lib = L.load()
    torch.manual_seed(20 + layer)
    cin, cout = FT._CIN[layer], FT._COUT[layer]
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.1
    dz = (torch.randn(n, P, cout, device="cuda") * 0.5).to(torch.bfloat16)
    wt = pack_conv(w, cout, transposed=True)
    zero = torch.zeros(128, device="cuda")
    zb = (torch.randn(n, P, cin, device="cuda") * 1.5 + 0.3).to(torch.bfloat16)          # z of the layer below
    gw = (1.0 + 0.2 * torch.randn(cin, H, W, device="cuda")).half().float()
    gb = (0.2 * torch.randn(cin, H, W, device="cuda")).half().float()
    ln = pack_ln(gw, gb, cin)
    zf = zb.float()
    mean = zf.mean(dim=(1, 2))
    rstd = 1.0 / torch.sqrt(zf.var(dim=(1, 2), unbiased=False) + 1e-5)
    stats = torch.zeros(n, 4, 2, device="cuda")
    stats[:, layer - 1, 0], stats[:, layer - 1, 1] = rstd, -mean * rstd
    da = torch.empty(n, P, cin, dtype=torch.bfloat16, device="cuda")
    L.raise_for(lib.inv_policy_band_conv_t(layer, H, W, _p(wt), _p(zero), _p(dz), _p(da), n, _st()), lib)
    want = torch.full((n, 2), float("nan"), device="cuda")
    L.raise_for(lib.inv_policy_band_ln_sums(layer - 1, H, W, _p(da), _p(zb), _p(stats), _p(ln), _p(want), n, _st()), lib)
    da2 = torch.full_like(da, 7.0)
    got = torch.full((n, 2), float("nan"), device="cuda")
    L.raise_for(lib.inv_policy_band_conv_t_sums(layer, H, W, _p(wt), _p(zero), _p(dz), _p(da2), _p(zb), _p(stats), _p(ln), _p(got), n, _st()), lib)
    assert torch.equal(da2, da)
    # a sum of ~1e5 terms of mixed sign in f32: compare against the size of the terms, not of the (cancelling) total
    scale = (da.float().abs().sum(dim=(1, 2)) * float(gw.abs().max()) * 5.0).clamp_min(1.0)       # |x^| < 5 on these inputs
    assert float(((got - want).abs().max(dim=1).values / scale).max()) < 1e-5, ((got - want).abs().max(), scale.min())
    with pytest.raises(ValueError, match="NULL tensor"):
        L.raise_for(lib.inv_policy_band_conv_t_sums(layer, H, W, _p(wt), _p(zero), _p(dz), _p(da2), None, _p(stats), _p(ln), _p(got), n, _st()), lib)


@pytest.mark.parametrize("layer", [0, 3])
def test_band_ln_bwd_act_and_param_grads_match_autograd(layer):
    import torch
    from inversus_reinforcement_learning_amd import _lib as L, fused_train as FT
    from inversus_reinforcement_learning_amd.fused_policy import pack_ln
    lib = L.load()
    torch.manual_seed(10 + layer)
    n, c = 5, FT._COUT[layer]
    z = (torch.randn(n, P, c, device="cuda") * 1.5 + 0.3).to(torch.bfloat16)
    da = (torch.randn(n, P, c, device="cuda") * 0.01).to(torch.bfloat16)
    gw = (1.0 + 0.2 * torch.randn(c, H, W, device="cuda")).half().float()   # fp16-representable, like the packed form
    gb = (0.2 * torch.randn(c, H, W, device="cuda")).half().float()
    ln = pack_ln(gw, gb, c)
    zf = z.float()
    mean = zf.mean(dim=(1, 2))
    rstd = 1.0 / torch.sqrt(zf.var(dim=(1, 2), unbiased=False) + 1e-5)
    stats = torch.zeros(n, 4, 2, device="cuda")
    stats[:, layer, 0], stats[:, layer, 1] = rstd, -mean * rstd
    dy, dz, act_k = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
    L.raise_for(lib.inv_policy_band_ln_bwd(layer, H, W, _p(da), _p(z), _p(stats), _p(ln), _p(dy), _p(dz), _p(act_k), n, _st()), lib)
    dz2 = torch.empty_like(z)                                              # dy and act are optional outputs: same dz without them
    L.raise_for(lib.inv_policy_band_ln_bwd(layer, H, W, _p(da), _p(z), _p(stats), _p(ln), None, _p(dz2), None, n, _st()), lib)
    assert torch.equal(dz, dz2)
    # torch: a = relu(LN(z) * w + b) on the same z, gradient da
    zt = zf.clone().requires_grad_(True)
    wcl, bcl = gw.reshape(c, P).t().contiguous().requires_grad_(True), gb.reshape(c, P).t().contiguous().requires_grad_(True)   # [P][c]
    xh = (zt - zt.mean(dim=(1, 2), keepdim=True)) / torch.sqrt(zt.var(dim=(1, 2), unbiased=False, keepdim=True) + 1e-5)
    a = torch.relu(xh * wcl + bcl)
    a.backward(da.float())
    assert rel(dz, zt.grad) < 8e-3, rel(dz, zt.grad)
    mask = (xh * wcl + bcl) > 0
    assert rel(dy, da.float() * mask) < 2e-3
    assert rel(act_k, a.detach()) < 5e-3                                      # the activation pass A writes on the way
    if layer < 3:
        act = torch.empty_like(z)
        L.raise_for(lib.inv_policy_band_act(layer, H, W, _p(z), _p(stats), _p(ln), _p(act), n, _st()), lib)
        assert torch.equal(act, act_k)                                       # == the stand-alone activation kernel, bit for bit
    part = torch.empty(4, 2, P, c, device="cuda")
    L.raise_for(lib.inv_policy_band_param_grads(layer, _p(da), _p(z), _p(stats), _p(ln), _p(part), 4, P * c, n, _st()), lib)
    red = part.sum(0)
    assert rel(red[0], wcl.grad) < 3e-3 and rel(red[1], bcl.grad) < 3e-3, (rel(red[0], wcl.grad), rel(red[1], bcl.grad))
    # round 4: the two-pass form (sample sums, then dz + activation + parameter sums in ONE pass) is the four-pass form bit for bit
    sums = torch.full((n, 2), float("nan"), device="cuda")
    L.raise_for(lib.inv_policy_band_ln_sums(layer, H, W, _p(da), _p(z), _p(stats), _p(ln), _p(sums), n, _st()), lib)
    dz3, act3 = torch.full_like(z, 7.0), torch.full_like(z, 7.0)
    part3 = torch.full((4, 2, P, c), float("nan"), device="cuda")
    L.raise_for(lib.inv_policy_band_ln_bwd2(layer, H, W, _p(da), _p(z), _p(stats), _p(ln), _p(sums), _p(dz3), _p(act3), _p(part3), 4, n, _st()), lib)
    assert torch.equal(dz3, dz) and torch.equal(act3, act_k) and torch.equal(part3, part)
    dz4 = torch.empty_like(z)                                              # act is optional; more slices than samples leave zero partials
    part4 = torch.full((8, 2, P, c), float("nan"), device="cuda")
    L.raise_for(lib.inv_policy_band_ln_bwd2(layer, H, W, _p(da), _p(z), _p(stats), _p(ln), _p(sums), _p(dz4), None, _p(part4), 8, n, _st()), lib)
    assert torch.equal(dz4, dz) and bool((part4[n:] == 0).all()) and rel(part4.sum(0), red) < 1e-6


@pytest.mark.parametrize("n", [3, 40])
def test_banded_training_forward_is_the_inference_forward_and_close_to_the_module(n):
    import torch
    from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
    from inversus_reinforcement_learning_amd.fused_train import FusedTrainPolicy
    pol = _policy()
    gf, gb, x = _obs(n)
    ft, fi = FusedTrainPolicy(pol), FusedPolicy(pol, train_numerics=True)
    with torch.no_grad():
        f_train = ft.features(gb)
        assert torch.equal(f_train, ft.features(gf))
        assert torch.equal(f_train, fi.features(gb))           # rollouts and the update's first forward: bit for bit
        want = pol.trunk(gf).reshape(n, 128, P).permute(0, 2, 1).reshape(n, 128 * P)
    assert (f_train.float() - want).abs().max() < 0.25 and (f_train.float() - want).abs().mean() < 0.008
    lt, vt = ft(gb, x)
    with torch.no_grad():
        lm, vm = pol(gf, x)
    assert (lt - lm).abs().max() < 0.08 and (vt - vm).abs().max() < 0.08


@pytest.mark.parametrize("n,packed", [(6, True), (20, False)])
def test_parameter_gradients_of_the_32x32_policy_match_f32_autograd(n, packed):
    """The same bar as tests/test_gpu_fused_train.py::test_parameter_gradients_match_f32_autograd on the 15 x 10 board: per tensor,
    the fused path's relative L2 error against f32 autograd stays in the class of torch's own bf16 autocast."""
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

    want = grads(lambda: pol(gf, x))
    scale = {k: rel(g, want[k]) for k, g in grads(autocast_fwd).items()}
    got = grads(lambda: FusedTrainPolicy(pol)(gb if packed else gf, x))
    errs = {k: rel(g, want[k]) for k, g in got.items()}
    assert set(errs) == set(want) and all(np.isfinite(list(errs.values())))
    for k in errs:
        if k.startswith("fc_"):
            # the heads kernels (tests/test_gpu_heads.py: < 12 % against the plain f32 module -- ReLU masks flip where a
            # pre-activation is within a bf16 rounding of zero; their inputs here are sums over 131 072 features)
            assert errs[k] <= max(3.0 * scale[k], 0.05) and errs[k] < 0.12, (k, errs[k], scale[k])
        else:
            assert errs[k] <= max(2.0 * scale[k], 0.03), (k, errs[k], scale[k])
        assert errs[k] < 0.2, (k, errs[k])


def test_c5_trains_through_the_fused_update():
    """PPOAgent(fused_act, fused_update) on the 32 x 32 board: self-play collection with packed observations, one update without
    the autograd engine (direct gradients), finite statistics, parameters moved, and the update's first forward reproduces the
    rollout's log-probs (the PPO ratio starts at 1)."""
    import torch
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
    from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
    from inversus_reinforcement_learning_amd.training import RolloutCollector
    torch.manual_seed(0)
    n, T = 96, 6
    b = DeviceEnvBatch(n, W, H, opponent="selfplay")
    b.seed(np.arange(n, dtype=np.int64)); b.reset()
    agent = PPOAgent(InversusCNNPolicy(12, H, W, 4), fused_act=True, fused_update=True, batch_size=n * T // 2, epochs=1, lr=1e-4)
    assert agent.direct_grads and agent.fused.banded
    col = RolloutCollector(b, agent, FusedPolicy(InversusCNNPolicy(12, H, W, 4).cuda().eval(), actor_only=True))
    col.collect(T)
    grid, extra = torch.cat(agent.obs_grid_buffer), torch.cat(agent.obs_extra_buffer)
    old_lp, acts = torch.cat(agent.log_prob_buffer), torch.cat(agent.action_buffer)
    with torch.no_grad():
        lo, _ = agent.train_fwd(grid[:n], extra[:n])
        lp = torch.log_softmax(lo.float(), dim=1).gather(1, acts[:n, None]).squeeze(1)
    assert float((lp - old_lp[:n]).abs().max()) < 1e-5
    before = agent.flat_adam.flat.clone()
    stats = agent.update(trace=True)
    assert all(np.isfinite(v) for v in (stats["policy_loss"], stats["value_loss"], stats["entropy"]))
    assert np.all(np.isfinite(stats["minibatches"]["grad_norm"])) and np.all(stats["minibatches"]["grad_norm"] > 0)
    assert float((agent.flat_adam.flat - before).abs().max()) > 0
