"""GPU tests of the first Linear's three products as kernels of this library (csrc/inv_fc1.hip; reference: the two
Linear(128 P + 4, 256) of inversus_rl/policies.py:60-75 and their autograd derivatives, ppo_agent.py:211-232) and of inv_heads_dw2.

bf16 operands, f32 accumulation.  The parity check is EXACT: small-integer operands make every partial sum an integer below 2^24,
so any summation order gives the same f32 bits and a wrong tile, a raced LDS read or a dropped K-tile shows as a mismatch
somewhere in the full output; torch's f32 matmul of the same integers is the checker (test infrastructure).  Random-data runs
bound the rounding behaviour (f32 accumulation: 1e-6 relative)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


def _lib():
    from inversus_reinforcement_learning_amd import _lib as L
    return L, L.load()


def _st():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _ints(shape, lo, hi, seed):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, device="cuda", generator=g).to(torch.bfloat16)


def _fwd(feat, wf):
    import torch
    L, lib = _lib()
    B, K = feat.shape
    nout = wf.shape[0]
    slabs = int(lib.inv_fc1_fwd_slabs(B, nout, K))
    assert slabs >= 1
    out = torch.full((slabs, B, nout), float("nan"), dtype=torch.float32, device="cuda")
    L.raise_for(lib.inv_fc1_fwd(_p(feat), _p(wf), B, nout, K, _p(out), _st()), lib)
    return out


@pytest.mark.parametrize("B,nout,K", [(1, 512, 19200), (37, 256, 19200), (256, 512, 19200), (700, 512, 19200), (2048, 256, 1024),
                                      (300, 512, 131072), (4096, 512, 19200)])
def test_forward_product_is_exact_on_integer_operands(B, nout, K):
    import torch
    feat, wf = _ints((B, K), -2, 2, 1), _ints((nout, K), -3, 3, 2)
    wf[:, ::7] = 0                                          # asymmetric operands: a transposed or permuted tile cannot pass
    got = _fwd(feat, wf).sum(0)
    want = feat.float() @ wf.float().t()
    assert torch.equal(got, want)


def test_forward_product_at_full_size_is_exact_repeatable_and_independent_of_the_batch():
    import torch
    B, nout, K = 16384, 512, 19200
    feat, wf = _ints((B, K), -2, 2, 3), _ints((nout, K), -3, 3, 4)
    a = _fwd(feat, wf)
    assert a.shape[0] == 2                                 # 64 x 2 tiles x 2 K-halves = 256 workgroups on a 256-CU part
    want = feat.float() @ wf.float().t()
    assert torch.equal(a.sum(0), want)
    # random data: bit-identical between launches (fixed K order per output element, no atomics), close to the f32 product
    g = torch.Generator(device="cuda").manual_seed(5)
    feat = (torch.randn(B, K, device="cuda", generator=g).clamp_min(0) * 0.5).to(torch.bfloat16)
    wf = (torch.randn(nout, K, device="cuda", generator=g) * 0.01).to(torch.bfloat16)
    r1, r2 = _fwd(feat, wf), _fwd(feat, wf)
    assert torch.equal(r1, r2)
    ref = feat[:2048].float() @ wf.float().t()
    got = (r1[0] + r1[1])[:2048]
    assert float((got - ref).norm() / ref.norm()) < 2e-6
    # a sample's row does not depend on the rest of the batch when the K split is the same (another 16 384-row batch, rows permuted)
    perm = torch.randperm(B, device="cuda", generator=g)
    r3 = _fwd(feat[perm].contiguous(), wf)
    assert torch.equal(r3[:, torch.argsort(perm)], r1)


@pytest.mark.parametrize("B,nfeat", [(1, 19200), (45, 19200), (256, 19200), (1000, 19200), (513, 131072)])
def test_feature_gradient_product_is_exact_on_integer_operands(B, nfeat):
    import torch
    L, lib = _lib()
    dpre, wft = _ints((B, 512), -2, 2, 6), _ints((nfeat, 512), -1, 1, 7)
    wft[::5] = 0
    out = torch.full((B + 3, nfeat), 7.0, dtype=torch.bfloat16, device="cuda")          # rows past B must stay untouched
    L.raise_for(lib.inv_fc1_dfeat(_p(dpre), _p(wft), B, 512, nfeat, _p(out), _st()), lib)
    want = (dpre.float() @ wft.float().t()).to(torch.bfloat16)                          # exact f32 sums, one rounding
    assert torch.equal(out[:B], want) and bool((out[B:] == 7).all())


def _dw(dpre, feat, dwe, P):
    """inv_fc1_dw through the C ABI: dpre bf16 [B][512] (transposed + zero-padded here), feat bf16 [B][128 P] -> two f32 [256][128 P + 4]."""
    import torch
    L, lib = _lib()
    B = dpre.shape[0]
    ld = (B + 63) // 64 * 64
    dpt = torch.zeros((512, ld), dtype=torch.bfloat16, device="cuda")
    dpt[:, :B] = dpre.t()
    g = [torch.full((256, 128 * P + 4), float("nan"), dtype=torch.float32, device="cuda") for _ in range(2)]
    ws = torch.empty(int(lib.inv_fc1_dw_workspace_bytes(B, 512, 128 * P)), dtype=torch.uint8, device="cuda")
    L.raise_for(lib.inv_fc1_dw(_p(dpt), _p(feat), _p(dwe), B, 512, P, _p(g[0]), _p(g[1]), _p(ws), _st()), lib)
    return g


def _dw_want(dpre, feat, dwe, P):
    import torch
    d = dpre.float().t() @ feat.float()                                                  # [512][p * 128 + c]
    d = d.reshape(512, P, 128).permute(0, 2, 1).reshape(512, 128 * P)                    # the module's c * P + p
    full = torch.cat([d, dwe], dim=1)
    return [full[:256], full[256:]]


@pytest.mark.parametrize("B,P", [(1, 150), (100, 150), (700, 150), (4096, 150), (192, 8), (130, 1024)])
def test_weight_gradient_is_exact_and_lands_in_the_modules_layout(B, P):
    import torch
    dpre, feat = _ints((B, 512), -2, 2, 8), _ints((B, 128 * P), -3, 3, 9)
    dpre[:, ::3] = 0
    dwe = torch.randn(512, 4, device="cuda")
    got, want = _dw(dpre, feat, dwe, P), _dw_want(dpre, feat, dwe, P)
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_weight_gradient_at_full_size():
    import torch
    B, P = 16384, 150
    dpre, feat = _ints((B, 512), -2, 2, 10), _ints((B, 128 * P), -3, 3, 11)
    dwe = torch.randn(512, 4, device="cuda")
    got, want = _dw(dpre, feat, dwe, P), _dw_want(dpre, feat, dwe, P)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    again = _dw(dpre, feat, dwe, P)
    assert all(torch.equal(a, b) for a, b in zip(got, again))


@pytest.mark.parametrize("B", [1, 45, 256, 1000, 16384])
def test_second_linear_weight_gradient(B):
    import torch
    L, lib = _lib()
    da2, hs = _ints((B, 256), -2, 2, 12), _ints((B, 512), -2, 2, 13)
    g = [torch.full((128, 256), 1.0, dtype=torch.float32, device="cuda") for _ in range(2)]   # the kernel ADDS
    L.raise_for(lib.inv_heads_dw2(_p(da2), _p(hs), B, _p(g[0]), _p(g[1]), _st()), lib)
    for h in range(2):
        want = da2[:, 128 * h:128 * h + 128].float().t() @ hs[:, 256 * h:256 * h + 256].float() + 1.0
        assert torch.equal(g[h], want)


def test_bad_arguments_are_refused():
    L, lib = _lib()
    assert lib.inv_fc1_fwd_slabs(16, 500, 19200) == 0 and lib.inv_fc1_fwd_slabs(16, 512, 100) == 0
    assert lib.inv_fc1_fwd(None, None, 4, 512, 19200, None, None) == L.INV_ERR_BAD_ARG
    assert lib.inv_fc1_dfeat(None, None, 4, 512, 19200, None, None) == L.INV_ERR_BAD_ARG
    assert lib.inv_fc1_dw(None, None, None, 4, 512, 150, None, None, None, None) == L.INV_ERR_BAD_ARG
    assert lib.inv_fc1_dw_workspace_bytes(4, 300, 19200) == 0
    assert lib.inv_heads_dw2(None, None, 4, None, None, None) == L.INV_ERR_BAD_ARG
