"""CPU tests of the host-side math around the fused update kernels (fused_train.py): the packings and the torch
restatements that the GPU tests use as cross-checks are themselves checked against torch autograd of the reference's
layer (conv3x3 pad 1, inversus_rl/policies.py:29-42).  No GPU, no HIP library calls."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from inversus_reinforcement_learning_amd import fused_train as FT
from inversus_reinforcement_learning_amd.fused_policy import _pack_conv


def test_im2col_weight_gradient_equals_conv2d_autograd():
    torch.manual_seed(0)
    n, cin, cout = 5, 12, 32
    x = torch.randn(n, cin, 10, 15)
    w = torch.randn(cout, cin, 3, 3, requires_grad=True)
    dz = torch.randn(n, cout, 10, 15)
    F.conv2d(x, w, padding=1).backward(dz)
    # channels-last operands, as the kernels keep them; bf16-representable values so that the bf16 GEMM is exact enough
    xs = x.permute(0, 2, 3, 1).reshape(n, 150, cin)
    ds = dz.permute(0, 2, 3, 1).reshape(n, 150, cout)
    got = FT._wgrad_torch(xs.to(torch.bfloat16), ds.to(torch.bfloat16), cin, cout)
    ref = FT._wgrad_torch(xs.to(torch.bfloat16).float().to(torch.bfloat16), ds.to(torch.bfloat16), cin, cout)
    assert torch.equal(got, ref) and tuple(got.shape) == (cout, cin, 3, 3)
    rel = float((got - w.grad).norm() / w.grad.norm())
    assert rel < 2e-2, rel                         # bf16 rounding of the operands only; the tap / channel order is exact


def test_transposed_weights_give_the_data_gradient():
    """Wt[ci][co][ky][kx] = W[co][ci][2-ky][2-kx] turns the data gradient into a plain conv3x3 (pad 1) over dz --
    the packing inv_policy_trunk_bwd is fed with."""
    torch.manual_seed(1)
    n, cin, cout = 3, 32, 64
    x = torch.randn(n, cin, 10, 15, requires_grad=True)
    w = torch.randn(cout, cin, 3, 3)
    dz = torch.randn(n, cout, 10, 15)
    F.conv2d(x, w, padding=1).backward(dz)
    wt = w.flip(2, 3).permute(1, 0, 2, 3)
    got = F.conv2d(dz, wt, padding=1)
    assert torch.allclose(got, x.grad, atol=1e-4, rtol=1e-4)


def test_weight_fragment_packing_layout():
    """bf16 [Cout/16][9*Cin_pad/32][64][8]: lane 16*g + r of fragment (mt, ks) holds W[16*mt + r][k = 32*ks + 8*g + j],
    k = (3*ky + kx) * Cin_pad + c (include/inversus_hip.h)."""
    cout, cin, cin_pad = 32, 12, 32
    w = torch.arange(cout * cin * 9, dtype=torch.float32).reshape(cout, cin, 3, 3) % 251      # bf16-exact integers
    pk = _pack_conv(w, cin_pad).float()
    assert tuple(pk.shape) == (cout // 16, 9 * cin_pad // 32, 64, 8)
    rng = np.random.RandomState(0)
    for _ in range(200):
        mt, ks, lane, j = rng.randint(cout // 16), rng.randint(9 * cin_pad // 32), rng.randint(64), rng.randint(8)
        r, g = lane & 15, lane >> 4
        k = 32 * ks + 8 * g + j
        tap, c = divmod(k, cin_pad)
        want = float(w[16 * mt + r, c, tap // 3, tap % 3]) if c < cin else 0.0
        assert float(pk[mt, ks, lane, j]) == want


def test_layernorm_parameter_packing_layout():
    """fp16 [150][C/4][weight x4 | bias x4], pixel-major."""
    c = 64
    g = torch.randn(c, 10, 15)
    b = torch.randn(c, 10, 15)
    pk = FT._pack_ln(g, b, c)
    assert pk.dtype == torch.float16 and tuple(pk.shape) == (150, c // 4, 8)
    for p, ch in ((0, 0), (149, 63), (77, 13), (15, 4)):
        y, x = divmod(p, 15)
        assert float(pk[p, ch // 4, ch % 4]) == float(g[ch, y, x].half())
        assert float(pk[p, ch // 4, 4 + ch % 4]) == float(b[ch, y, x].half())


def test_fused_train_policy_needs_a_gpu_policy():
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    with pytest.raises(RuntimeError):
        FT.FusedTrainPolicy(InversusCNNPolicy(12, 10, 15, 4))
    with pytest.raises(RuntimeError):
        FT.FusedTrainPolicy(InversusCNNPolicy(12, 32, 32, 4))             # round 3: the 32 x 32 board trains through the banded kernels
    with pytest.raises(ValueError):
        FT.FusedTrainPolicy(InversusCNNPolicy(12, 16, 16, 4))             # no fused trunk for other boards
