"""Fused forward of InversusCNNPolicy for rollouts.

The convolutional trunk (conv1..4 + LayerNorm + ReLU + residual, policies.py:91-99) runs in ONE
HIP kernel per batch (csrc/inv_policy.hip: bf16 MFMA, activations resident in LDS); the two MLP heads
run as plain bf16/f32 GEMMs (one concatenated [512 x 19204] first layer for actor and critic).
Weights are re-packed from the f32 `nn.Module` by `refresh()` -- call it after every optimizer step
you want the actor to see.  Inference only: the PPO update keeps using the autograd module.

Numerics: bf16 operands with f32 accumulation, i.e. NOT bit-equal to the f32 module; logits agree to
~1e-2 (tests/test_gpu_policy.py).  Boards: the reference's 15 x 10 (activations resident in LDS) and the 32 x 32 stress board
of BASELINE config C5 (inference only: the "banded" kernel streams the layers' pre-norm outputs through a scratch buffer).
"""
import ctypes as C

import torch

from . import _lib as L
from . import heads as HD

_CIN_PAD = (32, 32, 64, 128)


def _pack_conv(w, cin_pad):
    """[Cout, Cin, 3, 3] f32 -> bf16 [Cout/16][9*cin_pad/32][64][8] MFMA A fragments (see inversus_hip.h)."""
    cout, cin = w.shape[:2]
    wp = torch.zeros(cout, 3, 3, cin_pad, dtype=torch.float32, device=w.device)
    wp[..., :cin] = w.permute(0, 2, 3, 1)                       # k = (3*ky + kx) * cin_pad + c
    k = 9 * cin_pad
    wk = wp.reshape(cout // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4)   # [mt][ks][g][r][8], lane = 16*g + r
    return wk.reshape(cout // 16, k // 32, 64, 8).contiguous().to(torch.bfloat16)


def pack_conv(w, cin_pad, transposed=False, out=None):
    """_pack_conv (transposed=False) or the data-gradient form Wt[ci][8 - tap][co] = W[co][ci][tap] (transposed=True) as ONE kernel
    (inv_policy_pack_conv) instead of five torch index kernels; `w`: f32 [Cout, Cin, 3, 3] on the GPU.  out: re-pack in place."""
    lib = L.load()
    w = w.detach().float().contiguous()
    cout, cin = int(w.shape[0]), int(w.shape[1])
    rows, kc = (cin, cout) if transposed else (cout, int(cin_pad))
    shape = (rows // 16, 9 * kc // 32, 64, 8)
    if out is None:
        out = torch.empty(shape, dtype=torch.bfloat16, device=w.device)
    elif tuple(out.shape) != shape or out.dtype != torch.bfloat16 or not out.is_contiguous():
        raise ValueError(f"pack_conv: out must be contiguous bf16 {shape}")
    with torch.cuda.device(w.device):
        rc = lib.inv_policy_pack_conv(C.c_void_p(w.data_ptr()), cout, cin, int(cin_pad), int(bool(transposed)), C.c_void_p(out.data_ptr()),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
    L.raise_for(rc, lib)
    return out


def pack_ln(weight, bias, cout, out=None):
    """LayerNorm([C,H,W]) affine -> fp16 [H*W][C/4][weight x4 | bias x4] as one kernel (inv_policy_pack_ln).  out: re-pack in place."""
    lib = L.load()
    w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
    pixels = w.numel() // int(cout)
    if out is None:
        out = torch.empty((pixels, cout // 4, 8), dtype=torch.float16, device=w.device)
    elif tuple(out.shape) != (pixels, cout // 4, 8) or out.dtype != torch.float16 or not out.is_contiguous():
        raise ValueError(f"pack_ln: out must be contiguous fp16 {(pixels, cout // 4, 8)}")
    with torch.cuda.device(w.device):
        rc = lib.inv_policy_pack_ln(C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), int(cout), pixels, C.c_void_p(out.data_ptr()),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
    L.raise_for(rc, lib)
    return out


class FusedPolicy:
    """Callable like the module in eval mode: (grid f32 [B,12,10,15] or packed int32 [B,57], extra f32 [B,4]) -> (logits, value)."""

    def __init__(self, policy, train_numerics=False, actor_only=False):
        """actor_only: compute the action logits only (value = None) -- the frozen self-play opponent never needs its critic, and
        without it the first Linear of the heads is a [256 x 19204] GEMM instead of [512 x 19204] (half of the heads' time).
        train_numerics: run the trunk with the numerics of the fused UPDATE's forward (x^ rounded to bf16 before the
        LayerNorm affine, csrc/inv_policy.hip SAVE path) instead of the slightly more accurate inference epilogue, so that
        log-probs / values stored during rollouts are what the update's first forward recomputes (PPO ratio = 1)."""
        self.policy = policy
        self.train_numerics = bool(train_numerics)
        self.actor_only = bool(actor_only)
        self.lib = L.load()
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        p = self.policy
        dev = p.conv1.weight.device
        if dev.type != "cuda":
            raise RuntimeError("FusedPolicy needs the policy on the GPU")
        shape = tuple(p.norm1.normalized_shape)
        if shape not in ((32, 10, 15), (32, 32, 32)) or p.conv1.in_channels != 12:
            raise ValueError("the fused trunk supports 12-channel observations of the 15 x 10 and 32 x 32 boards only")
        self.H, self.W = shape[1], shape[2]
        self.P = self.H * self.W
        self.OW = (12 * self.P + 31) // 32
        self.banded = (self.H, self.W) != (10, 15)
        # (banded boards: the training forward IS the inference kernel with other destinations for its intermediates, so
        # train_numerics changes nothing there -- the rollout's features are the update's bit for bit either way)
        if self.banded:
            with torch.cuda.device(dev):                                 # the size depends on the CU count of THIS device
                self._scratch_bytes = int(self.lib.inv_policy_trunk_banded_scratch_bytes(self.H, self.W))
            if self._scratch_bytes <= 0:
                raise ValueError(f"no fused trunk for a {self.W} x {self.H} board")
        # Packed operands.  After the first call they are re-packed IN PLACE (same device pointers, same `params` struct): kernels
        # captured in a HIP graph (RolloutCollector(graph=True)) read the refreshed weights without a re-capture.  The conv bias is the
        # live f32 parameter itself.
        old = getattr(self, "_packed", None)
        packed, params = [], L.InvPolicyParams()
        for i, (conv, norm) in enumerate(((p.conv1, p.norm1), (p.conv2, p.norm2), (p.conv3, p.norm3), (p.conv4, p.norm4))):
            cout = conv.out_channels
            w = pack_conv(conv.weight, _CIN_PAD[i], out=None if old is None else old[i][0])
            b = conv.bias.detach().float().contiguous()
            if old is not None and b.data_ptr() != old[i][1].data_ptr():         # (a bias that is not plain contiguous f32: keep the first copy's address)
                old[i][1].copy_(b)
                b = old[i][1]
            # LayerNorm affine, [pixel][channel/4][weight x4 | bias x4] in fp16: one 16-byte load per accumulator tile
            ln = pack_ln(norm.weight, norm.bias, cout, out=None if old is None else old[i][2])
            packed.append((w, b, ln))
            params.layer[i] = L.InvPolicyLayer(w.data_ptr(), b.data_ptr(), ln.data_ptr())
        self._packed = packed
        self.params = params
        # heads: the reference's sizes run on the fused kernels (heads.py / csrc/inv_heads.hip) ...
        if HD.supported(p):
            tensors = HD.head_tensors(p, self.actor_only)
            if getattr(self, "heads", None) is not None:
                self.heads.repack(tensors)
            else:
                self.heads = HD.HeadsOperands(tensors, self.P)
            return
        self.heads = None
        # ... any other hidden_dim keeps the torch tail: first Linear of actor and critic concatenated; columns permuted from the
        # module's flatten order (c*150 + pixel) to the kernel's channels-last order (pixel*128 + c)
        firsts = [p.fc_actor[0]] if self.actor_only else [p.fc_actor[0], p.fc_critic[0]]
        w1 = torch.cat([m.weight for m in firsts]).detach().float()                         # [512 (256: actor only), 19204]
        hid = w1.shape[0]
        nf = 128 * self.P
        self.w1_feat = w1[:, :nf].reshape(hid, 128, self.P).permute(0, 2, 1).reshape(hid, nf).contiguous().to(torch.bfloat16)
        self.w1_extra = w1[:, nf:].contiguous()
        self.b1 = torch.cat([m.bias for m in firsts]).detach().float()
        self.hidden = p.fc_actor[0].out_features

    # The banded kernel's per-workgroup z scratch (134 MB on a 256-CU part).  It is only live inside one launch, and launches of
    # one stream run one after the other, so every policy on a (device, stream) pair shares ONE buffer (the learner and the frozen
    # self-play opponent used to hold one each).
    _banded_scratch = {}

    @classmethod
    def release_banded_scratch(cls):
        """Drop the cached scratch buffers (one per device and stream that ever ran the banded kernel)."""
        cls._banded_scratch.clear()

    @classmethod
    def _scratch_for(cls, device, nbytes):
        key = (device.index, int(torch.cuda.current_stream(device).cuda_stream))
        buf = cls._banded_scratch.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = cls._banded_scratch[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return buf

    @torch.no_grad()
    def features(self, grid):
        """Trunk only: bf16 [B, H*W*128] in channels-last order.  `grid`: f32 planes [B,12,H,W] or the packed
        observations [B, ceil(12*H*W/32)] int32 that the env writes under INV_STEP_OBS_BITS (32x fewer bytes to read)."""
        bits = grid.dtype == torch.int32
        if bits:
            if grid.dim() != 2 or grid.shape[1] != self.OW:
                raise ValueError(f"packed observations must be [B, {self.OW}] int32, got {tuple(grid.shape)}")
            grid, fn = grid.contiguous(), self.lib.inv_policy_trunk_bits
        else:
            if tuple(grid.shape[1:]) != (12, self.H, self.W):
                raise ValueError(f"observations must be [B, 12, {self.H}, {self.W}], got {tuple(grid.shape)}")
            grid, fn = grid.contiguous().float(), self.lib.inv_policy_trunk
        feat = torch.empty((grid.shape[0], 128 * self.P), dtype=torch.bfloat16, device=grid.device)
        with torch.cuda.device(grid.device):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            if self.banded:
                scratch = self._scratch_for(grid.device, self._scratch_bytes)
                rc = self.lib.inv_policy_trunk_banded(C.byref(self.params), C.c_void_p(grid.data_ptr()), int(bits), self.H, self.W,
                                                      C.c_void_p(feat.data_ptr()), C.c_void_p(scratch.data_ptr()), scratch.numel(),
                                                      grid.shape[0], st)
            elif self.train_numerics:                                  # 15 x 10 only
                rc = self.lib.inv_policy_trunk_train_fwd(C.byref(self.params), C.c_void_p(grid.data_ptr()), int(bits),
                                                         C.c_void_p(feat.data_ptr()), None, grid.shape[0], st)
            else:
                rc = fn(C.byref(self.params), C.c_void_p(grid.data_ptr()), C.c_void_p(feat.data_ptr()), grid.shape[0], st)
        L.raise_for(rc, self.lib)
        return feat

    @torch.no_grad()
    def __call__(self, grid, extra):
        p = self.policy
        feat = self.features(grid)
        if self.heads is not None:
            return self.heads.forward(feat, extra)
        h = (feat @ self.w1_feat.t()).float() + extra.float() @ self.w1_extra.t() + self.b1
        h = torch.relu(h)
        logits = p.fc_actor[4](torch.relu(p.fc_actor[2](h[:, :self.hidden])))
        if self.actor_only:
            return logits, None
        value = p.fc_critic[4](torch.relu(p.fc_critic[2](h[:, self.hidden:])))
        return logits, value
