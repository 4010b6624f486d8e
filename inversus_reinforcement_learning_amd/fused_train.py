"""Training path of InversusCNNPolicy through the fused trunk kernels.

`FusedTrainPolicy(policy)(grid, extra)` returns the same (logits, value) as the module and is differentiable
with respect to the module's parameters, but the convolutional trunk (policies.py:91-99) runs as

  forward   csrc/inv_policy.hip  k_policy_trunk<.., SAVE>   one kernel, activations in LDS, keeps x^ / act / rstd
  backward  csrc/inv_policy.hip  k_policy_trunk_bwd         one kernel: LayerNorm + ReLU backward and the data
                                                            gradient of the convolutions (transposed implicit GEMMs)

and the parameter gradients are batch reductions over the tensors that kernel writes (dy, dz; channels-last bf16):
LayerNorm weight / bias by one reduction kernel per layer (k_policy_param_grads), conv weights and biases by
k_policy_wgrad (MFMA over pixels through transposing LDS reads).
The two MLP heads stay in torch autograd; their 19 200-column first layer runs in bf16 on the kernel's feature order.

Numerics: bf16 operands with f32 accumulation, like `PPOAgent(update_autocast=torch.bfloat16)` -- gradients agree with
the f32 module to ~1e-2 relative (tests/test_gpu_fused_train.py), not bit for bit.  Only the 12 x 10 x 15 observation.
"""
import ctypes as C
import os

import torch
import torch.nn.functional as F

from . import _lib as L
from .device_env import unpack_obs
from . import heads as HD
from .fused_policy import _CIN_PAD, _pack_conv, pack_conv, pack_ln

_COUT = (32, 64, 128, 128)
_CIN = (12, 32, 64, 128)
_CINP = (16, 32, 64, 128)          # channels of the wgrad kernel's x operand (layer 1: 12 planes + 4 zero channels)
_RED_SLICES = 128               # sample slices of the banded path's LayerNorm gradient reduction (partials summed by torch)
_PG_SLICES = 64                 # sample slices of inv_policy_param_grads_all (each adds its sums with f32 atomics)
_WGRAD_CHUNK = 2048            # samples per im2col GEMM (bounds the 9x-expanded operand to ~0.7 GB)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _pack_ln(weight, bias, cout):
    """LayerNorm([C,10,15]) affine -> fp16 [150][C/4][weight x4 | bias x4] (see inversus_hip.h)."""
    w = weight.detach().float().reshape(cout, 150).t().reshape(150, cout // 4, 4)
    b = bias.detach().float().reshape(cout, 150).t().reshape(150, cout // 4, 4)
    return torch.cat([w, b], dim=2).contiguous().to(torch.float16)


def _wgrad_torch(x, dz, cin, cout):
    """dW[co][ci][ky][kx] = sum_n sum_p dz[n][p][co] * x[n][p + (ky-1, kx-1)][ci] as chunked im2col GEMMs in torch --
    the cross-check of the HIP kernel (tests), not the training path.  x: bf16 [N,150,>=cin], dz: bf16 [N,150,cout]."""
    x = x[..., :cin]
    n_all = x.shape[0]
    out = torch.zeros(cout, 9 * cin, dtype=torch.float32, device=x.device)
    for lo in range(0, n_all, _WGRAD_CHUNK):
        xs, ds = x[lo:lo + _WGRAD_CHUNK], dz[lo:lo + _WGRAD_CHUNK]
        n = xs.shape[0]
        xp = F.pad(xs.reshape(n, 10, 15, cin), (0, 0, 1, 1, 1, 1))                        # zero halo: [n,12,17,cin]
        cols = xp.unfold(1, 3, 1).unfold(2, 3, 1)                                         # [n,10,15,cin,ky,kx]
        cols = cols.permute(0, 1, 2, 4, 5, 3).reshape(n * 150, 9 * cin)                   # k = (ky, kx, ci)
        out += (ds.reshape(n * 150, cout).t() @ cols).float()
    return out.reshape(cout, 3, 3, cin).permute(0, 3, 1, 2).contiguous()


def _wgrad(layer, x, dz):
    """Weight and bias gradient of conv `layer` through inv_policy_wgrad (MFMA over pixels, transposing LDS reads).
    x: bf16 [N,150,CinP] contiguous layer input, dz: bf16 [N,150,Cout] -> (f32 [Cout, Cin, 3, 3], f32 [Cout])."""
    lib = L.load()
    cin, cinp, cout = _CIN[layer], _CINP[layer], _COUT[layer]
    assert x.dtype == dz.dtype == torch.bfloat16 and x.is_contiguous() and dz.is_contiguous()
    assert tuple(x.shape[1:]) == (150, cinp) and tuple(dz.shape[1:]) == (150, cout)
    dw = torch.zeros((cout, 9, cinp), dtype=torch.float32, device=x.device)
    db = torch.zeros((cout,), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.inv_policy_wgrad(layer, _ptr(x), _ptr(dz), _ptr(dw), _ptr(db), x.shape[0],
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
    L.raise_for(rc, lib)
    return dw[:, :, :cin].reshape(cout, 3, 3, cin).permute(0, 3, 1, 2).contiguous(), db


_SIDE_STREAMS = {}


def _side_stream(dev):
    """One extra stream per device for work that can run beside the main stream's kernels."""
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return _SIDE_STREAMS[key]


class _TrunkTrainFn(torch.autograd.Function):
    """features = trunk(obs; conv / LayerNorm parameters of the four layers); obs f32 planes or packed int32."""

    @staticmethod
    def forward(ctx, obs, *params):
        lib = L.load()
        dev = obs.device
        n = obs.shape[0]
        bits = obs.dtype == torch.int32
        obs = obs.contiguous() if bits else obs.contiguous().float()
        keep, fwd = [], L.InvPolicyParams()
        lns = []
        for i in range(4):
            w, b, g, e = params[4 * i:4 * i + 4]
            wp = pack_conv(w, _CIN_PAD[i])
            bp = b.detach().float().contiguous()
            ln = pack_ln(g, e, _COUT[i])
            keep += [wp, bp, ln]
            lns.append(ln)
            fwd.layer[i] = L.InvPolicyLayer(wp.data_ptr(), bp.data_ptr(), ln.data_ptr())
        xhat = [torch.empty((n, 150, c), dtype=torch.bfloat16, device=dev) for c in _COUT]
        act = [torch.empty((n, 150, c), dtype=torch.bfloat16, device=dev) for c in _COUT[:3]]
        rstd = torch.empty((n, 4), dtype=torch.float32, device=dev)
        feat = torch.empty((n, 19200), dtype=torch.bfloat16, device=dev)
        saved = L.InvPolicySaved()
        for i in range(4):
            saved.xhat[i] = xhat[i].data_ptr()
        for i in range(3):
            saved.act[i] = act[i].data_ptr()
        saved.rstd = rstd.data_ptr()
        with torch.cuda.device(dev):
            rc = lib.inv_policy_trunk_train_fwd(C.byref(fwd), _ptr(obs), 1 if bits else 0, _ptr(feat), C.byref(saved), n,
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream))
        L.raise_for(rc, lib)
        ctx.save_for_backward(obs, rstd, *xhat, *act, *lns, *[params[4 * i] for i in range(4)])
        ctx.bits = bits
        return feat

    @staticmethod
    def backward(ctx, dfeat, out=None, mid=None):
        """out: the 16 gradient tensors in parameter order (conv weight [Cout,Cin,3,3], conv bias, LayerNorm weight / bias [C,10,15]),
        ZEROED: the kernels add into them in the module's own layouts -- the direct path (loss_and_grads) passes the parameters'
        .grad slices of the flat bucket, autograd gets fresh tensors.  mid(): called on the main stream after the backward kernel is
        queued and the LayerNorm reduction has been put on the side stream -- the direct path runs the heads' first-Linear
        weight-gradient GEMM there (MFMA-bound, beside the HBM-bound reduction)."""
        lib = L.load()
        sv = ctx.saved_tensors
        obs, rstd = sv[0], sv[1]
        xhat, act, lns, ws = sv[2:6], sv[6:9], sv[9:13], sv[13:17]
        dev, n = obs.device, obs.shape[0]
        dfeat = dfeat.contiguous().to(torch.bfloat16)
        # no dy tensor for the top layer (ABI 8): the parameter-sum kernel recomputes it from dfeat; INVERSUS_TOP_DY=1 keeps the old path
        top_dy = os.environ.get("INVERSUS_TOP_DY", "0") == "1"
        dy = [torch.empty_like(t) if (i < 3 or top_dy) else None for i, t in enumerate(xhat)]
        dz = [torch.empty_like(t) for t in xhat]
        bp, keep = L.InvPolicyBwdParams(), []
        for i in range(1, 4):                       # Wt[ci][8 - tap][co] = W[co][ci][tap], packed as a conv with Cin = Cout_i
            pk = pack_conv(ws[i], _COUT[i], transposed=True)
            keep.append(pk)
            bp.wt[i - 1] = pk.data_ptr()
        for i in range(4):
            bp.ln[i] = lns[i].data_ptr()
        saved, grads = L.InvPolicySaved(), L.InvPolicyGrads()
        for i in range(4):
            saved.xhat[i] = xhat[i].data_ptr()
            grads.dy[i] = dy[i].data_ptr() if dy[i] is not None else None
            grads.dz[i] = dz[i].data_ptr()
        for i in range(3):
            saved.act[i] = act[i].data_ptr()
        saved.rstd = rstd.data_ptr()
        if out is None:
            out = []
            for i in range(4):
                c = _COUT[i]
                out += [torch.zeros((c, _CIN[i], 3, 3), dtype=torch.float32, device=dev), torch.zeros((c,), dtype=torch.float32, device=dev),
                        torch.zeros((c, 10, 15), dtype=torch.float32, device=dev), torch.zeros((c, 10, 15), dtype=torch.float32, device=dev)]
        else:
            out = list(out)
            if len(out) != 16 or any(t.dtype != torch.float32 or not t.is_contiguous() for t in out):
                raise ValueError("backward(out=...): 16 contiguous f32 tensors in parameter order")
        x0 = torch.empty((n, 150, 16), dtype=torch.bfloat16, device=dev)              # channels-last layer-1 input, 12 -> 16 channels (queued FIRST:
        # a 30-us kernel behind the side stream's reduction waits for that kernel's workgroups to drain -- 0.67 ms under rocprofv3)
        with torch.cuda.device(dev):
            rc = lib.inv_policy_obs_to_x0(_ptr(obs), 1 if ctx.bits else 0, _ptr(x0), n, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        L.raise_for(rc, lib)
        # The backward kernel runs in `parts` launches over consecutive sample ranges, and each range's LayerNorm reductions (HBM-bound:
        # dy and x^ once, 3.4 GB in all; ONE launch for the four layers, ADDED straight into the module-layout gradients) go to a side
        # stream as soon as its range is queued: the first range's reduction runs beside the second range's backward kernel, the last
        # one's beside `mid` (the first Linear's weight-gradient GEMM) and the convolutions' weight gradients, which only read what the
        # backward kernel wrote.  The main stream waits for the side stream before returning, so every tensor used there outlives its
        # use.  Measured (tools/bwd_parts_ab.py, profiles/r04_bwd_parts_ab.txt): 1, 2, 3 and 4 parts are all 6.6-6.7 ms per minibatch --
        # the board runs these kernels at its power cap and an HBM-bound kernel beside an MFMA-bound one hides little -- so the default
        # stays ONE launch.  INVERSUS_BWD_PARTS=k: k launches; INVERSUS_PG_STREAM=0: one stream.
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if os.environ.get("INVERSUS_PG_STREAM", "1") != "0" else main
        parts = max(1, int(os.environ.get("INVERSUS_BWD_PARTS", "1")))
        if side is main or n < 4096 * parts:
            parts = 1
        lng = L.InvPolicyLnGrads()
        for i in range(4):
            lng.gw[i], lng.gb[i] = out[4 * i + 2].data_ptr(), out[4 * i + 3].data_ptr()
        lo = 0
        for k in range(parts):
            m = (n - lo) // (parts - k)
            sv_k, gr_k = L.InvPolicySaved(), L.InvPolicyGrads()
            for i in range(4):
                off = lo * 150 * _COUT[i] * 2
                sv_k.xhat[i], gr_k.dz[i] = saved.xhat[i] + off, grads.dz[i] + off
                gr_k.dy[i] = grads.dy[i] + off if grads.dy[i] else None
            for i in range(3):
                sv_k.act[i] = saved.act[i] + lo * 150 * _COUT[i] * 2
            sv_k.rstd = saved.rstd + lo * 16
            with torch.cuda.device(dev):
                rc = lib.inv_policy_trunk_bwd(C.byref(bp), C.byref(sv_k), C.c_void_p(dfeat.data_ptr() + lo * 19200 * 2), C.byref(gr_k), m,
                                              C.c_void_p(main.cuda_stream))
            L.raise_for(rc, lib)
            if side is not main:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                with torch.cuda.device(dev):
                    rc = lib.inv_policy_param_grads_all2(C.byref(gr_k), C.byref(sv_k), C.byref(lng),
                                                         None if top_dy else C.c_void_p(dfeat.data_ptr() + lo * 19200 * 2),
                                                         None if top_dy else C.c_void_p(lns[3].data_ptr()),
                                                         _PG_SLICES if m >= 8192 else max(16, _PG_SLICES // 2), m,
                                                         C.c_void_p(torch.cuda.current_stream().cuda_stream))
                L.raise_for(rc, lib)
            lo += m
        if mid is not None:
            mid()
        # weight gradients: the kernel's [Cout][9][CinP] blocks of the four layers in one zeroed scratch, converted to the module's
        # [Cout][Cin][3][3] by one launch; the conv bias gradients are added straight into their tensors
        sizes = [_COUT[i] * 9 * _CINP[i] for i in range(4)]
        scratch = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        cg, off = L.InvPolicyConvGrads(), 0
        with torch.cuda.device(dev):
            for i in range(4):
                x_in = x0 if i == 0 else act[i - 1]
                rc = lib.inv_policy_wgrad(i, _ptr(x_in), _ptr(dz[i]), C.c_void_p(scratch.data_ptr() + 4 * off), _ptr(out[4 * i + 1]), n,
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream))
                L.raise_for(rc, lib)
                cg.dw[i] = out[4 * i].data_ptr()
                off += sizes[i]
            L.raise_for(lib.inv_policy_wgrad_to_module(_ptr(scratch), C.byref(cg), C.c_void_p(torch.cuda.current_stream().cuda_stream)), lib)
        if side is not main:
            main.wait_stream(side)
        return (None, *out)


class _TrunkTrainBandedFn(torch.autograd.Function):
    """features = trunk(obs; ...) for boards whose layers do not fit LDS (32 x 32, BASELINE config C5): the banded forward keeps
    the layers' pre-norm outputs z and the LayerNorm statistics; the backward runs layer by layer over the whole batch
    (csrc/inv_policy.hip: k_band_ln_bwd, k_band_conv_t, k_band_act, k_policy_wgrad<.., WgGeoBand>, k_band_param_grads).
    Reference: autograd over policies.py:91-99 at H = W = 32."""

    @staticmethod
    def forward(ctx, obs, *params):
        lib = L.load()
        dev, n = obs.device, obs.shape[0]
        bits = obs.dtype == torch.int32
        obs = obs.contiguous() if bits else obs.contiguous().float()
        cout0, hb, wb = (int(v) for v in params[2].shape)                          # LayerNorm([C, H, W]) weight of layer 1
        pix = hb * wb
        keep, fwd, lns = [], L.InvPolicyParams(), []
        for i in range(4):
            w, b, g, e = params[4 * i:4 * i + 4]
            wp = pack_conv(w, _CIN_PAD[i])
            bp = b.detach().float().contiguous()
            ln = pack_ln(g, e, _COUT[i])
            keep += [wp, bp, ln]
            lns.append(ln)
            fwd.layer[i] = L.InvPolicyLayer(wp.data_ptr(), bp.data_ptr(), ln.data_ptr())
        z = [torch.empty((n, pix, c), dtype=torch.bfloat16, device=dev) for c in _COUT]
        stats = torch.empty((n, 4, 2), dtype=torch.float32, device=dev)
        feat = torch.empty((n, pix * 128), dtype=torch.bfloat16, device=dev)
        saved = L.InvPolicyBandSaved()
        for i in range(4):
            saved.z[i] = z[i].data_ptr()
        saved.stats = stats.data_ptr()
        with torch.cuda.device(dev):
            rc = lib.inv_policy_trunk_banded_train_fwd(C.byref(fwd), _ptr(obs), 1 if bits else 0, hb, wb, _ptr(feat), C.byref(saved), n,
                                                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
        L.raise_for(rc, lib)
        ctx.save_for_backward(obs, stats, *z, *lns, *[params[4 * i] for i in range(4)])
        ctx.bits, ctx.hw = bits, (hb, wb)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        lib = L.load()
        sv = ctx.saved_tensors
        obs, stats = sv[0], sv[1]
        z, lns, ws = sv[2:6], sv[6:10], sv[10:14]
        dev, n = obs.device, obs.shape[0]
        hb, wb = ctx.hw
        pix = hb * wb
        zero_bias = torch.zeros(128, dtype=torch.float32, device=dev)
        da = dfeat.contiguous().to(torch.bfloat16)
        grads = [None] * 16

        def st():
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def wgrad(l, x, dz):
            c, cin, cinp = _COUT[l], _CIN[l], _CINP[l]
            dw = torch.zeros((c, 9, cinp), dtype=torch.float32, device=dev)
            db = torch.zeros((c,), dtype=torch.float32, device=dev)
            L.raise_for(lib.inv_policy_band_wgrad(l, hb, wb, _ptr(x), _ptr(dz), _ptr(dw), _ptr(db), n, st()), lib)
            grads[4 * l] = dw[:, :, :cin].reshape(c, 3, 3, cin).permute(0, 3, 1, 2).contiguous()
            grads[4 * l + 1] = db

        # l = 3 .. 0:  ln_sums(l) [da_l, z_l -> the sample's two sums; l < 3: left behind by conv_t(l + 1)'s write-back, ABI 7]
        #              ->  ln_bwd2(l) [da_l, z_l -> dz_l, a_l, LayerNorm parameter sums]
        #              ->  wgrad(l + 1) [a_l, dz_{l+1}]  ->  conv_t(l) [dz_l -> da_{l-1}, sums of layer l - 1];  finally wgrad(0) [x0, dz_0]
        fused_sums = os.environ.get("INVERSUS_BAND_FUSED_SUMS", "1") != "0"
        with torch.cuda.device(dev):
            dz_above, sums = None, None
            for l in (3, 2, 1, 0):
                c = _COUT[l]
                dz = torch.empty_like(z[l])
                act = torch.empty_like(z[l]) if l < 3 else None
                # two tensor passes over (da, z): the sample's two sums, then dz + activation + LayerNorm parameter partial sums in one
                if sums is None:
                    sums = torch.empty((n, 2), dtype=torch.float32, device=dev)
                    L.raise_for(lib.inv_policy_band_ln_sums(l, hb, wb, _ptr(da), _ptr(z[l]), _ptr(stats), _ptr(lns[l]), _ptr(sums), n, st()), lib)
                slices = max(1, min(n, _RED_SLICES))
                part = torch.empty((slices, 2, pix, c), dtype=torch.float32, device=dev)
                L.raise_for(lib.inv_policy_band_ln_bwd2(l, hb, wb, _ptr(da), _ptr(z[l]), _ptr(stats), _ptr(lns[l]), _ptr(sums), _ptr(dz),
                                                        _ptr(act) if act is not None else None, _ptr(part), slices, n, st()), lib)
                if l < 3:
                    wgrad(l + 1, act, dz_above)
                    del act, dz_above
                red = part.sum(dim=0)
                grads[4 * l + 2], grads[4 * l + 3] = red[0].t().reshape(c, hb, wb), red[1].t().reshape(c, hb, wb)
                del part
                if l >= 1:
                    wt = pack_conv(ws[l], _COUT[l], transposed=True)
                    da = torch.empty((n, pix, _COUT[l - 1]), dtype=torch.bfloat16, device=dev)
                    if fused_sums:
                        sums = torch.empty((n, 2), dtype=torch.float32, device=dev)
                        L.raise_for(lib.inv_policy_band_conv_t_sums(l, hb, wb, _ptr(wt), _ptr(zero_bias), _ptr(dz), _ptr(da), _ptr(z[l - 1]), _ptr(stats),
                                                                    _ptr(lns[l - 1]), _ptr(sums), n, st()), lib)
                    else:
                        sums = None
                        L.raise_for(lib.inv_policy_band_conv_t(l, hb, wb, _ptr(wt), _ptr(zero_bias), _ptr(dz), _ptr(da), n, st()), lib)
                dz_above = dz
            x0 = torch.empty((n, pix, 16), dtype=torch.bfloat16, device=dev)
            L.raise_for(lib.inv_policy_obs_to_x0_any(_ptr(obs), 1 if ctx.bits else 0, pix, _ptr(x0), n, st()), lib)
            wgrad(0, x0, dz_above)
        return (None, *grads)


class _DirectCtx:
    """Stand-in for the autograd context when a Function's forward / backward are called directly (loss_and_grads)."""
    needs_input_grad = (True,) * 32
    saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass


def _trunk_params(p):
    out = []
    for conv, norm in ((p.conv1, p.norm1), (p.conv2, p.norm2), (p.conv3, p.norm3), (p.conv4, p.norm4)):
        out += [conv.weight, conv.bias, norm.weight, norm.bias]
    return out


class FusedTrainPolicy:
    """Differentiable stand-in for `policy(grid, extra)` in the PPO update: fused trunk forward / backward kernels,
    torch heads.  `grid`: f32 planes [B,12,10,15] or the packed int32 observations [B,57]."""

    def __init__(self, policy):
        p = policy
        shape = tuple(p.norm1.normalized_shape)
        if shape not in ((32, 10, 15), (32, 32, 32)) or p.conv1.in_channels != 12:
            raise ValueError("the fused trunk trains 12-channel observations of the 15 x 10 and 32 x 32 boards only")
        if p.conv1.weight.device.type != "cuda":
            raise RuntimeError("FusedTrainPolicy needs the policy on the GPU")
        self.policy = policy
        # 15 x 10: one fused forward kernel + one fused backward kernel, activations in LDS; 32 x 32: the banded forward and a
        # layer-by-layer backward over the whole batch
        self.trunk_fn = _TrunkTrainFn if shape == (32, 10, 15) else _TrunkTrainBandedFn
        L.load()

    def features(self, grid):
        return self.trunk_fn.apply(grid, *_trunk_params(self.policy))

    @torch.no_grad()
    def loss_and_grads(self, grid, extra, actions, old_log_probs, advantages, returns, clip_ratio, value_coef, entropy_coef,
                       after_heads=None):
        """One minibatch of the PPO loss (ppo_agent.py:206-225) forward AND backward without the autograd engine: the same kernels
        in the same order as `loss.backward()` over __call__ + flat_optim.ppo_loss, but every parameter's gradient is COPIED into
        its existing .grad (the slices of the flat gradient bucket) by two multi-tensor copies instead of 28 accumulate kernels
        over a zeroed bucket.  after_heads(): called once the heads' gradients are in place (the reducer starts their all-reduce
        while the trunk's backward runs).  -> means f32 [3] = policy loss, value loss, entropy."""
        from .flat_optim import _PPOLossFn
        p = self.policy
        if not HD.supported(p):
            raise ValueError("loss_and_grads needs the reference's head sizes (hidden_dim = 256)")
        tparams, hparams = _trunk_params(p), [t for h in HD.head_tensors(p) for t in h]
        if any(q.grad is None for q in tparams + hparams):
            raise RuntimeError("loss_and_grads writes into existing .grad tensors (FlatAdam / FlatGradReducer create them)")
        tctx, hctx, lctx = _DirectCtx(), _DirectCtx(), _DirectCtx()
        feat = self.trunk_fn.forward(tctx, grid, *tparams)
        logits, value = HD._HeadsTrainFn.forward(hctx, feat, extra, *hparams)
        if actions.dtype != torch.int64:
            actions = actions.to(torch.int64)
        _, means = _PPOLossFn.forward(lctx, logits, value.squeeze(-1), actions, old_log_probs, advantages, returns,
                                      clip_ratio, value_coef, entropy_coef)
        dlogits, dvalues = lctx.saved_tensors                             # d loss / d logits, d values (loss.backward(): g_loss = 1)
        w1 = (hparams[0], hparams[6])                                     # 39.3 of the 41 MB: written in place by the unpack kernel
        if all(q.grad.is_contiguous() for q in w1):
            hctx.fc1_grad_out = [q.grad for q in w1]
        if self.trunk_fn is _TrunkTrainFn:
            # 15 x 10: the heads' backward in two halves around the trunk's backward kernel -- d features first (the trunk needs it),
            # then, once that kernel and the LayerNorm reduction (side stream, HBM-bound) are queued, the first Linear's weight-
            # gradient GEMM (MFMA-bound) and the small head gradients; the trunk's gradients are ADDED into the (zeroed) .grad slices
            dfeat = HD._HeadsTrainFn.backward_features(hctx, dlogits, dvalues.reshape(-1, 1))
            tgrads = [q.grad for q in tparams]
            torch._foreach_zero_(tgrads)

            def mid():
                hg = HD._HeadsTrainFn.backward_params(hctx)
                pairs = [(q.grad, g) for q, g in zip(hparams, hg) if g is not q.grad]
                torch._foreach_copy_([a for a, _ in pairs], [b for _, b in pairs])
                if after_heads is not None:
                    after_heads()
            # Default: all of the heads' backward first (the round-2 order: their gradients -- 39 of the 41 MB -- are complete before
            # the trunk's backward starts, so a multi-GPU run's early all-reduce travels under it).  INVERSUS_FC1_MID=1 moves the
            # first Linear's weight-gradient GEMM behind the trunk's backward kernel, beside the LayerNorm reduction: measured equal
            # (6.59 vs 6.57-6.61 ms, tools/update_ab.py), and a library stream-K GEMM next to another stream's kernels is one
            # concurrency pattern less to depend on.
            if os.environ.get("INVERSUS_FC1_MID", "0") != "1":
                mid()
                mid = None
            self.trunk_fn.backward(tctx, dfeat, out=tgrads, mid=mid)
            return means
        hg = HD._HeadsTrainFn.backward(hctx, dlogits, dvalues.reshape(-1, 1))
        pairs = [(q.grad, g) for q, g in zip(hparams, hg[2:]) if g is not q.grad]
        torch._foreach_copy_([a for a, _ in pairs], [b for _, b in pairs])
        if after_heads is not None:
            after_heads()
        tg = self.trunk_fn.backward(tctx, hg[0])
        torch._foreach_copy_([q.grad for q in tparams], list(tg[1:]))
        return means

    def __call__(self, grid, extra):
        p = self.policy
        feat = self.features(grid)                                                     # bf16 [B, 19200], index pixel*128 + c
        if HD.supported(p):
            return HD.heads_train(p, feat, extra)                                      # fused heads (csrc/inv_heads.hip)
        extra = extra.float()
        # first Linear of both heads as ONE [512 x 128 P] bf16 GEMM on the kernel's feature order (columns permuted from
        # the module's c*P + pixel to pixel*128 + c); autograd splits the gradient back onto the two weight tensors
        w = torch.cat([p.fc_actor[0].weight, p.fc_critic[0].weight])                    # [512, 128 P + 4]
        b = torch.cat([p.fc_actor[0].bias, p.fc_critic[0].bias])
        hid = p.fc_actor[0].out_features
        nf = feat.shape[1]
        wf = w[:, :nf].reshape(-1, 128, nf // 128).permute(0, 2, 1).reshape(-1, nf).to(torch.bfloat16)
        h = torch.relu((feat @ wf.t()).float() + extra @ w[:, nf:].t() + b)
        logits = p.fc_actor[4](torch.relu(p.fc_actor[2](h[:, :hid])))
        value = p.fc_critic[4](torch.relu(p.fc_critic[2](h[:, hid:])))
        return logits, value
