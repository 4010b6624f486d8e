"""Host side of csrc/inv_fc1.hip + csrc/inv_heads.hip: the two heads of InversusCNNPolicy (inversus_rl/policies.py:60-75, :102-106)
after the trunk.

    pre    = inv_fc1_fwd(features, W1[:, :128 P])            [B x 128 P] x [128 P x 512] bf16 MFMA kernel, f32 split-K slabs
    logits, value = inv_heads_fwd(pre, extra)                one kernel (first-layer epilogue, Linear 256 -> 128, Linear 128 -> 13 | 1)

The backward's two big products (d features, d W1) are inv_fc1_dfeat / inv_fc1_dw, d W2 is inv_heads_dw2: no GEMM-library call.

`HeadsOperands` packs a policy's head parameters for the kernels (rollouts: once per refresh); `heads_train` is the differentiable
form the fused PPO update uses (packs per call -- the parameters change every minibatch -- and returns gradients for all 12 head
tensors and for the features).  Only the reference's head sizes (hidden_dim = 256) have kernels; callers keep their torch tail for
anything else.

Batch size and summation order: inv_fc1_fwd splits K into inv_fc1_fwd_slabs(B, nout, K) slabs (a function of B and the CU count: 2 at
B = 16 384, up to 8 for small minibatches on 256 CUs), summed in slab order -- so a sample's `pre` is bit-identical between launches of
ONE batch size and equal to f32 reassociation between DIFFERENT ones: |d logits| < 2e-3 max, < 2e-5 mean at 16 384 vs 512 rows
(tests/test_gpu_heads.py).  PPO's epoch-0 ratio is therefore exactly 1 when minibatch = rollout batch (the 16 384-env configurations)
and 1 +- 2e-3 otherwise; results on GPUs with another CU count differ by the same reassociation."""
import ctypes as C

import torch

from . import _lib as L

HID, MID, NACT = 256, 128, 13


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def supported(policy):
    """The reference's head sizes AND an even pixel count: inv_fc1_dfeat / inv_fc1_dw tile the 128 P feature columns in 256s."""
    a, c = policy.fc_actor, policy.fc_critic
    return (a[0].out_features == HID and a[2].out_features == MID and a[4].out_features == NACT and c[0].out_features == HID
            and c[2].out_features == MID and c[4].out_features == 1 and (a[0].in_features - 4) % 256 == 0)


def _pack_fc1(weights, pixels, out=None):
    """[256, 128 P + 4] f32 per head -> (wf bf16 [256 nheads, 128 P] in the trunk's feature order, we f32 [256 nheads, 4]).
    out = (wf, we): re-pack into existing buffers (same device pointers: a captured HIP graph that reads them stays valid)."""
    lib = L.load()
    dev = weights[0].device
    n = len(weights)
    if out is None:
        wf = torch.empty((HID * n, 128 * pixels), dtype=torch.bfloat16, device=dev)
        we = torch.empty((HID * n, 4), dtype=torch.float32, device=dev)
    else:
        wf, we = out
    with torch.cuda.device(dev):
        for i, w in enumerate(weights):
            w = w.detach()
            if w.dtype != torch.float32 or not w.is_contiguous() or tuple(w.shape) != (HID, 128 * pixels + 4):
                raise ValueError(f"first Linear of a head must be contiguous f32 [{HID}, {128 * pixels + 4}], got {tuple(w.shape)} {w.dtype}")
            L.raise_for(lib.inv_heads_pack_fc1(_ptr(w), HID, pixels, _ptr(wf[HID * i]), _ptr(we[HID * i]), _stream()), lib)
    return wf, we


def _pack_fragments(mats, need_t, out=None):
    """mats: per head (W2 [128, 256], W3 [13 | 1, 128]) f32 -> fragment tensors {w2, w3, w2t, w3t} per head, one launch.
    out: the list a previous call returned -- the fragments are re-packed into those tensors."""
    lib = L.load()
    dev = mats[0][0].device
    lst, res, k = L.InvHeadsPackList(), [], 0
    for hi, (w2, w3) in enumerate(mats):
        w2, w3 = w2.detach(), w3.detach()
        for w, shape in ((w2, (MID, HID)), (w3, (w3.shape[0], MID))):
            if w.dtype != torch.float32 or not w.is_contiguous() or tuple(w.shape) != shape:
                raise ValueError(f"head Linear must be contiguous f32 {shape}, got {tuple(w.shape)} {w.dtype}")
        if out is not None:
            d = out[hi]
        else:
            d = {"w2": torch.empty(MID * HID, dtype=torch.bfloat16, device=dev), "w3": torch.empty(16 * MID, dtype=torch.bfloat16, device=dev)}
            if need_t:
                d["w2t"] = torch.empty(HID * MID, dtype=torch.bfloat16, device=dev)
                d["w3t"] = torch.empty(MID * 32, dtype=torch.bfloat16, device=dev)
        items = [(w2, d["w2"], MID, HID, HID, 0, MID, HID), (w3, d["w3"], w3.shape[0], MID, MID, 0, 16, MID)]
        if need_t:
            items += [(w2, d["w2t"], HID, MID, HID, 1, HID, MID), (w3, d["w3t"], MID, w3.shape[0], MID, 1, MID, 32)]
        for src, dst, rows, cols, ld, tr, mpad, kpad in items:
            lst.item[k] = L.InvHeadsPackItem(src.data_ptr(), dst.data_ptr(), rows, cols, ld, tr, mpad, kpad)
            k += 1
        res.append(d)
    with torch.cuda.device(dev):
        L.raise_for(lib.inv_heads_pack(C.byref(lst), k, _stream()), lib)
    return res


class HeadsOperands:
    """Packed head parameters of `policy` for inv_heads_fwd / inv_heads_bwd.  nheads = 1: actor only."""

    def __init__(self, heads, pixels, need_t=False):
        """heads: per head the six tensors (W1, b1, W2, b2, W3, b3) of its nn.Sequential."""
        self.nheads, self.pixels, self.need_t = len(heads), int(pixels), bool(need_t)
        self.wf = self.we = self.b1 = self.frags = None
        self.repack(heads)

    def repack(self, heads):
        """(Re-)pack the parameters.  After the first call everything is written IN PLACE -- wf / we / b1 / the fragment tensors keep
        their device pointers and b2 / b3 are read from the live parameters -- so kernels captured in a HIP graph
        (RolloutCollector(graph=True)) see the new weights without a re-capture."""
        if len(heads) != self.nheads:
            raise ValueError(f"repack: {self.nheads} heads were packed, got {len(heads)}")
        first = self.wf is None
        self.wf, self.we = _pack_fc1([h[0] for h in heads], self.pixels, None if first else (self.wf, self.we))
        b1 = torch.cat([h[1].detach().float() for h in heads]).contiguous()
        if first:
            self.b1 = b1
        else:
            self.b1.copy_(b1)
        self.frags = _pack_fragments([(h[2], h[4]) for h in heads], self.need_t, None if first else self.frags)
        small = [[h[3].detach().float().contiguous(), h[5].detach().float().contiguous()] for h in heads]
        if first:
            self.keep = [self.frags] + small
            p = L.InvHeadsParams()
            for i, f in enumerate(self.frags):
                p.w2[i], p.w3[i] = f["w2"].data_ptr(), f["w3"].data_ptr()
                if self.need_t:
                    p.w2t[i], p.w3t[i] = f["w2t"].data_ptr(), f["w3t"].data_ptr()
                p.b2[i], p.b3[i] = small[i][0].data_ptr(), small[i][1].data_ptr()
            p.we, p.b1 = self.we.data_ptr(), self.b1.data_ptr()
            self.params = p
        else:
            for i, (b2, b3) in enumerate(small):                                # views of the live parameters, or (non-f32 / strided) copies
                if b2.data_ptr() != self.keep[1 + i][0].data_ptr():
                    self.keep[1 + i][0].copy_(b2)
                if b3.data_ptr() != self.keep[1 + i][1].data_ptr():
                    self.keep[1 + i][1].copy_(b3)

    def forward(self, feat, extra, save=False):
        """feat bf16 [B, 128 P], extra f32 [B, 4] -> logits f32 [B, 13], value f32 [B, 1] | None (+ (pre... h, a2) when save)."""
        lib = L.load()
        B, n = feat.shape[0], self.nheads
        if feat.dtype != torch.bfloat16 or feat.dim() != 2 or feat.shape[1] != self.wf.shape[1] or not feat.is_contiguous():
            raise ValueError(f"features must be contiguous bf16 [B, {self.wf.shape[1]}], got {tuple(feat.shape)} {feat.dtype}")
        K = feat.shape[1]
        with torch.cuda.device(feat.device):
            slabs = int(lib.inv_fc1_fwd_slabs(B, HID * n, K)) if B > 0 else 1
            if slabs < 1:
                raise ValueError(f"inv_fc1_fwd does not support B = {B}, nout = {HID * n}, K = {K}")
            pre = torch.empty((slabs, B, HID * n), dtype=torch.float32, device=feat.device)   # f32 split-K partial sums
            L.raise_for(lib.inv_fc1_fwd(_ptr(feat), _ptr(self.wf), B, HID * n, K, _ptr(pre), _stream()), lib)
        extra = extra.float().contiguous()
        logits = torch.empty((B, NACT), dtype=torch.float32, device=feat.device)
        value = torch.empty((B, 1), dtype=torch.float32, device=feat.device) if n == 2 else None
        h = torch.empty((B, HID * n), dtype=torch.bfloat16, device=feat.device) if save else None
        a2 = torch.empty((B, MID * n), dtype=torch.bfloat16, device=feat.device) if save else None
        with torch.cuda.device(feat.device):
            rc = lib.inv_heads_fwd(C.byref(self.params), _ptr(pre), slabs, _ptr(extra), _ptr(logits), _ptr(value) if n == 2 else None,
                                   _ptr(h) if save else None, _ptr(a2) if save else None, n, B, _stream())
        L.raise_for(rc, lib)
        return (logits, value, extra, h, a2) if save else (logits, value)


def head_tensors(policy, actor_only=False):
    heads = [policy.fc_actor] if actor_only else [policy.fc_actor, policy.fc_critic]
    return [(s[0].weight, s[0].bias, s[2].weight, s[2].bias, s[4].weight, s[4].bias) for s in heads]


class _SmallGrads:
    """The small gradients inv_heads_bwd ADDS into: one zeroed f32 buffer with named slices
    [db1 512 | dwe 512 x 4 | db2 128, 128 | dw3 13 x 128, 1 x 128 | db3 13, 1]."""
    SIZES = (("db1", 2 * HID), ("dwe", 2 * HID * 4), ("db2_0", MID), ("db2_1", MID), ("dw3_0", NACT * MID), ("dw3_1", MID),
             ("db3_0", NACT), ("db3_1", 1))

    def __init__(self, device):
        self.off, n = {}, 0
        for name, size in self.SIZES:
            self.off[name] = (n, size)
            n += size
        self.buf = torch.zeros(n, dtype=torch.float32, device=device)

    def view(self, name):
        o, n = self.off[name]
        return self.buf[o:o + n]

    def ptr(self, name, elem=0):
        return self.buf.data_ptr() + 4 * (self.off[name][0] + elem)

    def struct(self):
        gr = L.InvHeadsGrads()
        gr.db1, gr.dwe = self.ptr("db1"), self.ptr("dwe")
        for i in range(2):
            gr.db2[i], gr.dw3[i], gr.db3[i] = self.ptr(f"db2_{i}"), self.ptr(f"dw3_{i}"), self.ptr(f"db3_{i}")
        return gr


class _HeadsTrainFn(torch.autograd.Function):
    """(features bf16, extra, 6 actor tensors, 6 critic tensors) -> (logits, value); gradients for everything but `extra`."""

    @staticmethod
    def forward(ctx, feat, extra, *tensors):
        pixels = feat.shape[1] // 128
        if feat.shape[1] != 128 * pixels or pixels % 2:
            raise ValueError(f"heads_train: the backward kernels (inv_fc1_dfeat / inv_fc1_dw) need an even pixel count, got "
                             f"{feat.shape[1]} features = 128 x {feat.shape[1] / 128:g} pixels")
        ops = HeadsOperands([tensors[:6], tensors[6:]], pixels, need_t=True)
        logits, value, extra_c, h, a2 = ops.forward(feat, extra, save=True)
        ctx.ops = ops
        ctx.save_for_backward(feat, extra_c, h, a2)
        ctx.pixels = pixels
        return logits, value

    @staticmethod
    def backward(ctx, dlogits, dvalue):
        dfeat = _HeadsTrainFn.backward_features(ctx, dlogits, dvalue)
        return (dfeat, None, *_HeadsTrainFn.backward_params(ctx))

    @staticmethod
    def backward_features(ctx, dlogits, dvalue):
        """First half: inv_heads_bwd (everything behind the first Linear, the small gradients added per block) and
        d features = d pre @ W1 -- what the trunk's backward waits for.  Leaves what the second half needs on ctx."""
        lib = L.load()
        feat, extra, h, a2 = ctx.saved_tensors
        ops, P, B, dev = ctx.ops, ctx.pixels, feat.shape[0], feat.device
        dlogits = dlogits.float().contiguous() if dlogits is not None else torch.zeros((B, NACT), dtype=torch.float32, device=dev)
        dvalue = dvalue.float().contiguous() if dvalue is not None else torch.zeros((B, 1), dtype=torch.float32, device=dev)
        small = _SmallGrads(dev)
        gr = small.struct()
        dpre = torch.empty((B, 2 * HID), dtype=torch.bfloat16, device=dev)
        # d pre^T: the K-contiguous operand of the weight-gradient kernel, rows padded to a multiple of 64 samples with zeros
        ldt = (B + 63) // 64 * 64
        dpre_t = (torch.empty if ldt == B else torch.zeros)((2 * HID, ldt), dtype=torch.bfloat16, device=dev)
        da2 = torch.empty((B, 2 * MID), dtype=torch.bfloat16, device=dev)
        with torch.cuda.device(dev):
            rc = lib.inv_heads_bwd(C.byref(ops.params), _ptr(dlogits), _ptr(dvalue), _ptr(h), _ptr(a2), _ptr(extra), _ptr(dpre),
                                   _ptr(dpre_t), ldt, _ptr(da2), C.byref(gr), B, _stream())
        L.raise_for(rc, lib)
        ctx.mid_state = (small, dpre, dpre_t, da2)
        needs = getattr(ctx, "needs_input_grad", (True,))
        if not needs[0]:
            return None
        # both operands of inv_fc1_dfeat are K-contiguous: d pre [B, 512] and the transposed copy of the packed weights [128 P, 512]
        wft = torch.empty((128 * P, 2 * HID), dtype=torch.bfloat16, device=dev)
        dfeat = torch.empty((B, 128 * P), dtype=torch.bfloat16, device=dev)
        with torch.cuda.device(dev):
            L.raise_for(lib.inv_transpose_bf16(_ptr(ops.wf), 2 * HID, 128 * P, _ptr(wft), _stream()), lib)
            L.raise_for(lib.inv_fc1_dfeat(_ptr(dpre), _ptr(wft), B, 2 * HID, 128 * P, _ptr(dfeat), _stream()), lib)
        return dfeat                                                                   # bf16 [B, 128 P]

    @staticmethod
    def backward_params(ctx):
        """Second half: the first Linear's weight gradient d pre^T @ features (the big GEMM), its unpacking into the module's
        column order, dW2 and the small gradients -> the 12 head tensors' gradients in head_tensors() order."""
        lib = L.load()
        feat, extra, h, a2 = ctx.saved_tensors
        P, B, dev = ctx.pixels, feat.shape[0], feat.device
        small, dpre, dpre_t, da2 = ctx.mid_state                                        # (kept: lives as long as the context does)
        dst = getattr(ctx, "fc1_grad_out", None)                                      # direct path: the parameters' .grad themselves
        gw1 = list(dst) if dst is not None else [torch.empty((HID, 128 * P + 4), dtype=torch.float32, device=dev) for _ in range(2)]
        gw2 = [torch.zeros((MID, HID), dtype=torch.float32, device=dev) for _ in range(2)]
        with torch.cuda.device(dev):
            # dW1 = d pre^T @ features (stream-K, f32 segment sums in `ws`), written in the module's column order with the extra columns
            ws = torch.empty(int(lib.inv_fc1_dw_workspace_bytes(B, 2 * HID, 128 * P)), dtype=torch.uint8, device=dev)
            L.raise_for(lib.inv_fc1_dw(_ptr(dpre_t), _ptr(feat), C.c_void_p(small.ptr("dwe")), B, 2 * HID, P, _ptr(gw1[0]), _ptr(gw1[1]),
                                       _ptr(ws), _stream()), lib)
            # dW2 of both heads: d a2^T @ h, added into the zeroed buffers
            L.raise_for(lib.inv_heads_dw2(_ptr(da2), _ptr(h), B, _ptr(gw2[0]), _ptr(gw2[1]), _stream()), lib)
        out = []
        for i in range(2):                                                             # order of head_tensors(): W1, b1, W2, b2, W3, b3
            out += [gw1[i], small.view("db1")[HID * i:HID * (i + 1)], gw2[i], small.view(f"db2_{i}"),
                    small.view(f"dw3_{i}").view(-1, MID), small.view(f"db3_{i}")]
        return out


def heads_train(policy, feat, extra):
    """Differentiable heads on fused kernels: (bf16 features [B, 128 P], extra [B, 4]) -> (logits [B, 13], value [B, 1])."""
    tensors = [t for head in head_tensors(policy) for t in head]
    return _HeadsTrainFn.apply(feat, extra, *tensors)
