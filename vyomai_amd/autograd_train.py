"""torch.autograd.Function wrappers whose forward AND backward are HIP kernel launches.

PyTorch supplies the tape only.  Parameters stay fp32 (master weights); activations and the
weights the kernels read are bf16 copies (``_shadow``), refreshed by the fused AdamW kernel when
the FlatTrainer owns the parameters.

Gradient delivery has two modes per parameter:
  * direct   -- ``param.grad`` is a preallocated fp32 view into the trainer's flat gradient arena
               (``param._vy_direct``): wgrad kernels accumulate straight into it (no temporaries,
               contiguous buckets for the RCCL reducer) and autograd receives ``None``;
  * returned -- otherwise a fresh fp32 gradient is computed and handed back to autograd.
"""
from __future__ import annotations

import os

import weakref
from typing import Optional

import torch

from . import _lib, ops
from ._lib import ACT_GELU_ERF, ACT_NONE, VyomHipError
from .layers.attention import _shadow
from .layers.mask import AttnMask
from .layers.positional_embeddings import resolve_freqs

BF16 = torch.bfloat16


# bumped by FlatTrainer.step(): its AdamW kernel rewrites weights without touching tensor versions
WEIGHT_EPOCH = [0]


class _WtRegistry:
    """Every W^T the dgrad GEMMs read.  The first backward transposes each matrix as it meets it; from
    then on ONE vy_transpose_batched launch per weight epoch refreshes all that were used in the
    previous epoch (the per-matrix launches are latency-bound: ~8 us each, 50 per step).  Owners are
    held weakly: a model that is dropped, or no longer trained, falls out of the batch."""

    def __init__(self) -> None:
        self.entries = []      # [weakref(owner), key_fn, src_fn, dst_view, last_used_epoch]
        self.batches = {}
        self.epoch_done = -1

    def register(self, owner, key_fn, src_fn, dst):
        e = [weakref.ref(owner), key_fn, src_fn, dst, WEIGHT_EPOCH[0]]
        self.entries.append(e)
        return e

    def refresh(self) -> None:
        """Re-transpose the recently used matrices for the current WEIGHT_EPOCH (once per epoch)."""
        ep = WEIGHT_EPOCH[0]
        if self.epoch_done == ep:
            return
        self.epoch_done = ep
        self.entries = [e for e in self.entries if e[0]() is not None]
        live = [e for e in self.entries if e[4] >= ep - 1 and getattr(e[0](), "_vy_wt_key", None) != e[1](e[0]())]
        if not live:
            return
        pairs = [(e[2](e[0]()), e[3]) for e in live]
        # one launch per dtype (a bf16 model and an fp32 one trained in the same process share this registry)
        for dt in sorted({s_.dtype for s_, _ in pairs}, key=str):
            sub = [(s_, d_) for s_, d_ in pairs if s_.dtype == dt]
            keys = [(s_.data_ptr(), d_.data_ptr()) for s_, d_ in sub]
            batch = self.batches.get(dt)
            if batch is None or batch.keys != keys:
                batch = self.batches[dt] = ops.TransposeBatch(sub)
            batch.run()
        for e in live:
            e[0]()._vy_wt_key = e[1](e[0]())


_WT = _WtRegistry()


def _wt_cached(owner, key_fn, src_fn) -> torch.Tensor:
    """Shared body of _wt / _wt_packed: owner carries _vy_wt_key / _vy_wt_view / _vy_wt_entry.
    key_fn / src_fn take the owner as their argument (the registry must not keep it alive)."""
    view = getattr(owner, "_vy_wt_view", None)
    key = key_fn(owner)
    entry = getattr(owner, "_vy_wt_entry", None)
    if entry is not None:
        entry[4] = WEIGHT_EPOCH[0]
    if view is not None and getattr(owner, "_vy_wt_key", None) == key:
        return view
    src = src_fn(owner)
    if view is not None and view.dtype == src.dtype and view.device == src.device:
        _WT.refresh()   # weights changed: all recently used matrices in one launch
        if owner._vy_wt_key != key:
            ops.transpose(src, view)   # not in the batch (first use after a pause, or an in-place edit)
            owner._vy_wt_key = key
        return view
    N, K = src.shape
    ld = _row_stride(N)
    view = torch.zeros((K, ld), dtype=src.dtype, device=src.device)[:, :N]  # pad columns stay zero
    ops.transpose(src, view)
    owner._vy_wt_view, owner._vy_wt_key = view, key
    owner._vy_wt_entry = _WT.register(owner, key_fn, src_fn, view)
    return view


_ROW_GRANULE = int(os.environ.get("VY_ROW_GRANULE", "64"))   # (8: the layout of rounds 1-2, for A/B runs)


def _row_stride(n: int) -> int:
    """Row stride (elements) of a matrix with n columns that a GEMM reads as an operand: whole 128-byte lines per row
    (64 bf16), so that no row starts in the middle of a line.  With the 16-byte granule of rounds 1-2 the 50265-wide logits had a
    100,544-byte stride: every second row of the dgrad / wgrad operand began 64 bytes into a line and each 128-byte piece an
    LDS-DMA instruction fetches came from two lines."""
    g = _ROW_GRANULE
    return (n + g - 1) // g * g


def _wt(param: torch.Tensor, dtype) -> torch.Tensor:
    """W^T ([K, N], row stride padded to 8) of a 2-D parameter in `dtype`, cached per version."""
    return _wt_cached(param, lambda p: (p._version, WEIGHT_EPOCH[0], dtype), lambda p: _shadow(p, dtype))


def _direct(p: Optional[torch.Tensor]) -> bool:
    return p is not None and getattr(p, "_vy_direct", False) and p.grad is not None


def _notify(*params) -> None:
    for p in params:
        if p is not None:
            cb = getattr(p, "_vy_ready", None)
            if cb is not None:
                cb(p)


class _WgradGroup:
    """Weight gradients that accumulate straight into the gradient arena do not have to be launched where
    autograd reaches them: the projections of a layer are small wgrad problems (9-36 tiles of 256 x 256),
    and one at a time each needs M-splits -- fp32 atomic traffic -- to fill the chip.  They are collected
    here and go out as ONE launch per ~layer (vy_linear_wgrad_grouped); the parameters are reported ready
    (DDP buckets, per-bucket AdamW) when that launch has been enqueued.  Whatever is still pending when
    the backward pass ends is flushed by an autograd-engine callback."""

    TILES = int(os.environ.get("VY_WGRAD_GROUP_TILES", "100"))   # flush once this many 256 x 256 output tiles are pending (one post-LN layer: 108)

    def __init__(self):
        self.items, self.tiles, self.armed = [], 0, False

    def wants(self, dy, w, alpha) -> bool:
        if alpha is not None or not _GROUP_WGRADS:
            return False
        N, K = w.shape
        return dy.numel() // N >= _GROUP_MIN_ROWS and N < 8192 and N * K >= 512 * 512 and K % 8 == 0

    def add(self, dy, x, w, b) -> None:
        self.add_tensors(dy, x, w.grad, None if b is None else b.grad, [p for p in (w, b) if p is not None])

    def add_tensors(self, dy, x, dw, db, members) -> None:
        """dw / db: the fp32 gradient tensors the launch accumulates into (views of the arena); members: the
        parameters to report ready once it has been enqueued (a packed projection has several)."""
        self.items.append((dy, x, dw, db, members))
        # autograd's own post-accumulate hook fires for these parameters as soon as this backward function
        # returns -- the reducer must not take that for "gradient written" (training.BucketReducer._hook)
        for p in members:
            p._vy_deferred = True
        self.tiles += -(-dw.shape[0] // 256) * -(-dw.shape[1] // 256)
        if not self.armed:
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
            self.armed = True
        if self.tiles >= self.TILES or len(self.items) == 8:
            self.flush()

    def flush(self) -> None:
        items, self.items, self.tiles = self.items, [], 0
        if not items:
            return
        ops.linear_wgrad_grouped([(dy, x, dw, db) for dy, x, dw, db, _ in items])
        for *_, members in items:
            for p in members:
                p._vy_deferred = False
            _notify(*members)

    def _end_of_backward(self) -> None:
        self.armed = False
        self.flush()

    def discard(self) -> None:
        for *_, members in self.items:
            for p in members:
                p._vy_deferred = False
        self.items, self.tiles, self.armed = [], 0, False


_GROUP_WGRADS = os.environ.get("VY_WGRAD_GROUP", "1") != "0"
_DEFER_RESIDUALS = os.environ.get("VY_DEFER_RESIDUALS", "1") != "0"
_GROUP_MIN_ROWS = int(os.environ.get("VY_WGRAD_GROUP_ROWS", "2048"))   # measured on configs[3] (2112 decoder rows): -6 %
_GROUP_QKV = os.environ.get("VY_WGRAD_GROUP_QKV", "1") != "0"              # the packed QKV gradient joins the group
_wgrad_group = _WgradGroup()


def _wgrad(dy, x, w: torch.Tensor, b: Optional[torch.Tensor], alpha: Optional[torch.Tensor] = None):
    """-> (dw, db) to return to autograd (None when accumulated in place)."""
    if _direct(w) and (b is None or _direct(b)):
        if _wgrad_group.wants(dy, w, alpha):
            _wgrad_group.add(dy, x, w, b)
            return None, None
        ops.linear_wgrad(dy, x, w.grad, None if b is None else b.grad, accumulate=True, alpha=alpha)
        _notify(w, b)
        return None, None
    dw = torch.empty(w.shape, dtype=torch.float32, device=w.device)
    db = torch.empty(b.shape, dtype=torch.float32, device=w.device) if b is not None else None
    ops.linear_wgrad(dy, x, dw, db, accumulate=False, alpha=alpha)
    return dw.to(w.dtype), (db.to(b.dtype) if b is not None else None)


def _ln_bwd(dy, x, ln_w, ln_b, mean, rstd):
    dt = x.dtype
    if _direct(ln_w) and _direct(ln_b):
        dx = ops.layernorm_bwd(dy, x, _shadow(ln_w, dt), mean, rstd, ln_w.grad, ln_b.grad, accumulate=True)
        _notify(ln_w, ln_b)
        return dx, None, None
    dg = torch.empty(ln_w.shape, dtype=torch.float32, device=x.device)
    db = torch.empty(ln_b.shape, dtype=torch.float32, device=x.device)
    dx = ops.layernorm_bwd(dy, x, _shadow(ln_w, dt), mean, rstd, dg, db, accumulate=False)
    return dx, dg.to(ln_w.dtype), db.to(ln_b.dtype)


def _defer_list(t: torch.Tensor):
    """The list a layer attached to its input tensor (`_vy_defer`): backward functions that produce a
    residual-path gradient for that tensor append it there instead of returning it, and the
    self-attention backward -- which autograd necessarily runs after them -- adds them in its dgrad
    epilogue.  None when the layer did not opt in."""
    return getattr(t, "_vy_defer", None) if t is not None else None


def defer_residual_grads(x: torch.Tensor) -> None:
    """Called by a post-LN transformer layer on its input (training only): see _defer_list."""
    if torch.is_grad_enabled() and x.requires_grad and _DEFER_RESIDUALS:
        x._vy_defer = []


def _require_bf16(x: torch.Tensor) -> None:
    """The training kernels run in bf16 (the measured path: MFMA GEMMs, flash attention) and in fp32 (the parity
    path: plain-FMA kernels, gradients checked against the reference's autograd at 1e-4)."""
    if x.dtype != BF16 and x.dtype != torch.float32:
        raise VyomHipError("training kernels take bf16 or fp32 activations (fp32 master weights), not %s" % x.dtype)


class EmbeddingFn(torch.autograd.Function):
    """hidden = table[ids] read from the compute-dtype shadow of the fp32 table; the gradient rows are
    accumulated straight into the flat fp32 gradient arena (nn.Embedding: models/decoder.py:287)."""

    @staticmethod
    def forward(ctx, ids, weight, padding_idx, dtype):
        ctx.save_for_backward(ids)
        ctx.weight, ctx.padding_idx = weight, padding_idx
        return ops.embedding(_shadow(weight, dtype), ids)

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        w = ctx.weight
        if _direct(w):
            ops.embedding_bwd_(dout.contiguous(), ids, w.grad, ctx.padding_idx)
            _notify(w)
            return None, None, None, None
        dw = torch.zeros(w.shape, dtype=torch.float32, device=w.device)
        ops.embedding_bwd_(dout.contiguous(), ids, dw, ctx.padding_idx)
        return None, dw.to(w.dtype), None, None


class PatchifyFn(torch.autograd.Function):
    """tokens = patches W^T + b with W the stride == kernel Conv2d weight (d, C, p, p) viewed (d, C*p*p): the
    ViT patch embedding (reference models/vision_encoder.py:83-88, 114).  Backward is the weight / bias
    gradient only -- the pixels are data."""

    @staticmethod
    def forward(ctx, patches, w, b):
        _require_bf16(patches)
        dt = patches.dtype
        ctx.save_for_backward(patches)
        ctx.params = (w, b)
        return ops.linear(patches, _shadow(w, dt).reshape(w.shape[0], -1), _shadow(b, dt))

    @staticmethod
    def backward(ctx, dy):
        (patches,) = ctx.saved_tensors
        w, b = ctx.params
        dy = dy.contiguous()
        N = w.shape[0]
        if _direct(w) and (b is None or _direct(b)):
            ops.linear_wgrad(dy, patches, w.grad.view(N, -1), None if b is None else b.grad, accumulate=True)
            _notify(w, b)
            return None, None, None
        dw = torch.empty((N, w.numel() // N), dtype=torch.float32, device=w.device)
        db = torch.empty(b.shape, dtype=torch.float32, device=w.device) if b is not None else None
        ops.linear_wgrad(dy, patches, dw, db, accumulate=False)
        return None, dw.view(w.shape).to(w.dtype), (db.to(b.dtype) if b is not None else None)


class LinearResidualLayerNormFn(torch.autograd.Function):
    """y = LN(x W^T + b + residual).  AttentionSelfOutput (reference layers/attention.py:69-72)."""

    @staticmethod
    def forward(ctx, x, residual, w, b, ln_w, ln_b, eps, drop=None):
        _require_bf16(x)
        dt = x.dtype
        s = ops.linear(x, _shadow(w, dt), _shadow(b, dt), residual=residual, dropout=drop)
        y, mean, rstd = ops.layernorm(s, _shadow(ln_w, dt), _shadow(ln_b, dt), eps, save_stats=True)
        ctx.save_for_backward(x, s, mean, rstd)
        ctx.params = (w, b, ln_w, ln_b)
        ctx.defer = _defer_list(residual)
        ctx.drop = drop
        return y

    @staticmethod
    def backward(ctx, dy):
        x, s, mean, rstd = ctx.saved_tensors
        w, b, ln_w, ln_b = ctx.params
        dy = dy.contiguous()
        ds, dg, dbt = _ln_bwd(dy, s, ln_w, ln_b, mean, rstd)
        # dropout sits between the projection and the residual add: the projection's gradient is ds under
        # the forward's mask (regenerated from (seed, offset)), the residual's is ds itself
        dz = ops.dropout(ds, *ctx.drop) if ctx.drop is not None else ds
        dx = ops.linear_dgrad(dz, _wt(w, x.dtype))
        dw, db = _wgrad(dz, x, w, b)
        if ctx.defer is not None:      # the residual gradient rides to the QKV dgrad epilogue
            ctx.defer.append(ds)
            ds = None
        return dx, ds, dw, db, dg, dbt, None, None


class FfnBlockFn(torch.autograd.Function):
    """y = LN(act(x W1^T + b1) W2^T + b2 + residual).  FeedForward (reference layers/ffn.py:32-40)."""

    @staticmethod
    def forward(ctx, x, residual, w1, b1, w2, b2, ln_w, ln_b, eps, act, drop=None):
        _require_bf16(x)
        dt = x.dtype
        pre = torch.empty((*x.shape[:-1], w1.shape[0]), dtype=dt, device=x.device)
        hmid = torch.empty_like(pre)
        # `pre` holds act'(x W1^T + b1), not the pre-activation: the backward multiplies by it directly
        ops.linear(x, _shadow(w1, dt), _shadow(b1, dt), act=act | _lib.ACT_SAVE_DERIV, pre_out=pre, out=hmid)
        s = ops.linear(hmid, _shadow(w2, dt), _shadow(b2, dt), residual=residual, dropout=drop)
        ctx.drop = drop
        y, mean, rstd = ops.layernorm(s, _shadow(ln_w, dt), _shadow(ln_b, dt), eps, save_stats=True)
        ctx.save_for_backward(x, pre, hmid, s, mean, rstd)
        ctx.params = (w1, b1, w2, b2, ln_w, ln_b)
        ctx.act = act
        ctx.defer = _defer_list(residual) if residual is not x else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pre, hmid, s, mean, rstd = ctx.saved_tensors
        w1, b1, w2, b2, ln_w, ln_b = ctx.params
        dt = x.dtype
        dy = dy.contiguous()
        ds, dg, dbt = _ln_bwd(dy, s, ln_w, ln_b, mean, rstd)
        dz = ops.dropout(ds, *ctx.drop) if ctx.drop is not None else ds   # the forward's mask (see above)
        dpre = ops.linear_dgrad(dz, _wt(w2, dt), pre=pre, act=ctx.act | _lib.ACT_SAVE_DERIV)  # (dz W2) * act'(pre), act' saved
        dw2, db2 = _wgrad(dz, hmid, w2, b2)
        dx = ops.linear_dgrad(dpre, _wt(w1, dt))
        dw1, db1 = _wgrad(dpre, x, w1, b1)
        if ctx.defer is not None:
            ctx.defer.append(ds)
            ds = None
        return dx, ds, dw1, db1, dw2, db2, dg, dbt, None, None, None


class SelfAttentionFn(torch.autograd.Function):
    """o = merge_heads(softmax(rope(q) rope(k)^T / sqrt(dh) + mask) v) with q,k,v = x W^T + b.
    Inputs after `mod` are the projection parameters in module order (q,k,v weights then biases,
    or the fused qkv weight and bias)."""

    @staticmethod
    def forward(ctx, x, mod, attention_mask, freqs, start_pos, *params):
        _require_bf16(x)
        B, L, _ = x.shape
        h, hk, dh = mod.num_attention_heads, mod.num_key_value_heads, mod.head_dim
        dt, dev = x.dtype, x.device
        w, b = mod._packed()
        if attention_mask is not None and not isinstance(attention_mask, AttnMask):
            raise VyomHipError("training needs a mask descriptor (AttnMask): dense additive masks have no "
                               "backward kernel")
        cos, sin, pos0 = resolve_freqs(freqs, dev)
        q = torch.empty((B, h, L, dh), dtype=dt, device=dev)
        k = torch.empty((B, hk, L, dh), dtype=dt, device=dev)
        v = torch.empty_like(k)
        sw, sb = mod._packed_shadow(dt)
        ops.qkv_rope(x, sw, sb, h, hk, dh, cos, sin, pos0, q, k, v)
        lse = torch.empty((B, h, L), dtype=torch.float32, device=dev)
        causal, kp, sp = False, None, 0
        if attention_mask is not None:
            causal, kp, sp = attention_mask.causal, attention_mask.keypad, attention_mask.start_pos
            if kp is not None:
                kp = kp[:, :L].contiguous() if kp.shape[1] != L else kp
        o = ops.attention(q, k, v, causal=causal, start_pos=sp, keypad=kp, lse=lse)
        ctx.save_for_backward(x, q, k, v, o, lse)
        ctx.meta = (mod, causal, kp, sp, cos, sin, pos0, params)
        ctx.defer = _defer_list(x)
        return o

    @staticmethod
    def backward(ctx, do):
        x, q, k, v, o, lse = ctx.saved_tensors
        mod, causal, kp, sp, cos, sin, pos0, params = ctx.meta
        B, L, _ = x.shape
        h, hk, dh = mod.num_attention_heads, mod.num_key_value_heads, mod.head_dim
        dt = x.dtype
        do = do.contiguous()
        W = (h + 2 * hk) * dh
        packed = torch.empty((B, L, W), dtype=dt, device=x.device)  # [dq | dk | dv], 'b l (h d)'
        dq = packed[:, :, : h * dh].view(B, L, h, dh).permute(0, 2, 1, 3)
        dk = packed[:, :, h * dh:(h + hk) * dh].view(B, L, hk, dh).permute(0, 2, 1, 3)
        dv = packed[:, :, (h + hk) * dh:].view(B, L, hk, dh).permute(0, 2, 1, 3)
        # RoPE is orthogonal: its backward (the inverse rotation of dq, dk) runs in the epilogues
        ops.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, causal=causal, start_pos=sp, keypad=kp,
                          cos=cos, sin=sin, rope_pos0=pos0)
        w, b = mod._packed()
        # the layer input's other gradient contributions (residual paths of this layer's out-projection
        # and feed-forward, deferred by their backward) are added in this GEMM's epilogue
        adds = ctx.defer if ctx.defer is not None else []
        extra = None
        for t in adds[2:]:
            extra = t if extra is None else extra + t
        a1 = adds[0] if len(adds) > 0 else None
        a2 = adds[1] if len(adds) > 1 else None
        if extra is not None:
            a2 = a2 + extra
        dx = ops.linear_dgrad(packed, _wt_packed(mod, w, dt), add_to=a1, add_to2=a2)
        if ctx.defer is not None:
            ctx.defer.clear()
        grads = _packed_wgrad(mod, packed, x, w, b, params)
        return (dx, None, None, None, None, *grads)


class CrossAttentionFn(torch.autograd.Function):
    """o = merge_heads(softmax(q k^T / sqrt(dh) + key-padding mask) v), q = x Wq^T + bq,
    k/v = enc Wk/v^T + b: the encoder-decoder attention of the seq2seq decoder layer
    (reference layers/attention.py:410-474, 512-573).  Gradients flow to the decoder state, to the
    ENCODER output and to the three projections."""

    @staticmethod
    def forward(ctx, x, enc, mod, enc_mask, wq, bq, wk, bk, wv, bv):
        _require_bf16(x)
        B, L, _ = x.shape
        S = enc.shape[1]
        h, hk, dh = mod.num_attention_heads, mod.num_key_value_heads, mod.head_dim
        dt, dev = x.dtype, x.device
        if enc_mask is not None and not isinstance(enc_mask, AttnMask):
            raise VyomHipError("training needs a mask descriptor (AttnMask): dense additive masks have no "
                               "backward kernel")
        q2 = ops.linear(x, _shadow(wq, dt), _shadow(bq, dt))
        k2 = ops.linear(enc, _shadow(wk, dt), _shadow(bk, dt))
        v2 = ops.linear(enc, _shadow(wv, dt), _shadow(bv, dt))
        kp = None
        if enc_mask is not None:
            if enc_mask.causal:
                raise VyomHipError("cross-attention takes a key-padding mask, not a causal one")
            kp = enc_mask.keypad
            if kp is not None:
                kp = kp[:, :S].contiguous() if kp.shape[1] != S else kp
        lse = torch.empty((B, h, L), dtype=torch.float32, device=dev)
        o = ops.attention(_heads(q2, h, dh), _heads(k2, hk, dh), _heads(v2, hk, dh), causal=False, keypad=kp, lse=lse)
        ctx.save_for_backward(x, enc, q2, k2, v2, o, lse)
        ctx.meta = (mod, kp, (wq, bq, wk, bk, wv, bv))
        return o

    @staticmethod
    def backward(ctx, do):
        x, enc, q2, k2, v2, o, lse = ctx.saved_tensors
        mod, kp, (wq, bq, wk, bk, wv, bv) = ctx.meta
        h, hk, dh = mod.num_attention_heads, mod.num_key_value_heads, mod.head_dim
        dt = x.dtype
        do = do.contiguous()
        dq2, dk2, dv2 = torch.empty_like(q2), torch.empty_like(k2), torch.empty_like(v2)
        ops.attention_bwd(_heads(q2, h, dh), _heads(k2, hk, dh), _heads(v2, hk, dh), o, do, lse,
                          _heads(dq2, h, dh), _heads(dk2, hk, dh), _heads(dv2, hk, dh), causal=False, keypad=kp)
        dx = ops.linear_dgrad(dq2, _wt(wq, dt))
        denc = ops.linear_dgrad(dk2, _wt(wk, dt))
        denc = ops.linear_dgrad(dv2, _wt(wv, dt), add_to=denc)
        dwq, dbq = _wgrad(dq2, x, wq, bq)
        dwk, dbk = _wgrad(dk2, enc, wk, bk)
        dwv, dbv = _wgrad(dv2, enc, wv, bv)
        return dx, denc, None, None, dwq, dbq, dwk, dbk, dwv, dbv


def _heads(x2: torch.Tensor, heads: int, dh: int) -> torch.Tensor:
    B, L, _ = x2.shape
    return x2.view(B, L, heads, dh).permute(0, 2, 1, 3)


def _wt_packed(mod, w, dtype):
    """W^T of the packed projection: keyed on the versions of the member parameters."""
    return _wt_cached(mod, lambda m: tuple(p._version for p in m._params()) + (WEIGHT_EPOCH[0], dtype),
                      lambda m: m._packed_shadow(dtype)[0])


def _packed_wgrad(mod, dy, x, w, b, params):
    members = mod._params()
    nw = 1 if mod._fused_qkv else 3
    ws, bs = members[:nw], members[nw:]
    if all(_direct(p) for p in members):
        # the trainer lays the member gradients out adjacently, mirroring the packed weights
        g0 = ws[0].grad
        N = sum(p.shape[0] for p in ws)
        dw = torch.as_strided(g0, (N, g0.shape[1]), (g0.stride(0), 1))
        ok = all(p.grad.data_ptr() == g0.data_ptr() + off * g0.shape[1] * 4
                 for p, off in zip(ws, _offsets(ws)))
        db = None
        if bs:
            db0 = bs[0].grad
            db = torch.as_strided(db0, (N,), (1,))
            ok = ok and all(p.grad.data_ptr() == db0.data_ptr() + off * 4 for p, off in zip(bs, _offsets(bs)))
        if ok:
            dy2 = dy.view(-1, dy.shape[-1])
            if _GROUP_WGRADS and _GROUP_QKV and dy2.shape[0] >= _GROUP_MIN_ROWS and N * dw.shape[1] >= 512 * 512 and dw.shape[1] % 8 == 0:
                # with the layer's other weight gradients in one grouped launch (27 of its 108 tiles: on its own
                # this launch ran at 670 TFLOP/s, the group at 850)
                _wgrad_group.add_tensors(dy2, x.view(-1, x.shape[-1]), dw, db, list(members))
            else:
                ops.linear_wgrad(dy, x, dw, db, accumulate=True)
                _notify(*members)
            return [None] * len(params)
    N, K = w.shape
    dw = torch.empty((N, K), dtype=torch.float32, device=w.device)
    db = torch.empty((N,), dtype=torch.float32, device=w.device) if b is not None else None
    ops.linear_wgrad(dy, x, dw, db, accumulate=False)
    out, off = [], 0
    for p in ws:
        out.append(dw[off:off + p.shape[0]].to(p.dtype))
        off += p.shape[0]
    off = 0
    for p in bs:
        out.append(db[off:off + p.shape[0]].to(p.dtype))
        off += p.shape[0]
    return out


def _offsets(ps):
    offs, o = [], 0
    for p in ps:
        offs.append(o)
        o += p.shape[0]
    return offs


class LMHeadFn(torch.autograd.Function):
    """logits = LN(gelu(h Wd^T + bd)) Wv^T + bias (reference models/decoder.py:267-275).  The logits
    row stride is padded to 8 and the pad columns are zero, so the backward GEMMs can contract
    over the padded width."""

    @staticmethod
    def forward(ctx, hidden, wd, bd, ln_w, ln_b, wv, bias, eps):
        _require_bf16(hidden)
        dt = hidden.dtype
        pre = torch.empty_like(hidden)
        g = torch.empty_like(hidden)
        ops.linear(hidden, _shadow(wd, dt), _shadow(bd, dt), act=ACT_GELU_ERF, pre_out=pre, out=g)
        n, mean, rstd = ops.layernorm(g, _shadow(ln_w, dt), _shadow(ln_b, dt), eps, save_stats=True)
        V = wv.shape[0]
        ld = _row_stride(V)
        buf = torch.zeros((*hidden.shape[:-1], ld), dtype=dt, device=hidden.device)
        logits = buf[..., :V]
        ops.linear(n, _shadow(wv, dt), _shadow(bias, dt), out=logits)
        ctx.save_for_backward(hidden, pre, g, n, mean, rstd)
        ctx.params = (wd, bd, ln_w, ln_b, wv, bias)
        ctx.ld = ld
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        hidden, pre, g, n, mean, rstd = ctx.saved_tensors
        wd, bd, ln_w, ln_b, wv, bias = ctx.params
        dt = hidden.dtype
        V, ld = wv.shape[0], ctx.ld
        # the contraction of the dgrad GEMM runs over the PADDED vocabulary width (multiple of 8, pad
        # columns of both operands zero): re-home the gradient unless it already sits in such a buffer
        if dlogits.stride(-1) != 1 or dlogits.stride(-2) != ld or dlogits.dtype != dt:
            buf = torch.zeros((*dlogits.shape[:-1], ld), dtype=dt, device=dlogits.device)
            buf[..., :V] = dlogits
        else:
            buf = torch.as_strided(dlogits, (*dlogits.shape[:-1], ld), dlogits.stride())
            if ld != V:
                buf[..., V:].zero_()
        dlogits = buf[..., :V]
        dn = ops.linear_dgrad(buf, _wt_padded(wv, dt, ld))
        dwv, dbias = _wgrad(dlogits, n, wv, bias)
        dg, dgam, dbet = _ln_bwd(dn, g, ln_w, ln_b, mean, rstd)
        dpre = _gelu_bwd(dg, pre)
        dh = ops.linear_dgrad(dpre, _wt(wd, dt))
        dwd, dbd = _wgrad(dpre, hidden, wd, bd)
        return dh, dwd, dbd, dgam, dbet, dwv, dbias, None


def _gelu_bwd(dy, pre):
    return ops.act_bwd(dy, pre, ACT_GELU_ERF)


class LMHeadLossFn(torch.autograd.Function):
    """Shifted causal-LM loss fused with the LM head: mean cross-entropy of logits[:, :-1] against
    labels[:, 1:] with ignore_index (Examples/vyom-ai-decoder_clm.ipynb cell 29).  The logits
    ([M, V] bf16, padded row stride) are produced by the vocabulary GEMM, then reduced AND
    overwritten IN PLACE by their unit gradient in one pass (vy_xent_fused; vy_xent_fwd + vy_xent_bwd
    beyond 65536 columns) -- they are never copied, up-cast or re-materialised (SURVEY.md section 8f
    item 1).  Backward multiplies by the upstream gradient through the GEMMs (a device scalar)."""

    @staticmethod
    def forward(ctx, hidden, labels, ignore_index, wd, bd, ln_w, ln_b, wv, bias, eps, err_flag=None):
        _require_bf16(hidden)
        dt, dev = hidden.dtype, hidden.device
        B, L, _ = hidden.shape
        pre = torch.empty_like(hidden)
        g = torch.empty_like(hidden)
        ops.linear(hidden, _shadow(wd, dt), _shadow(bd, dt), act=ACT_GELU_ERF, pre_out=pre, out=g)
        n, mean, rstd = ops.layernorm(g, _shadow(ln_w, dt), _shadow(ln_b, dt), eps, save_stats=True)
        V = wv.shape[0]
        ld = _row_stride(V)
        buf = torch.empty((B * L, ld), dtype=dt, device=dev)
        if ld != V:
            buf[:, V:].zero_()  # only the pad columns: the GEMM writes the rest
        logits = buf[:, :V]
        ops.linear(n.view(B * L, -1), _shadow(wv, dt), _shadow(bias, dt), out=logits)
        shifted = torch.full((B, L), ignore_index, dtype=torch.long, device=dev)
        shifted[:, :-1] = labels[:, 1:]
        shifted = shifted.view(-1)
        lse = torch.empty(B * L, dtype=torch.float32, device=dev)
        acc = torch.zeros(2, dtype=torch.float32, device=dev)  # [loss_sum, count]
        fused = V <= 65536 and dt == BF16   # (vy_xent_fused is a bf16 kernel; fp32 takes the two-pass pair)
        if fused:
            # one pass: loss AND the unit gradient (d loss / d logits for an upstream gradient of 1),
            # written over the logits; backward scales by the actual upstream gradient (linearity)
            # (rows with an out-of-range label contribute neither loss nor gradient -- the kernel raises err_flag for
            # them -- so they must not count in the mean either, exactly as on the two-pass path)
            acc[1] = ((shifted != ignore_index) & (shifted >= 0) & (shifted < V)).sum()
            ops.xent_fused_(logits, shifted, ignore_index, lse, acc[0:1], acc[1:2], _one(dev), err_flag)
        else:
            ops.xent_fwd(logits, shifted, ignore_index, lse, acc[0:1], acc[1:2], err_flag)
        ctx.save_for_backward(hidden, pre, g, n, mean, rstd, buf, shifted, lse, acc)
        ctx.params = (wd, bd, ln_w, ln_b, wv, bias)
        ctx.ignore = ignore_index
        ctx.fused = fused
        return acc[0] / acc[1].clamp_min(1.0)

    @staticmethod
    def backward(ctx, gout):
        hidden, pre, g, n, mean, rstd, buf, shifted, lse, acc = ctx.saved_tensors
        wd, bd, ln_w, ln_b, wv, bias = ctx.params
        dt = hidden.dtype
        V = wv.shape[0]
        logits = buf[:, :V]
        gs = gout.detach().to(torch.float32).reshape(1).contiguous()
        alpha = None
        if ctx.fused:
            alpha = gs          # logits already hold the unit gradient
        else:
            ops.xent_bwd_(logits, shifted, ctx.ignore, lse, gs, acc[1:2])   # logits <- dlogits
        # contract over the padded vocabulary width (pad columns of both operands are zero)
        dn = ops.linear_dgrad(buf, _wt_padded(wv, dt, buf.shape[1]))
        if alpha is not None:
            dn = dn * alpha.to(dt)
        dwv, dbias = _wgrad(logits, n.view(buf.shape[0], -1), wv, bias, alpha=alpha)
        dg, dgam, dbet = _ln_bwd(dn.view(g.shape), g, ln_w, ln_b, mean, rstd)
        dpre = _gelu_bwd(dg, pre)
        dh = ops.linear_dgrad(dpre, _wt(wd, dt))
        dwd, dbd = _wgrad(dpre, hidden, wd, bd)
        return dh, None, None, dwd, dbd, dgam, dbet, dwv, dbias, None, None


_ONES = {}


def _one(dev) -> torch.Tensor:
    t = _ONES.get(dev)
    if t is None:
        t = _ONES[dev] = torch.ones(1, dtype=torch.float32, device=dev)
    return t


def _wt_padded(param, dtype, ld):
    """[K, ld] zero-padded W^T (full padded width, for contractions over the padded vocabulary)."""
    wt = _wt(param, dtype)
    full = torch.as_strided(wt, (wt.shape[0], ld), (wt.stride(0), 1))
    assert wt.stride(0) == ld
    return full
