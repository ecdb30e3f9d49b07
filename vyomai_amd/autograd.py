"""The three fused groups of a VyomAI layer, forward and backward, on the HIP kernels:

  self_attention_block      QKV+RoPE -> flash attention -> out-proj + residual -> LayerNorm
  linear_residual_layernorm AttentionSelfOutput on its own
  ffn_block                 GEMM+GELU -> GEMM + residual -> LayerNorm

The grouping follows the reference author's own fused ops (Examples/vyom-ai-decoder-fused.ipynb
cells 2-7: LinearRms, FFNGeLU, ScaledDotProductAttention, RotaryEmbeddingFunction).  Inference
calls go straight to the forward kernels; when gradients are required the same kernels run inside
torch.autograd.Function wrappers whose backward is the vy_*_bwd / dgrad / wgrad entry points.
PyTorch autograd is only the tape; no PyTorch math runs in either direction.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from ._lib import VyomHipError
from .layers.mask import AttnMask
from .layers.positional_embeddings import resolve_freqs


def _shadow(param: Optional[torch.Tensor], dtype: torch.dtype) -> Optional[torch.Tensor]:
    from .layers.attention import _shadow as s
    return s(param, dtype)


def _wants_grad(*ts) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


# ------------------------------------------------------------------------------------------
# forward-only (inference) paths
# ------------------------------------------------------------------------------------------


def _drop_args(p_drop: float, training: bool):
    """(p, seed, offset) of one dropout call when the module is in train() with p > 0 (nn.Dropout's
    condition in the reference: layers/attention.py:70, layers/ffn.py:38), else None."""
    if not training or p_drop <= 0.0:
        return None
    from .rng import next_dropout
    return next_dropout(p_drop)


def linear_residual_layernorm(x, residual, w, b, ln_w, ln_b, eps, p_drop: float = 0.0, training: bool = False):
    """LN(dropout(x W^T + b) + residual): AttentionSelfOutput (reference layers/attention.py:69-72)."""
    drop = _drop_args(p_drop, training)
    if _wants_grad(x, residual, w, b, ln_w, ln_b):
        from .autograd_train import LinearResidualLayerNormFn
        return LinearResidualLayerNormFn.apply(x, residual, w, b, ln_w, ln_b, eps, drop)
    dt = x.dtype
    s = ops.linear(x, _shadow(w, dt), _shadow(b, dt), residual=residual, dropout=drop)
    y, _, _ = ops.layernorm(s, _shadow(ln_w, dt), _shadow(ln_b, dt), eps)
    return y


def ffn_block(x, residual, w1, b1, w2, b2, ln_w, ln_b, eps, act, p_drop: float = 0.0, training: bool = False):
    """LN(dropout(act(x W1^T + b1) W2^T + b2) + residual): FeedForward (reference layers/ffn.py:32-40)."""
    drop = _drop_args(p_drop, training)
    if _wants_grad(x, residual, w1, b1, w2, b2, ln_w, ln_b):
        from .autograd_train import FfnBlockFn
        return FfnBlockFn.apply(x, residual, w1, b1, w2, b2, ln_w, ln_b, eps, act, drop)
    dt = x.dtype
    hmid = ops.linear(x, _shadow(w1, dt), _shadow(b1, dt), act=act)
    s = ops.linear(hmid, _shadow(w2, dt), _shadow(b2, dt), residual=residual, dropout=drop)
    y, _, _ = ops.layernorm(s, _shadow(ln_w, dt), _shadow(ln_b, dt), eps)
    return y


def _mask_args(mask, B, L, S, device):
    """-> dict(causal, start_pos, keypad, addmask) for ops.attention."""
    if mask is None:
        return dict(causal=False, start_pos=0, keypad=None, addmask=None)
    if isinstance(mask, AttnMask):
        kp = mask.keypad
        if kp is not None:
            if kp.shape[-1] < S:
                raise ValueError(f"attention mask covers {kp.shape[-1]} keys, attention sees {S}")
            kp = kp[:, :S]
            if kp.device != device:
                kp = kp.to(device)
            if not kp.is_contiguous() and kp.stride(1) != 1:
                kp = kp.contiguous()
        return dict(causal=mask.causal, start_pos=mask.start_pos, keypad=kp, addmask=None)
    # dense additive float mask, the reference's interface: (B|1, 1, L|1, S)
    if mask.dim() != 4 or mask.shape[-1] != S:
        raise ValueError(f"additive attention_mask must be (B,1,L,S) with S={S}, got {tuple(mask.shape)}")
    am = mask.to(device=device, dtype=torch.float32)
    if am.stride(3) != 1:
        am = am.contiguous()
    return dict(causal=False, start_pos=0, keypad=None, addmask=am)


def self_attention_block(mod, x, attention_mask, freqs, cache, cache_index, start_pos):
    """mod: a vyomai_amd.layers.attention._SelfAttentionBase.  Returns LN(out(attn(x)) + x)."""
    B, L, _ = x.shape
    h, hk, dh = mod.num_attention_heads, mod.num_key_value_heads, mod.head_dim
    dt, dev = x.dtype, x.device
    w, b = mod._packed()
    aso = mod.out
    train = _wants_grad(x, w, b, aso.dense.weight)
    if train:
        if cache is not None:
            raise VyomHipError("KV caching is an inference feature; call under torch.no_grad()")
        from .autograd_train import SelfAttentionFn
        o = SelfAttentionFn.apply(x, mod, attention_mask, freqs, start_pos, *mod._params())
        return linear_residual_layernorm(o, x, aso.dense.weight, aso.dense.bias, aso.layernorm.weight,
                                         aso.layernorm.bias, aso.layernorm.eps, aso.dropout.p, aso.training)
    cos, sin, pos0 = resolve_freqs(freqs, dev)
    q = torch.empty((B, h, L, dh), dtype=dt, device=dev)
    if cache is not None:
        if cache_index is None:
            kw, vw = cache.reserve(B, hk, L, dh, start_pos, dt, dev)
        else:
            kw, vw = cache.reserve(cache_index, B, hk, L, dh, start_pos, dt, dev)
    else:
        kw = torch.empty((B, hk, L, dh), dtype=dt, device=dev)
        vw = torch.empty_like(kw)
    sw, sb = mod._packed_shadow(dt)
    ops.qkv_rope(x, sw, sb, h, hk, dh, cos, sin, pos0, q, kw, vw)
    if cache is not None:
        k_all, v_all = cache.commit() if cache_index is None else cache.commit(cache_index)
    else:
        k_all, v_all = kw, vw
    S = k_all.shape[2]
    if L == 1 and attention_mask is None:
        o = ops.attention_decode(q, k_all, v_all, S)
    else:
        o = ops.attention(q, k_all, v_all, **_mask_args(attention_mask, B, L, S, dev))
    return linear_residual_layernorm(o, x, aso.dense.weight, aso.dense.bias, aso.layernorm.weight,
                                     aso.layernorm.bias, aso.layernorm.eps, aso.dropout.p, aso.training)


def _split(x2: torch.Tensor, heads: int, dh: int) -> torch.Tensor:
    """(B, L, heads*dh) projection output -> (B, heads, L, dh) strided view (no copy)."""
    B, L, _ = x2.shape
    return x2.view(B, L, heads, dh).permute(0, 2, 1, 3)


def cross_attention_block(mod, x, enc, enc_mask, use_cache: bool):
    """mod: layers.attention._CrossAttentionBase.  Returns LN(out(attn(q(x), k(enc), v(enc))) + x).
    Reference layers/attention.py:410-474 / 512-573."""
    B, L, _ = x.shape
    h, hk, dh = mod.num_attention_heads, mod.num_key_value_heads, mod.head_dim
    dt, dev = x.dtype, x.device
    aso = mod.out
    wq, bq = mod.query.weight, mod.query.bias
    wk, bk = mod.key.weight, mod.key.bias
    wv, bv = mod.value.weight, mod.value.bias
    if _wants_grad(x, enc, wq, wk, wv, aso.dense.weight):
        if use_cache:
            raise VyomHipError("KV caching is an inference feature; call under torch.no_grad()")
        from .autograd_train import CrossAttentionFn
        o = CrossAttentionFn.apply(x, enc, mod, enc_mask, wq, bq, wk, bk, wv, bv)
        return linear_residual_layernorm(o, x, aso.dense.weight, aso.dense.bias, aso.layernorm.weight,
                                         aso.layernorm.bias, aso.layernorm.eps, aso.dropout.p, aso.training)

    def project():
        k2 = ops.linear(enc, _shadow(wk, dt), _shadow(bk, dt))
        v2 = ops.linear(enc, _shadow(wv, dt), _shadow(bv, dt))
        return _split(k2, hk, dh), _split(v2, hk, dh)

    q = _split(ops.linear(x, _shadow(wq, dt), _shadow(bq, dt)), h, dh)
    if use_cache:
        cache = getattr(mod, "cache", None)
        if cache is None:
            raise ValueError("use_cache is True please enable model._setup_cache() to use kv-cache")
        if len(cache) == 0:   # first step: the encoder states are the same for the whole generation
            k, v = cache.update(*project())
        else:
            k, v = cache.get()
    else:
        k, v = project()
    S = k.shape[2]
    o = ops.attention(q, k, v, **_mask_args(enc_mask, B, L, S, dev))
    return linear_residual_layernorm(o, x, aso.dense.weight, aso.dense.bias, aso.layernorm.weight,
                                     aso.layernorm.bias, aso.layernorm.eps, aso.dropout.p, aso.training)

