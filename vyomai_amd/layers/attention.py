"""Attention modules with the reference's class names, constructor / forward signatures and
parameter names (VyomAI/layers/attention.py; the DecoderAttention twins that thread ``kv_cache``
live in vyomai_amd/models/decoder.py like in the reference).

Every forward is three launches on the current HIP stream:
  1. vy_qkv_rope_fwd  -- packed Q/K/V projection + bias + RoPE + head split, K/V written straight
                         into the KV cache window when caching;
  2. vy_attn_fwd / vy_attn_decode -- flash attention with an in-register mask descriptor, GQA by
                         head indexing (repeat_kv is never materialised), heads merged on output;
  3. vy_linear_fwd (+residual) and vy_layernorm_fwd -- AttentionSelfOutput.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from .._lib import VyomHipError
from .mask import AttnMask
from .positional_embeddings import resolve_freqs


def repeat_kv(hidden_states: torch.Tensor, n_rep: int) -> torch.Tensor:
    """(B, hk, S, dh) -> (B, hk*n_rep, S, dh).  Kept for API parity (reference :8-19); the
    kernels never call it -- they map query head i to kv head i // n_rep."""
    b, hk, s, d = hidden_states.shape
    if n_rep == 1:
        return hidden_states
    return hidden_states[:, :, None, :, :].expand(b, hk, n_rep, s, d).reshape(b, hk * n_rep, s, d)


repeat_kv_einops = repeat_kv


def _cast_into(dst: torch.Tensor, src: torch.Tensor) -> None:
    """dst <- src (dtype conversion), in place: dst may be a view of the trainer's shadow arena."""
    src = src.detach()
    if src.is_cuda and src.is_contiguous() and dst.is_contiguous():
        ops.cast(src, dst)
    else:
        with torch.no_grad():
            dst.copy_(src)


def _shadow(param: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Parameter in the compute dtype.  fp32 master weights with bf16 activations use a cached
    bf16 copy, refreshed when the parameter changes.  A cache entry is (version, tensor, pinned):
    `pinned` copies are views of the trainer's shadow arena (the fused AdamW kernel rewrites them
    without touching tensor versions) and are refreshed IN PLACE when somebody else writes the
    parameter (load_state_dict, re-initialisation) -- replacing them by a detached copy would leave
    the model reading weights the optimizer no longer updates."""
    if param is None or param.dtype == dtype:
        return param
    cache = getattr(param, "_vy_shadow", None)
    ver = param._version
    if cache is not None and cache[1].dtype == dtype and cache[1].device == param.device \
            and cache[1].shape == param.shape:
        if cache[0] == ver:
            return cache[1]
        if len(cache) > 2 and cache[2]:
            _cast_into(cache[1], param)
            param._vy_shadow = (ver, cache[1], True)
            return cache[1]
    cache = (ver, param.detach().to(dtype), False)
    param._vy_shadow = cache
    return cache[1]


class AttentionSelfOutput(nn.Module):
    """LN(dropout(dense(hidden)) + input).  Reference :42-72."""

    def __init__(self, config, bias: Optional[bool] = True, out_features: Optional[int] = None):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size,
                               config.hidden_size if out_features is None else out_features, bias=bias)
        self.layernorm = nn.LayerNorm(config.hidden_size, eps=getattr(config, "layer_norm_eps", 1e-6))
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states: torch.Tensor, input_tensor: torch.Tensor) -> torch.Tensor:
        from ..autograd import linear_residual_layernorm
        return linear_residual_layernorm(hidden_states, input_tensor, self.dense.weight, self.dense.bias,
                                         self.layernorm.weight, self.layernorm.bias, self.layernorm.eps,
                                         self.dropout.p, self.training)


class _SelfAttentionBase(nn.Module):
    """Shared machinery of the self-attention variants (not a reference class)."""

    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int

    def _setup(self, config, layer_idx: int, kv_heads: Optional[int], fused_qkv: bool) -> None:
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError(
                f"The hidden size ({config.hidden_size}) is not a multiple of the number of attention "
                f"heads ({config.num_attention_heads})")
        self.layer_idx = layer_idx
        self.num_attention_heads = config.num_attention_heads
        self.head_dim = int(config.hidden_size // config.num_attention_heads)
        self.head_size = self.head_dim
        self.attention_bias = getattr(config, "attention_bias", True)
        if kv_heads is None:
            self.num_key_value_heads = self.num_attention_heads
        else:
            self.num_key_value_heads = kv_heads
            self.num_key_value_groups = self.num_attention_heads // kv_heads
            if self.num_attention_heads % kv_heads != 0 or self.num_attention_heads < kv_heads:
                raise ValueError(
                    f"num_key_value_heads {kv_heads }  should be less than equal num_attention_heads "
                    f"{config.num_attention_heads} and  multiple of num_attention_heads {config.num_attention_heads} ")
        d = config.hidden_size
        if fused_qkv:
            self.qkv = nn.Linear(d, 3 * d)
        else:
            kvw = self.num_key_value_heads * self.head_dim
            self.query = nn.Linear(d, d, bias=self.attention_bias)
            self.key = nn.Linear(d, kvw, bias=self.attention_bias)
            self.value = nn.Linear(d, kvw, bias=self.attention_bias)
        self.out = AttentionSelfOutput(config=config, bias=self.attention_bias)
        self._fused_qkv = fused_qkv

    # ---- packed [Wq; Wk; Wv] ------------------------------------------------------------------
    def _packed(self) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """The three projection weights as ONE (Nq+2Nkv, K) matrix for the fused kernel.  The
        nn.Linear parameters stay (state_dict names, LoRA wrapping) but are re-pointed to be
        views of one packed buffer the first time, or after .to()/load_state_dict moved them."""
        self._plain_projections()
        if self._fused_qkv:
            return self.qkv.weight, self.qkv.bias
        ws = [self.query.weight, self.key.weight, self.value.weight]
        packed = getattr(self, "_packed_w", None)
        ok = packed is not None and packed.device == ws[0].device and packed.dtype == ws[0].dtype
        if ok:
            off = 0
            for w in ws:
                if w.data_ptr() != packed.data_ptr() + off * packed.element_size() * packed.shape[1]:
                    ok = False
                    break
                off += w.shape[0]
        if not ok:
            with torch.no_grad():
                packed = torch.cat([w.detach() for w in ws], dim=0).contiguous()
                off = 0
                for w in ws:
                    w.data = packed[off:off + w.shape[0]]
                    off += w.shape[0]
                self._packed_w = packed
                if self.attention_bias:
                    bs = [self.query.bias, self.key.bias, self.value.bias]
                    pb = torch.cat([b.detach() for b in bs], dim=0).contiguous()
                    off = 0
                    for b in bs:
                        b.data = pb[off:off + b.shape[0]]
                        off += b.shape[0]
                    self._packed_b = pb
                else:
                    self._packed_b = None
        elif self.attention_bias:
            bs = [self.query.bias, self.key.bias, self.value.bias]
            pb = self._packed_b
            off = 0
            for b in bs:
                if b.data_ptr() != pb.data_ptr() + off * pb.element_size():
                    with torch.no_grad():
                        pb = torch.cat([x.detach() for x in bs], dim=0).contiguous()
                        o2 = 0
                        for x in bs:
                            x.data = pb[o2:o2 + x.shape[0]]
                            o2 += x.shape[0]
                        self._packed_b = pb
                    break
                off += b.shape[0]
        return self._packed_w, self._packed_b

    def _plain_projections(self) -> None:
        """The fused QKV kernel reads query/key/value.weight itself and never calls the sub-modules, so a projection that
        has been replaced by a wrapper -- the reference's adapters swap `attention.query` for a LoraLinear / DoraLinear
        whose weight lives at `.linear.weight` (VyomAI/layers/adapters.py:21-24, 58-62) -- would be ignored (or crash on
        a missing attribute).  Adapters are outside SURVEY section 8; say so instead of computing without them."""
        for name in (("qkv",) if self._fused_qkv else ("query", "key", "value")):
            m = getattr(self, name)
            if not isinstance(m, nn.Linear):
                raise VyomHipError(
                    f"attention.{name} is a {type(m).__name__}, not an nn.Linear: the MI355X attention path reads the "
                    "projection weights directly (one fused QKV launch) and would skip the wrapper's extra term.  "
                    "Adapter wrappers (LoRA / DoRA) are out of scope here -- merge the low-rank update into "
                    f"`{name}.weight` (W + alpha * B A) and restore the nn.Linear before running on the HIP path.")

    def _packed_shadow(self, dtype: torch.dtype) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """_packed() in the compute dtype.  The packed buffer has a version counter of its own that
        in-place writes to query/key/value.weight (optimizer.step() of a torch optimizer,
        load_state_dict) never bump, so the copy is keyed on the MEMBER parameters' versions (as the
        transposed copy for the backward GEMMs is).  Entry: (key, w, b, packed data_ptr, pinned);
        pinned = views of the trainer's shadow arena, refreshed in place (see _shadow)."""
        w, b = self._packed()
        if w.dtype == dtype:
            return w, b
        if self._fused_qkv:
            return _shadow(w, dtype), _shadow(b, dtype)
        key = tuple(p._version for p in self._params())
        c = getattr(self, "_vy_pshadow", None)
        if c is not None and c[1].dtype == dtype and c[1].device == w.device and c[3] == w.data_ptr() \
                and (c[2] is None) == (b is None):
            if c[0] == key:
                return c[1], c[2]
            if c[4]:
                _cast_into(c[1], w)
                if b is not None:
                    _cast_into(c[2], b)
                self._vy_pshadow = (key, c[1], c[2], c[3], True)
                return c[1], c[2]
        sw = w.detach().to(dtype)
        sb = None if b is None else b.detach().to(dtype)
        self._vy_pshadow = (key, sw, sb, w.data_ptr(), False)
        return sw, sb

    def _params(self):
        self._plain_projections()
        if self._fused_qkv:
            return [self.qkv.weight, self.qkv.bias]
        ps = [self.query.weight, self.key.weight, self.value.weight]
        if self.attention_bias:
            ps += [self.query.bias, self.key.bias, self.value.bias]
        return ps

    def _attend(self, hidden_state: torch.Tensor, attention_mask, freqs, cache=None,
                cache_index: Optional[int] = None, start_pos: int = 0) -> torch.Tensor:
        from ..autograd import self_attention_block
        if not hidden_state.is_cuda:
            raise VyomHipError("vyomai_amd attention runs on MI355X only (got a CPU tensor; there is "
                               "no CPU fallback -- the CPU restatement lives in oracle/ for tests)")
        return self_attention_block(self, hidden_state, attention_mask, freqs, cache, cache_index, start_pos)


class EncoderAttention(_SelfAttentionBase):
    """forward(hidden_state, attention_mask, freqs=None) -> Tensor.  Reference :75-133."""

    def __init__(self, config, layer_idx: int) -> None:
        super().__init__()
        self._setup(config, layer_idx, None, fused_qkv=False)

    def forward(self, hidden_state, attention_mask, freqs=None) -> torch.Tensor:
        return self._attend(hidden_state, attention_mask, freqs)


class EncoderAttentionGqa(_SelfAttentionBase):
    """Grouped-query encoder attention (num_key_value_heads default 4).  Reference :136-215."""

    def __init__(self, config, layer_idx: int) -> None:
        super().__init__()
        self._setup(config, layer_idx, getattr(config, "num_key_value_heads", 4), fused_qkv=False)

    def forward(self, hidden_state, attention_mask, freqs=None) -> torch.Tensor:
        return self._attend(hidden_state, attention_mask, freqs)


class DecoderAttention(_SelfAttentionBase):
    """Decoder self-attention with the cache attached as ``self.cache`` (model._setup_cache()).
    forward(hidden_state, attention_mask, freqs, use_cache, start_pos).  Reference :218-289."""

    def __init__(self, config, layer_idx: int) -> None:
        super().__init__()
        self._setup(config, layer_idx, None, fused_qkv=False)

    def forward(self, hidden_state, attention_mask, freqs=None, use_cache: Optional[bool] = False,
                start_pos: Optional[int] = 0) -> torch.Tensor:
        cache = None
        if use_cache:
            cache = getattr(self, "cache", None)
            if cache is None:
                raise ValueError("you need to setup cache for every attention layer with model.setup_cache()")
        return self._attend(hidden_state, attention_mask, freqs, cache, None, start_pos)


class DecoderAttentionGqa(_SelfAttentionBase):
    """Reference :292-379."""

    def __init__(self, config, layer_idx: int) -> None:
        super().__init__()
        self._setup(config, layer_idx, getattr(config, "num_key_value_heads", 4), fused_qkv=False)

    def forward(self, hidden_state, attention_mask, freqs=None, use_cache: Optional[bool] = False,
                start_pos: Optional[int] = 0) -> torch.Tensor:
        cache = None
        if use_cache:
            cache = getattr(self, "cache", None)
            if cache is None:
                raise ValueError("you need to setup cache for every attention layer with model._setup_cache()")
        return self._attend(hidden_state, attention_mask, freqs, cache, None, start_pos)


class VisionAttention(_SelfAttentionBase):
    """ViT attention with one fused ``qkv`` Linear.  Reference :576-624."""

    def __init__(self, config, layer_idx: int) -> None:
        super().__init__()
        self._setup(config, layer_idx, None, fused_qkv=True)

    def forward(self, hidden_state, attention_mask, freqs=None) -> torch.Tensor:
        return self._attend(hidden_state, attention_mask, freqs)


class _CrossAttentionBase(_SelfAttentionBase):
    """Shared body of the encoder-decoder (cross) attention variants (not a reference class)."""

    def forward(self, hidden_state: torch.Tensor, encoder_hidden_state: torch.Tensor, encoder_attention_mask,
                freqs=None, use_cache: Optional[bool] = False) -> torch.Tensor:
        """q from `hidden_state`, k/v from `encoder_hidden_state` (computed once and kept in
        ``self.cache`` when use_cache), no RoPE (the reference leaves it commented out), then
        AttentionSelfOutput with `hidden_state` as residual.  `encoder_attention_mask` is a key-padding
        descriptor (AttnMask) or the reference's additive (B,1,1,S) tensor."""
        from ..autograd import cross_attention_block
        if not hidden_state.is_cuda:
            raise VyomHipError("vyomai_amd attention runs on MI355X only (got a CPU tensor; there is "
                               "no CPU fallback -- the CPU restatement lives in oracle/ for tests)")
        return cross_attention_block(self, hidden_state, encoder_hidden_state, encoder_attention_mask, bool(use_cache))


class EncoderDecoderAttention(_CrossAttentionBase):
    """Reference layers/attention.py:382-474."""

    def __init__(self, config, layer_idx: int) -> None:
        super().__init__()
        self._setup(config, layer_idx, None, fused_qkv=False)


class EncoderDecoderAttentionGqa(_CrossAttentionBase):
    """Reference layers/attention.py:477-573 (K/V projections of num_key_value_heads heads, default 4)."""

    def __init__(self, config, layer_idx: int) -> None:
        super().__init__()
        self._setup(config, layer_idx, getattr(config, "num_key_value_heads", 4), fused_qkv=False)

