"""CPU oracle for the VyomAI transformer hot path.  TEST INFRASTRUCTURE ONLY.

This file is a functional, plain-torch-on-CPU restatement of the reference
algorithm (Ajax0564/VyomAI, read-only at /root/reference in the build
container).  It is *not* part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline.  The product path
(``vyomai_amd``) never routes through it.

Pinning: every function below is checked against outputs of the real reference
(imported in the build container by ``tests/golden/make_golden.py``) stored as
fixtures under ``tests/golden/`` -- see ``tests/test_oracle_golden.py``.

All functions take a ``state_dict``-like mapping ``sd`` keyed with the
reference's parameter names, so the same deterministic weights
(``vyomai_amd.recipe``) drive the reference, the oracle and the HIP path.

Citations are ``path:line`` relative to the reference checkout.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# ----------------------------------------------------------------------------
# elementary ops
# ----------------------------------------------------------------------------


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """nn.Linear: y = x W^T + b, W stored (out, in).  layers/attention.py:87-95."""
    y = x @ w.t()
    return y if b is None else y + b


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() default (exact erf form).  layers/ffn.py:7-15,29."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def gelu_tanh(x: Tensor) -> Tensor:
    """tanh-GELU used by the PaliGemma cells (Examples/paligemma.ipynb cell 9, 13)."""
    c = math.sqrt(2.0 / math.pi)
    return 0.5 * x * (1.0 + torch.tanh(c * (x + 0.044715 * x * x * x)))


_ACT = {"gelu": gelu_erf, "gelu_tanh": gelu_tanh, "silu": torch.nn.functional.silu,
        "swish": torch.nn.functional.silu, "tanh": torch.tanh, "sigmoid": torch.sigmoid,
        "relu6": torch.nn.functional.relu6, "leaky_relu": torch.nn.functional.leaky_relu}


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """nn.LayerNorm over the last dim, biased variance.  layers/attention.py:52-54."""
    xf = x.float()
    mu = xf.mean(-1, keepdim=True)
    var = ((xf - mu) ** 2).mean(-1, keepdim=True)
    y = (xf - mu) * torch.rsqrt(var + eps)
    return (y * w.float() + b.float()).to(x.dtype)


def rms_norm_gemma(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """GemmaRMSNorm: x * rsqrt(mean x^2 + eps) * (1 + w) in fp32 (paligemma.ipynb cell 11)."""
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return (y * (1.0 + w.float())).to(x.dtype)


# ----------------------------------------------------------------------------
# rotary embedding  (layers/positional_embeddings.py:109-182)
# ----------------------------------------------------------------------------


def rotary_angles(head_dim: int, seq_len: int) -> Tensor:
    """RotaryEmbedding.forward: raw angles (1, seq_len, head_dim/2) fp32.  :121-137."""
    inv_freq = 1.0 / (10000 ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(seq_len).type_as(inv_freq)
    return torch.einsum("i,j->ij", t, inv_freq)[None, :, :]


def rotate_half(x: Tensor) -> Tensor:
    """NeoX half split: (-x2, x1).  :140-151."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rotary(q: Tensor, k: Tensor, freqs: Tensor) -> Tuple[Tensor, Tensor]:
    """apply_rotary_pos_emb with unsqueeze_dim=1.  :155-182.

    cos/sin are evaluated in fp32 on cat(freqs, freqs) and cast to q.dtype *before*
    the multiply, exactly like the reference.
    """
    emb = torch.cat((freqs, freqs), dim=-1)
    cos = emb.cos().to(q.dtype).unsqueeze(1)
    sin = emb.sin().to(q.dtype).unsqueeze(1)
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


# ----------------------------------------------------------------------------
# attention core
# ----------------------------------------------------------------------------


def repeat_kv(x: Tensor, n_rep: int) -> Tensor:
    """(B, hk, S, dh) -> (B, hk*n_rep, S, dh); layers/attention.py:8-19."""
    if n_rep == 1:
        return x
    b, hk, s, d = x.shape
    return x[:, :, None].expand(b, hk, n_rep, s, d).reshape(b, hk * n_rep, s, d)


def sdpa(q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor], fused: bool = False) -> Tensor:
    """softmax(q k^T / sqrt(dh) + mask) v -- F.scaled_dot_product_attention with an
    additive float mask and is_causal=False (all nine call sites, e.g.
    layers/attention.py:128, models/decoder.py:107).

    ``fused=True`` calls the same aten op the reference calls (used only when the
    oracle is *timed* as the CPU baseline); the default spells the math out.
    """
    if fused:
        return torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=mask)
    scale = 1.0 / math.sqrt(q.shape[-1])
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    if mask is not None:
        s = s + mask.float()
    p = torch.softmax(s, dim=-1)
    return (p @ v.float()).to(q.dtype)


def split_heads(x: Tensor, dh: int) -> Tensor:
    """rearrange 'b l (h d) -> b h l d'."""
    b, l, hd = x.shape
    return x.view(b, l, hd // dh, dh).permute(0, 2, 1, 3)


def merge_heads(x: Tensor) -> Tensor:
    """rearrange 'b h l d -> b l (h d)'."""
    b, h, l, d = x.shape
    return x.permute(0, 2, 1, 3).reshape(b, l, h * d)


# ----------------------------------------------------------------------------
# KV caches  (layers/kv_cache.py)
# ----------------------------------------------------------------------------


class OracleDynamicCache:
    """Per-layer list of growing K/V (DynamicCacheOne :171-252 / DynamicCache :11-78)."""

    def __init__(self, num_layers: int):
        self.k: List[Optional[Tensor]] = [None] * num_layers
        self.v: List[Optional[Tensor]] = [None] * num_layers

    def update(self, index: int, k: Tensor, v: Tensor, start_pos: int = 0):
        if self.k[index] is None:
            self.k[index], self.v[index] = k.clone(), v.clone()
        else:
            self.k[index] = torch.cat([self.k[index], k], dim=-2)
            self.v[index] = torch.cat([self.v[index], v], dim=-2)
        return self.k[index], self.v[index]


class OracleStaticCache:
    """Preallocated (B, heads, max_len, dh) per layer, slice-write at start_pos and
    return the [:start_pos+L] prefix (StaticCacheOne :255-361 / StaticCache :81-148)."""

    def __init__(self, num_layers: int, batch: int, heads: int, max_len: int, dh: int,
                 dtype=torch.float32):
        self.k = [torch.zeros(batch, heads, max_len, dh, dtype=dtype) for _ in range(num_layers)]
        self.v = [torch.zeros(batch, heads, max_len, dh, dtype=dtype) for _ in range(num_layers)]

    def update(self, index: int, k: Tensor, v: Tensor, start_pos: int = 0):
        b, _, l, _ = k.shape
        if l > self.k[index].shape[2]:
            raise ValueError("more tokens than the static cache holds")
        self.k[index][:b, :, start_pos:start_pos + l] = k
        self.v[index][:b, :, start_pos:start_pos + l] = v
        return self.k[index][:b, :, :start_pos + l], self.v[index][:b, :, :start_pos + l]


# ----------------------------------------------------------------------------
# modules (functional)
# ----------------------------------------------------------------------------


@dataclass
class Cfg:
    hidden_size: int = 768
    num_attention_heads: int = 12
    max_position_embeddings: int = 514
    num_hidden_layers: int = 4
    vocab_size: int = 50265
    layer_norm_eps: float = 1e-5
    hidden_act: str = "gelu"
    num_key_value_heads: int = 4  # only read by GQA variants (default 4, attention.py:150)

    @classmethod
    def of(cls, config) -> "Cfg":
        kw = {}
        for f in cls.__dataclass_fields__:
            if hasattr(config, f):
                kw[f] = getattr(config, f)
        return cls(**kw)


def attention_self_output(sd: SD, p: str, x: Tensor, residual: Tensor, eps: float,
                          drop: Optional[Tensor] = None) -> Tensor:
    """AttentionSelfOutput: LN(dropout(dense(x)) + residual); dropout is identity in eval.
    layers/attention.py:57-72.  `drop` = keep mask already scaled by 1/(1-p) (what nn.Dropout multiplies
    by in train(), :70), given explicitly so a test can use the mask the kernels generated."""
    y = linear(x, sd[p + "dense.weight"], sd.get(p + "dense.bias"))
    if drop is not None:
        y = y * drop.to(y.dtype)
    return layer_norm(y + residual, sd[p + "layernorm.weight"], sd[p + "layernorm.bias"], eps)


def self_attention(sd: SD, p: str, cfg: Cfg, x: Tensor, mask: Optional[Tensor],
                   freqs: Optional[Tensor], gqa: bool, fused_qkv: bool = False,
                   cache=None, layer_idx: int = 0, start_pos: int = 0,
                   fused_sdpa: bool = False, drop: Optional[Tensor] = None) -> Tensor:
    """Encoder/Decoder/Vision self-attention + AttentionSelfOutput.

    vanilla: layers/attention.py:99-133, 245-289; models/decoder.py:71-113
    gqa:     layers/attention.py:175-215, 331-379; models/decoder.py:155-201
    vision (fused qkv Linear + chunk): layers/attention.py:591-624
    """
    dh = cfg.hidden_size // cfg.num_attention_heads
    if fused_qkv:
        q, k, v = linear(x, sd[p + "qkv.weight"], sd.get(p + "qkv.bias")).chunk(3, dim=-1)
    else:
        q = linear(x, sd[p + "query.weight"], sd.get(p + "query.bias"))
        k = linear(x, sd[p + "key.weight"], sd.get(p + "key.bias"))
        v = linear(x, sd[p + "value.weight"], sd.get(p + "value.bias"))
    q, k, v = split_heads(q, dh), split_heads(k, dh), split_heads(v, dh)
    if freqs is not None:
        q, k = apply_rotary(q, k, freqs)
    if cache is not None:
        k, v = cache.update(layer_idx, k, v, start_pos)
    if gqa:
        n_rep = cfg.num_attention_heads // cfg.num_key_value_heads
        k, v = repeat_kv(k, n_rep), repeat_kv(v, n_rep)
    o = merge_heads(sdpa(q, k, v, mask, fused=fused_sdpa))
    eps = cfg.layer_norm_eps
    return attention_self_output(sd, p + "out.", o, x, eps, drop)


def feed_forward(sd: SD, p: str, cfg: Cfg, x: Tensor, input_tensor: Tensor,
                 drop: Optional[Tensor] = None) -> Tensor:
    """FeedForward: LN(dropout(W2 act(W1 x + b1) + b2) + input_tensor).  layers/ffn.py:32-40.
    Width is 4*hidden (``multiplier``), not config.intermediate_size (:19-23).  `drop`: see
    attention_self_output."""
    act = _ACT.get(cfg.hidden_act, gelu_erf)
    h = act(linear(x, sd[p + "intermediate.weight"], sd[p + "intermediate.bias"]))
    y = linear(h, sd[p + "out.weight"], sd[p + "out.bias"])
    if drop is not None:
        y = y * drop.to(y.dtype)
    return layer_norm(y + input_tensor, sd[p + "layernorm.weight"], sd[p + "layernorm.bias"],
                      cfg.layer_norm_eps)


def block(sd: SD, p: str, cfg: Cfg, h: Tensor, mask, freqs, gqa: bool, fused_qkv=False,
          cache=None, layer_idx=0, start_pos=0, fused_sdpa=False, drops=(None, None)) -> Tensor:
    """One layer: a = attn(h); return ffn(a, h) -- the FFN residual is the *layer input*
    (models/encoder.py:60-64, models/decoder.py:241-250, models/vision_encoder.py:49-53)."""
    a = self_attention(sd, p + "attention.", cfg, h, mask, freqs, gqa, fused_qkv, cache,
                       layer_idx, start_pos, fused_sdpa, drops[0])
    return feed_forward(sd, p + "feed_forward.", cfg, a, h, drops[1])


def lm_head(sd: SD, p: str, cfg: Cfg, h: Tensor) -> Tensor:
    """LMHead: decoder(LN(gelu(dense(h)))), LN eps from config (default 1e-6).
    models/decoder.py:253-275."""
    x = gelu_erf(linear(h, sd[p + "dense.weight"], sd[p + "dense.bias"]))
    x = layer_norm(x, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], cfg.layer_norm_eps)
    return linear(x, sd[p + "decoder.weight"], sd[p + "bias"])


def sinusoidal_table(max_pos: int, d: int) -> Tensor:
    """SinusoidalEncoding table (1, max_pos, d).  layers/positional_embeddings.py:80-101."""
    pe = torch.zeros(1, max_pos, d)
    pos = torch.arange(0, max_pos).unsqueeze(1).float()
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float) * -(torch.log(torch.tensor(10000.0)) / d))
    pe[:, :, 0::2] = torch.sin(pos * div)
    pe[:, :, 1::2] = torch.cos(pos * div)
    return pe


def _position_info(sd: SD, cfg: Cfg, pos_type: str, start: int, length: int) -> Optional[Tensor]:
    if pos_type == "absolute":  # AbsoluteEncoding :44-51
        if cfg.max_position_embeddings < start + length:
            raise ValueError("sequence longer than max_position_embeddings")
        return sd["position_embeddings.pos_embeddings.weight"][None, start:start + length]
    if pos_type == "sinusoidal":
        return sinusoidal_table(cfg.max_position_embeddings, cfg.hidden_size)[:, start:start + length]
    return None


def decoder_additive_mask(batch: int, seq: int, attention_mask: Optional[Tensor], start_pos: int,
                          dtype) -> Tensor:
    """create_mask_for_decoder + inversion: (B,1,L,start+L) additive mask with finfo.min.
    models/decoder.py:360-362, 376-419."""
    if attention_mask is None:
        attention_mask = torch.ones(batch, seq + start_pos)
    ids = torch.arange(seq)
    causal = (ids[None, None, :].repeat(batch, seq, 1) <= ids[None, :, None]).to(attention_mask.dtype)
    if start_pos > 0:
        causal = torch.cat([torch.ones(batch, seq, start_pos, dtype=causal.dtype), causal], dim=-1)
    ext = causal[:, None, :, :] * attention_mask[:, None, None, :]
    return ((1.0 - ext) * torch.finfo(dtype).min).to(dtype)


def padding_additive_mask(attention_mask: Tensor, dtype) -> Tensor:
    """(B,L) 0/1 -> (B,1,1,L) additive.  models/encoder.py:161-164."""
    m = attention_mask[:, None, None, :].to(dtype)
    return (1.0 - m) * torch.finfo(dtype).min


@dataclass
class DecoderOut:
    hidden_state: Tensor
    logits: Tensor


def decoder_forward(sd: SD, cfg: Cfg, input_ids: Tensor, attention_mask: Optional[Tensor] = None,
                    pos_type: str = "absolute", attn_type: Optional[str] = None, cache=None,
                    start_pos: int = 0, fused_sdpa: bool = False, with_head: bool = True) -> DecoderOut:
    """DecoderModel.forward.  models/decoder.py:324-374."""
    b, l = input_ids.shape
    h = sd["word_embeddings.weight"][input_ids]
    freqs = None
    pos = _position_info(sd, cfg, pos_type, start_pos, l)
    if pos is not None:
        h = h + pos.to(h.dtype)
    else:
        dh = cfg.hidden_size // cfg.num_attention_heads
        freqs = rotary_angles(dh, cfg.max_position_embeddings)[:, start_pos:start_pos + l]
    mask = None
    if l > 1:
        mask = decoder_additive_mask(b, l, attention_mask, start_pos, h.dtype)
    for i in range(cfg.num_hidden_layers):
        h = block(sd, f"all_layer.{i}.", cfg, h, mask, freqs, attn_type == "gqa", False, cache, i,
                  start_pos, fused_sdpa)
    logits = lm_head(sd, "lm_head.", cfg, h) if with_head else None
    return DecoderOut(h, logits)


def decoder_generate(sd: SD, cfg: Cfg, input_ids: Tensor, attention_mask: Tensor, max_len: int = 5,
                     pos_type: str = "absolute", attn_type: Optional[str] = None,
                     use_cache: bool = True, use_static_cache: bool = False,
                     pad_id: int = 1, eos_id: int = 2) -> Tensor:
    """DecoderModel.generate, greedy branch (do_sample=False).  models/decoder.py:430-514."""
    bsz, prompt = input_ids.shape
    min_prompt = max_prompt = prompt
    total = max_len + max_prompt
    tokens = torch.full((bsz, total), pad_id, dtype=torch.long)
    tokens[:, :prompt] = input_ids
    cache = None
    if use_cache:
        if use_static_cache:
            heads = cfg.num_key_value_heads if attn_type == "gqa" else cfg.num_attention_heads
            cache = OracleStaticCache(cfg.num_hidden_layers, bsz, heads, total,
                                      cfg.hidden_size // cfg.num_attention_heads)
        else:
            cache = OracleDynamicCache(cfg.num_hidden_layers)
    prev = 0
    eos = torch.zeros(bsz, dtype=torch.bool)
    text_mask = tokens != pad_id
    stop = torch.tensor(eos_id)
    for cur in range(min_prompt, total):
        out = decoder_forward(sd, cfg, tokens[:, prev:cur], attention_mask, pos_type, attn_type,
                              cache, prev)
        nxt = torch.topk(out.logits[:, -1], k=1, dim=-1)[1].reshape(-1)
        nxt = torch.where(text_mask[:, cur], tokens[:, cur], nxt)
        tokens[:, cur] = nxt
        eos |= (~text_mask[:, cur]) & torch.isin(nxt, stop)
        if use_cache:
            prev = cur
        attention_mask = torch.cat([attention_mask, torch.ones((bsz, 1))], dim=-1)
        if bool(eos.all()):
            break
    return tokens


def encoder_forward(sd: SD, cfg: Cfg, input_ids: Tensor, attention_mask: Optional[Tensor],
                    pos_type: str = "absolute", attn_type: Optional[str] = None,
                    fused_sdpa: bool = False) -> Tensor:
    """EncoderModel.forward -> last hidden state.  models/encoder.py:134-168."""
    b, l = input_ids.shape
    h = sd["word_embeddings.weight"][input_ids]
    freqs = None
    pos = _position_info(sd, cfg, pos_type, 0, l)
    if pos is not None:
        h = h + pos.to(h.dtype)
    else:
        dh = cfg.hidden_size // cfg.num_attention_heads
        freqs = rotary_angles(dh, cfg.max_position_embeddings)[:, :l]
    if attention_mask is None:
        attention_mask = torch.ones(b, l)
    mask = padding_additive_mask(attention_mask, h.dtype)
    for i in range(cfg.num_hidden_layers):
        h = block(sd, f"all_layer.{i}.", cfg, h, mask, freqs, attn_type == "gqa",
                  fused_sdpa=fused_sdpa)
    return h


def vit_forward(sd: SD, cfg, pixel_values: Tensor, fused_sdpa: bool = False) -> Tensor:
    """Vit.forward.  models/vision_encoder.py:102-145.

    Note the double positional add: VitAbsoluteEncoding.forward adds *in place* and returns the
    same tensor (layers/positional_embeddings.py:222-226), then Vit adds it to itself
    (:125-127), so the layer input is 2*(tokens + pos).
    """
    c = Cfg.of(cfg)
    ph, pw = cfg.patch_size
    x = torch.nn.functional.conv2d(pixel_values, sd["pixel_seq.weight"], sd["pixel_seq.bias"],
                                   stride=(ph, pw))
    b, d, g1, g2 = x.shape
    h = x.reshape(b, d, g1 * g2).transpose(1, 2)
    cls = sd["cls_token"].expand(b, 1, -1)
    h = torch.cat((cls, h), dim=1)
    n = h.shape[1]
    h = h + sd["position_embeddings.pos_embeddings"][:, :n + 1]
    h = h + h
    mask = padding_additive_mask(torch.ones(b, n), h.dtype)
    for i in range(cfg.num_hidden_layers):
        h = block(sd, f"all_layer.{i}.", c, h, mask, None, False, fused_qkv=True,
                  fused_sdpa=fused_sdpa)
    return h


class _PerLayerCache:
    """Adapter: per-layer caches attached to attention modules (multimodel.py:306-314)."""

    def __init__(self, inner):
        self.inner = inner

    def update(self, index, k, v, start_pos=0):
        return self.inner.update(index, k, v, start_pos)


def vlm_decoder_forward(sd: SD, cfg: Cfg, input_ids: Tensor, attention_mask: Optional[Tensor],
                        encoder_hidden_state: Tensor, pos_type="absolute", attn_type=None,
                        cache=None, start_pos: int = 0) -> Tensor:
    """VisionLanguageDecoderModel.forward -> logits.  models/multimodel.py:142-201."""
    b, _ = input_ids.shape
    h = sd["word_embeddings.weight"][input_ids]
    if start_pos == 0:
        h = torch.cat([encoder_hidden_state.unsqueeze(1), h], dim=1)
        if attention_mask is not None:
            attention_mask = torch.cat([torch.ones(b, 1, dtype=attention_mask.dtype), attention_mask], dim=1)
    l = h.shape[1]
    freqs = None
    pos = _position_info(sd, cfg, pos_type, start_pos, l)
    if pos is not None:
        h = h + pos.to(h.dtype)
    else:
        dh = cfg.hidden_size // cfg.num_attention_heads
        freqs = rotary_angles(dh, cfg.max_position_embeddings)[:, start_pos:start_pos + l]
    mask = None
    if l > 1:
        mask = decoder_additive_mask(b, l, attention_mask, start_pos, h.dtype)
    for i in range(cfg.num_hidden_layers):
        h = block(sd, f"all_layer.{i}.", cfg, h, mask, freqs, attn_type == "gqa", False, cache, i,
                  start_pos)
    return lm_head(sd, "lm_head.", cfg, h)


def generate_multimodel(sd_dec: SD, cfg: Cfg, encoder_output: Tensor, decoder_start: Tensor,
                        max_new_tokens: int, pos_type="absolute", attn_type=None,
                        cache=None) -> Tensor:
    """generation_utils.generate_multimodel, greedy.  generation_utils.py:128-197."""
    idx = decoder_start
    nxt = idx
    index = 0
    for _ in range(max_new_tokens):
        if cache is not None:
            logits = vlm_decoder_forward(sd_dec, cfg, nxt, None, encoder_output, pos_type, attn_type,
                                         cache, index)
        else:
            logits = vlm_decoder_forward(sd_dec, cfg, idx, None, encoder_output, pos_type, attn_type)
        probs = torch.softmax(logits[:, -1], dim=-1)
        nxt = torch.topk(probs, k=1, dim=-1)[1]
        idx = torch.cat((idx, nxt), dim=1)
        index = idx.shape[1]
    return idx


def generate(sd: SD, cfg: Cfg, ids: Tensor, max_new_tokens: int, pos_type="absolute",
             attn_type=None) -> Tensor:
    """generation_utils.generate without cache, greedy.  generation_utils.py:6-51."""
    idx = ids
    for _ in range(max_new_tokens):
        logits = decoder_forward(sd, cfg, idx, None, pos_type, attn_type).logits[:, -1]
        probs = torch.softmax(logits, dim=-1)
        idx = torch.cat((idx, torch.topk(probs, k=1, dim=-1)[1]), dim=1)
    return idx


# ----------------------------------------------------------------------------
# seq2seq: cross-attention + EncoderDecoderModel  (layers/attention.py:382-573,
# models/encoder_decoder.py:33-391, generation_utils.py:54-125)
# ----------------------------------------------------------------------------


class OracleCrossCache:
    """One cross-attention layer's cache: stored once, then returned unchanged
    (StaticCache.update/get/__len__ :115-166 as the cross-attention uses them)."""

    def __init__(self):
        self.kv = None

    def __len__(self):
        return 0 if self.kv is None else self.kv[0].shape[2]


def cross_attention(sd: SD, p: str, cfg: Cfg, x: Tensor, enc: Tensor, enc_mask: Optional[Tensor],
                    gqa: bool, cache: Optional[OracleCrossCache] = None) -> Tensor:
    """EncoderDecoderAttention{,Gqa}.forward: q from the decoder state, k/v from the encoder output
    (computed once when a cache is attached), no RoPE, AttentionSelfOutput with the decoder state as
    residual.  layers/attention.py:410-474 (vanilla), 512-573 (gqa)."""
    dh = cfg.hidden_size // cfg.num_attention_heads
    q = split_heads(linear(x, sd[p + "query.weight"], sd.get(p + "query.bias")), dh)
    if cache is not None and len(cache) != 0:
        k, v = cache.kv
    else:
        k = split_heads(linear(enc, sd[p + "key.weight"], sd.get(p + "key.bias")), dh)
        v = split_heads(linear(enc, sd[p + "value.weight"], sd.get(p + "value.bias")), dh)
        if cache is not None:
            cache.kv = (k, v)
    if gqa:
        n_rep = cfg.num_attention_heads // cfg.num_key_value_heads
        k, v = repeat_kv(k, n_rep), repeat_kv(v, n_rep)
    o = merge_heads(sdpa(q, k, v, enc_mask))
    return attention_self_output(sd, p + "out.", o, x, cfg.layer_norm_eps)


def seq2seq_lm_head(sd: SD, p: str, cfg: Cfg, h: Tensor) -> Tensor:
    """encoder_decoder.LMHead: dense -> GELU -> layer_norm -> vocab (bias = the tied `bias`
    parameter).  models/encoder_decoder.py:86-110."""
    x = gelu_erf(linear(h, sd[p + "dense.weight"], sd[p + "dense.bias"]))
    x = layer_norm(x, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], cfg.layer_norm_eps)
    return linear(x, sd[p + "vocab.weight"], sd[p + "bias"])


def seq2seq_decoder_forward(sd: SD, cfg: Cfg, input_ids: Tensor, attention_mask: Optional[Tensor],
                            enc: Tensor, enc_mask: Tensor, pos_type: str = "absolute",
                            attn_type: Optional[str] = None, self_cache=None, cross_caches=None,
                            start_pos: int = 0, p: str = "decoder.") -> Tensor:
    """Seq2SeqDecoderModel.forward -> hidden state.  models/encoder_decoder.py:156-212; layer :57-83:
    self-attention -> cross-attention -> FeedForward whose residual is the LAYER INPUT."""
    b, l = input_ids.shape
    h = sd[p + "word_embeddings.weight"][input_ids]
    freqs = None
    sub = {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    pos = _position_info(sub, cfg, pos_type, start_pos, l)
    if pos is not None:
        h = h + pos.to(h.dtype)
    else:
        dh = cfg.hidden_size // cfg.num_attention_heads
        freqs = rotary_angles(dh, cfg.max_position_embeddings)[:, start_pos:start_pos + l]
    mask = None
    if l > 1:
        mask = decoder_additive_mask(b, l, attention_mask, start_pos, h.dtype)
    gqa = attn_type == "gqa"
    for i in range(cfg.num_hidden_layers):
        lp = f"{p}all_layer.{i}."
        a = self_attention(sd, lp + "attention.", cfg, h, mask, freqs, gqa, False, self_cache, i, start_pos)
        c = cross_attention(sd, lp + "cross_attention.", cfg, a, enc, enc_mask, gqa,
                            None if cross_caches is None else cross_caches[i])
        h = feed_forward(sd, lp + "feed_forward.", cfg, c, h)
    return h


def encoder_decoder_forward(sd: SD, cfg_enc: Cfg, cfg_dec: Cfg, input_ids: Optional[Tensor],
                            attention_mask: Optional[Tensor], decoder_input_ids: Tensor,
                            decoder_attention_mask: Optional[Tensor] = None,
                            encoder_output: Optional[Tensor] = None, enc_pos="absolute", enc_attn=None,
                            dec_pos="absolute", dec_attn=None, self_cache=None, cross_caches=None,
                            start_pos: int = 0):
    """EncoderDecoderModel.forward -> (logits, encoder_output).  models/encoder_decoder.py:286-336."""
    if encoder_output is None:
        enc_sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
        encoder_output = encoder_forward(enc_sd, cfg_enc, input_ids, attention_mask, enc_pos, enc_attn)
    if attention_mask is None:
        attention_mask = torch.ones(encoder_output.shape[:2])
    enc_mask = padding_additive_mask(attention_mask, encoder_output.dtype)
    h = seq2seq_decoder_forward(sd, cfg_dec, decoder_input_ids, decoder_attention_mask, encoder_output,
                                enc_mask, dec_pos, dec_attn, self_cache, cross_caches, start_pos)
    return seq2seq_lm_head(sd, "lm_head.", cfg_dec, h), encoder_output


def generate_seq2seq(sd: SD, cfg_enc: Cfg, cfg_dec: Cfg, encoder_output: Tensor,
                     encoder_attention_mask: Tensor, decoder_start: Tensor, max_new_tokens: int,
                     dec_pos="absolute", dec_attn=None, use_cache: bool = False) -> Tensor:
    """generation_utils.generate_seq2seq, greedy.  generation_utils.py:54-125."""
    idx = decoder_start
    nxt = idx
    index = 0
    self_cache = OracleDynamicCache(cfg_dec.num_hidden_layers) if use_cache else None
    cross = [OracleCrossCache() for _ in range(cfg_dec.num_hidden_layers)] if use_cache else None
    for _ in range(max_new_tokens):
        if use_cache:
            logits, _ = encoder_decoder_forward(sd, cfg_enc, cfg_dec, None, encoder_attention_mask, nxt, None,
                                                encoder_output, dec_pos=dec_pos, dec_attn=dec_attn,
                                                self_cache=self_cache, cross_caches=cross, start_pos=index)
        else:
            logits, _ = encoder_decoder_forward(sd, cfg_enc, cfg_dec, None, encoder_attention_mask, idx, None,
                                                encoder_output, dec_pos=dec_pos, dec_attn=dec_attn)
        probs = torch.softmax(logits[:, -1], dim=-1)
        nxt = torch.topk(probs, k=1, dim=-1)[1]
        idx = torch.cat((idx, nxt), dim=1)
        index = idx.shape[1] - 1
    return idx


# ----------------------------------------------------------------------------
# PaliGemma-shape blocks (Examples/paligemma.ipynb cells 9, 11-13).  The notebook has no importable
# module; its class-definition cells are exec'd by tests/golden/make_golden.py to pin the layer
# blocks (tests/golden/paligemma_blocks.npz).  The cached decode loop around them is unpinned.
# ----------------------------------------------------------------------------


def gemma_mlp(x: Tensor, w_gate: Tensor, w_up: Tensor, w_down: Tensor) -> Tensor:
    """down(gelu_tanh(gate(x)) * up(x)), no biases (cell 13)."""
    return linear(gelu_tanh(linear(x, w_gate)) * linear(x, w_up), w_down)


def siglip_layer(sd: SD, p: str, x: Tensor, heads: int, eps: float) -> Tensor:
    """SiglipEncoderLayer: pre-LN, h + out_proj(attn(LN1 h)); h + fc2(gelu_tanh(fc1(LN2 h))).
    Softmax in fp32, no mask.  Examples/paligemma.ipynb cell 9."""
    d = x.shape[-1]
    dh = d // heads
    n = layer_norm(x, sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], eps)
    q = split_heads(linear(n, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]), dh)
    k = split_heads(linear(n, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"]), dh)
    v = split_heads(linear(n, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"]), dh)
    a = merge_heads(sdpa(q, k, v, None))
    x = x + linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
    n = layer_norm(x, sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], eps)
    m = gelu_tanh(linear(n, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + linear(m, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def gemma_layer(sd: SD, p: str, x: Tensor, heads: int, kv_heads: int, head_dim: int, eps: float,
                mask: Optional[Tensor], pos0: int = 0, cache=None, layer_idx: int = 0) -> Tensor:
    """GemmaDecoderLayer: pre-RMSNorm, RoPE (NeoX halves, theta 10000), MQA/GQA by repeat_kv, softmax in
    fp32, GeGLU MLP, no biases.  Examples/paligemma.ipynb cells 11-13."""
    L = x.shape[1]
    n = rms_norm_gemma(x, sd[p + "input_layernorm.weight"], eps)
    q = split_heads(linear(n, sd[p + "self_attn.q_proj.weight"]), head_dim)
    k = split_heads(linear(n, sd[p + "self_attn.k_proj.weight"]), head_dim)
    v = split_heads(linear(n, sd[p + "self_attn.v_proj.weight"]), head_dim)
    freqs = rotary_angles(head_dim, pos0 + L)[:, pos0:pos0 + L]
    q, k = apply_rotary(q, k, freqs)
    if cache is not None:
        k, v = cache.update(layer_idx, k, v, pos0)
    k, v = repeat_kv(k, heads // kv_heads), repeat_kv(v, heads // kv_heads)
    a = merge_heads(sdpa(q, k, v, mask))
    x = x + linear(a, sd[p + "self_attn.o_proj.weight"])
    n = rms_norm_gemma(x, sd[p + "post_attention_layernorm.weight"], eps)
    return x + gemma_mlp(n, sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"], sd[p + "mlp.down_proj.weight"])


def clm_loss(logits: Tensor, labels: Tensor) -> Tensor:
    """Shifted cross-entropy, ignore_index=-100 (Examples/vyom-ai-decoder_clm.ipynb cell 29)."""
    lg = logits[:, :-1].reshape(-1, logits.shape[-1]).float()
    lb = labels[:, 1:].reshape(-1)
    return torch.nn.functional.cross_entropy(lg, lb, ignore_index=-100)


# ----------------------------------------------------------------------------
# sampling processors and speculative decoding
# ----------------------------------------------------------------------------


def processor_masked_logits(logits: Tensor, top_k: int = 0, top_p: float = 0.0) -> Tensor:
    """``_process`` of Greedy/Multinomial (identity), TopK (logits_processors.py:59-63), Nucleus (:73-81)
    and TopKNucleus (:92-102) on a COPY of ``logits``: removed entries become -1e20."""
    logits = logits.clone()
    if top_k:
        k = min(top_k, logits.size(-1))
        kth = torch.topk(logits, k, dim=-1)[0][..., -1, None]
        logits[logits < kth] = -1e20
    if top_p:
        sorted_logits, sorted_indices = torch.sort(logits, descending=True)
        cumulative = torch.cumsum(torch.softmax(sorted_logits, dim=-1), dim=-1)
        remove = cumulative > top_p
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = 0
        sorted_logits[remove] = -1e20
        logits = torch.gather(sorted_logits, -1, sorted_indices.argsort(-1))
    return logits


def processor_probs(logits: Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 0.0) -> Tensor:
    """LogitsProcessor.__call__ (logits_processors.py:13-16): softmax(_process(logits) / temperature)."""
    return torch.softmax(processor_masked_logits(logits, top_k, top_p) / temperature, dim=-1)


def speculative_norm(x: Tensor) -> Tensor:
    """speculative_decoding.py:73-83."""
    x_max = torch.where(x > 0, x, torch.zeros_like(x))
    return x_max / torch.sum(x_max, dim=-1, keepdim=True)


def speculative_generate(inputs: Tensor, drafter_logits, target_logits, vocab_size: int, max_seq_length: int,
                         rand_fn, gamma: int = 5, temperature: float = 1.0, top_k: int = 0, top_p: float = 0.0,
                         max_gen_len: int = 128, eos_tokens_id=2, pad_token_id: int = 2,
                         skip_sample_adjustment: bool = False, first_target: bool = True):
    """speculative_decoding.py:86-245 with argmax sampling (GreedyProcessor.sample, logits_processors.py:35-36)
    and without a KV cache (use_cache=False: each call recomputes its prefix).  ``*_logits(ids)`` -> (1, L, V)
    logits of a model on the prefix ``ids``; ``rand_fn(n)`` -> the n acceptance draws of a round (:200)."""
    proc = lambda l: processor_probs(l, temperature, top_k, top_p)   # noqa: E731
    sample = lambda p: torch.argmax(p, dim=-1).unsqueeze(-1)          # noqa: E731
    stops = eos_tokens_id if isinstance(eos_tokens_id, list) else [eos_tokens_id]
    stop_tokens = torch.tensor(stops, dtype=torch.long).unsqueeze(1)
    accepted, speculated = .0, .0
    prompt_len = len(inputs[0])
    total_len = min(max_seq_length, prompt_len + max_gen_len)
    ids = torch.full((1, total_len), pad_token_id, dtype=torch.long)
    ids[0, :prompt_len] = inputs
    cur = prompt_len
    if first_target:   # :148-162
        t = sample(proc(target_logits(ids[..., :cur])[..., -1, :]))
        ids[0, cur] = t
        cur += 1
        if torch.isin(t, stop_tokens):
            return ids[0, prompt_len:cur].tolist(), 0
    while cur < total_len:
        g = min(gamma, total_len - cur - 1)
        q = torch.zeros((1, g, vocab_size))
        for k in range(g):   # :172-185
            dp = proc(drafter_logits(ids[..., :cur + k])[..., -1, :])
            q[0, k] = dp
            ids[0, cur + k] = sample(dp)
        speculated += g
        mp = target_logits(ids[..., :cur + g])   # :189-197
        p = proc(mp[..., cur - 1:cur + g - 1, :])
        r = rand_fn(g)
        fractions = p / q
        n = g
        for i in range(g):   # :200-206
            if r[i] > fractions[0, i, ids[0, cur + i]]:
                n = i
                break
        accepted += n
        loc = torch.nonzero(torch.eq(ids[..., cur:cur + n], stop_tokens))   # :211-216
        if loc.shape[0] > 0:
            return ids[0, prompt_len:cur + loc[0, 1].item() + 1].tolist(), accepted / speculated
        if n == g:   # :219-221
            p_p = proc(mp[..., cur + g - 1, :])
        else:        # :228-231
            p_p = p[..., n, :] if skip_sample_adjustment else speculative_norm(p[..., n, :] - q[0, n, :])
        x = sample(p_p)
        ids[0, cur + n:cur + g] = pad_token_id
        ids[0, cur + n] = x
        cur += n + 1
        if torch.isin(x, stop_tokens):
            return ids[0, prompt_len:cur].tolist(), accepted / speculated
    return ids[0, prompt_len:].tolist(), accepted / speculated
