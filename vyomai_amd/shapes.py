"""Parameter name -> shape tables of the reference models' state_dicts and the
configuration objects of the benchmark workloads.  Product-side so that bench.py and tools/ do not import the
test tree; tests/golden/cases.py re-exports everything here."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

@dataclass
class VitCfg:
    hidden_size: int = 768
    num_attention_heads: int = 12
    image_size: Tuple[int, int] = (224, 224)
    patch_size: Tuple[int, int] = (16, 16)
    num_channels: int = 3
    num_hidden_layers: int = 4
    hidden_dropout_prob: float = 0.1
    initializer_range: float = 0.02
    intermediate_size: int = 3072
    layer_norm_eps: float = 1e-05
    hidden_act: str = "gelu"


def vit_b16_config(num_hidden_layers: int = 12, hidden_dropout_prob: float = 0.0) -> VitCfg:
    """ViT-B/16 of BASELINE.json configs[3]: 224x224 images, 16x16 patches, d = 768, 12 heads, 197 tokens."""
    return VitCfg(num_hidden_layers=num_hidden_layers, hidden_dropout_prob=hidden_dropout_prob)


# ---------------------------------------------------------------------------
# parameter name -> shape tables (reference state_dict layout)
# ---------------------------------------------------------------------------


def attn_shapes(cfg, kind: str):
    """kind in {'vanilla','gqa','vision'}; names relative to the attention module."""
    d = cfg.hidden_size
    dh = d // cfg.num_attention_heads
    s = {}
    if kind == "vision":
        s["qkv.weight"], s["qkv.bias"] = (3 * d, d), (3 * d,)
    else:
        kv = d if kind == "vanilla" else getattr(cfg, "num_key_value_heads", 4) * dh
        s["query.weight"], s["query.bias"] = (d, d), (d,)
        s["key.weight"], s["key.bias"] = (kv, d), (kv,)
        s["value.weight"], s["value.bias"] = (kv, d), (kv,)
    s.update(aso_shapes(cfg, "out."))
    return s


def aso_shapes(cfg, p=""):
    d = cfg.hidden_size
    return {p + "dense.weight": (d, d), p + "dense.bias": (d,),
            p + "layernorm.weight": (d,), p + "layernorm.bias": (d,)}


def ffn_shapes(cfg, p=""):
    d = cfg.hidden_size
    return {p + "intermediate.weight": (4 * d, d), p + "intermediate.bias": (4 * d,),
            p + "layernorm.weight": (d,), p + "layernorm.bias": (d,),
            p + "out.weight": (d, 4 * d), p + "out.bias": (d,)}


def layer_shapes(cfg, kind: str, p=""):
    s = {p + "attention." + k: v for k, v in attn_shapes(cfg, kind).items()}
    s.update(ffn_shapes(cfg, p + "feed_forward."))
    return s


def lm_head_shapes(cfg, p="lm_head."):
    d, v = cfg.hidden_size, cfg.vocab_size
    return {p + "bias": (v,), p + "dense.weight": (d, d), p + "dense.bias": (d,),
            p + "layer_norm.weight": (d,), p + "layer_norm.bias": (d,),
            p + "decoder.weight": (v, d), p + "decoder.bias": (v,)}


def text_model_shapes(cfg, pos: str, attn_type, head: bool, p=""):
    d = cfg.hidden_size
    s = {p + "word_embeddings.weight": (cfg.vocab_size, d)}
    if pos == "absolute":
        s[p + "position_embeddings.pos_embeddings.weight"] = (cfg.max_position_embeddings, d)
    kind = "gqa" if attn_type == "gqa" else "vanilla"
    for i in range(cfg.num_hidden_layers):
        s.update(layer_shapes(cfg, kind, f"{p}all_layer.{i}."))
    if head:
        s.update(lm_head_shapes(cfg, p + "lm_head."))
    return s


def s2s_layer_shapes(cfg, kind: str, p=""):
    """Seq2SeqDecoderLayer: self-attention, cross-attention (same parameter names), FeedForward."""
    s = {p + "attention." + k: v for k, v in attn_shapes(cfg, kind).items()}
    s.update({p + "cross_attention." + k: v for k, v in attn_shapes(cfg, kind).items()})
    s.update(ffn_shapes(cfg, p + "feed_forward."))
    return s


def s2s_model_shapes(cfg, pos: str, attn_type):
    """EncoderDecoderModel state_dict (models/encoder_decoder.py:261-284): encoder.*, decoder.*,
    lm_head.{dense,layer_norm,vocab}.* and the tied lm_head.bias."""
    d, v = cfg.hidden_size, cfg.vocab_size
    kind = "gqa" if attn_type == "gqa" else "vanilla"
    s = text_model_shapes(cfg, pos, attn_type, head=False, p="encoder.")
    s["decoder.word_embeddings.weight"] = (v, d)
    if pos == "absolute":
        s["decoder.position_embeddings.pos_embeddings.weight"] = (cfg.max_position_embeddings, d)
    for i in range(cfg.num_hidden_layers):
        s.update(s2s_layer_shapes(cfg, kind, f"decoder.all_layer.{i}."))
    s.update({"lm_head.bias": (v,), "lm_head.dense.weight": (d, d), "lm_head.dense.bias": (d,),
              "lm_head.layer_norm.weight": (d,), "lm_head.layer_norm.bias": (d,),
              "lm_head.vocab.weight": (v, d), "lm_head.vocab.bias": (v,)})
    return s


def vit_shapes(cfg, p=""):
    d = cfg.hidden_size
    ph, pw = cfg.patch_size
    n = (cfg.image_size[0] // ph) * (cfg.image_size[1] // pw)
    pd = cfg.num_channels * ph * pw
    s = {p + "cls_token": (1, 1, pd), p + "position_embeddings.pos_embeddings": (1, n + 1, pd)}
    for i in range(cfg.num_hidden_layers):
        s.update(layer_shapes(cfg, "vision", f"{p}all_layer.{i}."))
    s[p + "pixel_seq.weight"] = (d, cfg.num_channels, ph, pw)
    s[p + "pixel_seq.bias"] = (d,)
    return s


# PaliGemma shapes (module dump in Examples/paligemma.ipynb cell 24 output)
SIGLIP = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16,
              num_channels=3, image_size=224, patch_size=14, layer_norm_eps=1e-6, attention_dropout=0.0)
GEMMA = dict(hidden_size=2048, intermediate_size=16384, num_hidden_layers=18, num_attention_heads=8,
             head_dim=256, num_key_value_heads=1, rms_norm_eps=1e-6, attention_bias=False,
             attention_dropout=0.0, max_position_embeddings=8192, rope_theta=10000.0, vocab_size=257216,
             pad_token_id=0)


def siglip_layer_shapes(p=""):
    d, i = SIGLIP["hidden_size"], SIGLIP["intermediate_size"]
    s = {}
    for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
        s[f"{p}self_attn.{n}.weight"], s[f"{p}self_attn.{n}.bias"] = (d, d), (d,)
    for n in ("layer_norm1", "layer_norm2"):
        s[f"{p}{n}.weight"], s[f"{p}{n}.bias"] = (d,), (d,)
    s[f"{p}mlp.fc1.weight"], s[f"{p}mlp.fc1.bias"] = (i, d), (i,)
    s[f"{p}mlp.fc2.weight"], s[f"{p}mlp.fc2.bias"] = (d, i), (d,)
    return s


def gemma_layer_shapes(p=""):
    d, i = GEMMA["hidden_size"], GEMMA["intermediate_size"]
    h, hk, dh = GEMMA["num_attention_heads"], GEMMA["num_key_value_heads"], GEMMA["head_dim"]
    return {f"{p}self_attn.q_proj.weight": (h * dh, d), f"{p}self_attn.k_proj.weight": (hk * dh, d),
            f"{p}self_attn.v_proj.weight": (hk * dh, d), f"{p}self_attn.o_proj.weight": (d, h * dh),
            f"{p}mlp.gate_proj.weight": (i, d), f"{p}mlp.up_proj.weight": (i, d), f"{p}mlp.down_proj.weight": (d, i),
            f"{p}input_layernorm.weight": (d,), f"{p}post_attention_layernorm.weight": (d,)}


