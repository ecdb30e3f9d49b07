"""Case definitions shared by the golden generator (reference side) and the tests
(oracle / HIP side): configs, masks and sub-sampling.  numpy only."""
from __future__ import annotations

import copy
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

FMIN = float(np.finfo(np.float32).min)

# (batch, seq) per module-level tag; 'wide' uses L=17 (not a tile multiple, as in the
# reference tests' 17-token rows)
MODULE_BL = {"micro": (2, 16), "wide": (2, 17)}


@dataclass
class TextCfg:
    hidden_size: int = 768
    num_attention_heads: int = 12
    max_position_embeddings: int = 514
    num_hidden_layers: int = 4
    vocab_size: int = 50265
    hidden_dropout_prob: float = 0.1
    initializer_range: float = 0.02
    intermediate_size: int = 3072
    layer_norm_eps: float = 1e-05
    hidden_act: str = "gelu"


@dataclass
class GqaCfg(TextCfg):
    num_key_value_heads: int = 4


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


def test_cfg() -> TextCfg:
    """The Config of the reference's tests (tests/test_decoder.py:12-23)."""
    return TextCfg()


test_cfg.__test__ = False  # not a pytest test


def micro_cfg() -> GqaCfg:
    return GqaCfg(hidden_size=64, num_attention_heads=4, max_position_embeddings=64,
                  num_hidden_layers=1, vocab_size=97, num_key_value_heads=2)


def wide_cfg() -> GqaCfg:
    return GqaCfg(num_hidden_layers=1, num_key_value_heads=4)


def vit_cfg() -> VitCfg:
    return VitCfg()


def with_kv(cfg, attn_type: Optional[str]):
    """Config for a vanilla / gqa model.  The reference's StaticCacheOne reads
    config.num_key_value_heads even for vanilla attention (layers/kv_cache.py:275-282), so
    vanilla configs must not carry the attribute."""
    base = {k: getattr(cfg, k) for k in TextCfg.__dataclass_fields__ if hasattr(cfg, k)}
    if attn_type == "gqa":
        return GqaCfg(**base, num_key_value_heads=getattr(cfg, "num_key_value_heads", 4))
    return TextCfg(**base)


def one_layer(cfg, gqa: bool):
    c = with_kv(cfg, "gqa" if gqa else None)
    c.num_hidden_layers = 1
    return c


def keypad(batch: int, seq: int) -> np.ndarray:
    """(B, L) int64 0/1 right-padding mask; row b keeps seq - (3*b + 2) % (seq//2) tokens, row 0 full."""
    m = np.ones((batch, seq), dtype=np.int64)
    for b in range(1, batch):
        keep = seq - ((3 * b + 2) % max(seq // 2, 1)) - 1
        m[b, keep:] = 0
    return m


def causal_additive(batch: int, seq: int, start_pos: int, keypad_full: Optional[np.ndarray]) -> np.ndarray:
    """Dense (B,1,L,start+L) additive fp32 mask as DecoderModel builds it
    (reference models/decoder.py:360-362, 376-419), restated in numpy."""
    total = seq + start_pos
    am = np.ones((batch, total), dtype=np.float32) if keypad_full is None else keypad_full.astype(np.float32)
    i = np.arange(seq)[:, None]
    j = np.arange(total)[None, :]
    causal = (j <= i + start_pos).astype(np.float32)
    ext = causal[None, None] * am[:, None, None, :]
    return ((1.0 - ext) * FMIN).astype(np.float32)


def reference_test_inputs():
    """Token rows / masks hard-coded in the reference tests (tests/test_decoder.py:28-46)."""
    ids = np.array([
        [0, 2387, 766, 16, 181, 967, 46035, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1],
        [0, 12196, 16, 110, 766, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1],
        [0, 37111, 1137, 162, 110, 766, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1]], dtype=np.int64)
    am = np.array([
        [1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0],
        [1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
        [1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]], dtype=np.int64)
    return ids, am


def sub(y):
    """Sub-sample a (B, L, D) activation: every 8th position (offset 1) x every 4th feature."""
    return y[:, 1::8, ::4]


def sub2(g):
    """Sub-sample a 2-D gradient: rows ::7, cols ::5."""
    return g[::7, ::5]


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


# ---- sampling processors / speculative decoding ----------------------------------

def sampling_logits() -> np.ndarray:
    """(3, 1531) fp32 logits: row 0 flat-ish, row 1 peaked, row 2 wide."""
    from vyomai_amd import recipe
    x = recipe.uniform("samp.logits", (3, 1531))
    return (x * np.array([[1.5], [9.0], [4.0]], dtype=np.float32)).astype(np.float32)


# name -> (reference class, constructor args)
PROCESSORS = {
    "greedy": ("GreedyProcessor", (1,)),
    "multinomial_t07": ("MultinomialProcessor", (0.7,)),
    "topk50_t08": ("TopKProcessor", (0.8, 50)),
    "topk_all": ("TopKProcessor", (1.0, 5000)),
    "nucleus09_t02": ("NucleusProcessor", (0.2, 0.9)),
    "nucleus05_t1": ("NucleusProcessor", (1.0, 0.5)),
    "topk40_nucleus08_t09": ("TopKNucleusProcessor", (0.9, 40, 0.8)),
}


def speculative_draws() -> np.ndarray:
    """The uniform [0,1) numbers the acceptance test consumes, in order."""
    from vyomai_amd import recipe
    return (recipe.uniform("spec.rand", (512,)) * 0.5 + 0.5).astype(np.float32)


SPECULATIVE = {
    # a small drafter against a deeper target: accepts and rejects mixed
    "mixed": dict(target_layers=3, drafter_layers=1, drafter_prefix="spec.drafter.", prompt_len=6, gamma=4,
                  max_gen_len=18, eos=2, skip=False, first_target=True, processor=("GreedyProcessor", (1,))),
    # the drafter IS the target (same weights): every draft is accepted
    "same": dict(target_layers=2, drafter_layers=2, drafter_prefix="spec.target.", prompt_len=5, gamma=3,
                 max_gen_len=11, eos=2, skip=False, first_target=True, processor=("GreedyProcessor", (1,))),
    # no target prefill, no sample adjustment, temperature != 1, two stop tokens
    "plain": dict(target_layers=2, drafter_layers=1, drafter_prefix="spec.drafter.", prompt_len=7, gamma=5,
                  max_gen_len=14, eos=[2, 7], skip=True, first_target=False, processor=("GreedyProcessor", (0.7,))),
}
