"""PaliGemma-shaped model (SigLIP vision tower + projector + Gemma decoder) on the HIP kernels.

The reference ships this model only as notebook cells (Examples/paligemma.ipynb cells 9, 11-13,
15-17); class and parameter names below follow those cells so a checkpoint laid out like the
notebook's modules loads with ``load_state_dict``.  Same kernels as the VyomAI layers, different
wiring: pre-norm residual blocks, LayerNorm (SigLIP) / RMSNorm with (1+w) (Gemma), GELU-tanh MLP /
GeGLU, grouped-query attention down to a single KV head, head_dim 72 (SigLIP) and 256 (Gemma), RoPE
with theta 10000, embeddings scaled by sqrt(d), LM head tied to the embedding table.

Inference only (config 5 of BASELINE.json is KV-cache decode).
"""
from __future__ import annotations

import os

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from .._lib import ACT_GELU_TANH
from ..layers.positional_embeddings import RopeTable


def _packed(mods: List[nn.Linear], cache_owner: nn.Module, key: str):
    """[W0; W1; ...] (and biases) of sibling projections as one matrix, cached per parameter version."""
    ver = tuple(m.weight._version for m in mods) + tuple(m.weight.data_ptr() for m in mods)
    hit = getattr(cache_owner, key, None)
    if hit is None or hit[0] != ver:
        w = torch.cat([m.weight.detach() for m in mods], dim=0).contiguous()
        b = None
        if mods[0].bias is not None:
            b = torch.cat([m.bias.detach() for m in mods], dim=0).contiguous()
        hit = (ver, w, b)
        setattr(cache_owner, key, hit)
    return hit[1], hit[2]


class SiglipVisionConfig:
    def __init__(self, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                 num_channels=3, image_size=224, patch_size=16, layer_norm_eps=1e-6, attention_dropout=0.0,
                 num_image_tokens: int = None, **kwargs):
        self.hidden_size, self.intermediate_size = hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.num_channels, self.patch_size, self.image_size = num_channels, patch_size, image_size
        self.attention_dropout, self.layer_norm_eps = attention_dropout, layer_norm_eps
        self.num_image_tokens = num_image_tokens


class SiglipAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.embed_dim, self.num_heads = config.hidden_size, config.num_attention_heads
        self.head_dim = self.embed_dim // self.num_heads
        self.k_proj = nn.Linear(self.embed_dim, self.embed_dim)
        self.v_proj = nn.Linear(self.embed_dim, self.embed_dim)
        self.q_proj = nn.Linear(self.embed_dim, self.embed_dim)
        self.out_proj = nn.Linear(self.embed_dim, self.embed_dim)


class SiglipMLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.fc1 = nn.Linear(config.hidden_size, config.intermediate_size)
        self.fc2 = nn.Linear(config.intermediate_size, config.hidden_size)


class SiglipEncoderLayer(nn.Module):
    """h + out_proj(attn(LN1 h)); h + fc2(gelu_tanh(fc1(LN2 h))).  Notebook cell 9."""

    def __init__(self, config: SiglipVisionConfig):
        super().__init__()
        self.embed_dim = config.hidden_size
        self.self_attn = SiglipAttention(config)
        self.layer_norm1 = nn.LayerNorm(self.embed_dim, eps=config.layer_norm_eps)
        self.mlp = SiglipMLP(config)
        self.layer_norm2 = nn.LayerNorm(self.embed_dim, eps=config.layer_norm_eps)

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        a = self.self_attn
        B, L, d = hidden_states.shape
        h, dh = a.num_heads, a.head_dim
        n, _, _ = ops.layernorm(hidden_states, self.layer_norm1.weight, self.layer_norm1.bias, self.layer_norm1.eps)
        w, b = _packed([a.q_proj, a.k_proj, a.v_proj], self, "_qkv")
        q = torch.empty((B, h, L, dh), dtype=n.dtype, device=n.device)
        k, v = torch.empty_like(q), torch.empty_like(q)
        ops.qkv_rope(n, w, b, h, h, dh, None, None, 0, q, k, v)
        o = ops.attention(q, k, v)
        x = ops.linear(o, a.out_proj.weight, a.out_proj.bias, residual=hidden_states)
        n, _, _ = ops.layernorm(x, self.layer_norm2.weight, self.layer_norm2.bias, self.layer_norm2.eps)
        m = ops.linear(n, self.mlp.fc1.weight, self.mlp.fc1.bias, act=ACT_GELU_TANH)
        return ops.linear(m, self.mlp.fc2.weight, self.mlp.fc2.bias, residual=x)


class SiglipVisionTransformer(nn.Module):
    """Patch embedding (stride == kernel conv as one GEMM) + learned positions + N layers + LN."""

    def __init__(self, config: SiglipVisionConfig):
        super().__init__()
        self.config = config
        d, p = config.hidden_size, config.patch_size
        self.patch_embedding = nn.Conv2d(config.num_channels, d, kernel_size=p, stride=p, padding="valid")
        self.num_patches = (config.image_size // p) ** 2
        self.position_embedding = nn.Embedding(self.num_patches, d)
        self.layers = nn.ModuleList([SiglipEncoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.post_layernorm = nn.LayerNorm(d, eps=config.layer_norm_eps)

    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        b, c, hh, ww = pixel_values.shape
        p = self.config.patch_size
        gh, gw = hh // p, ww // p
        x = pixel_values.reshape(b, c, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(b, gh * gw, c * p * p)
        kdim = c * p * p
        kpad = (kdim + 7) // 8 * 8  # the GEMM wants 16-byte rows (3*14*14 = 588 is fine; keep general)
        w = self.patch_embedding.weight.reshape(self.patch_embedding.weight.shape[0], -1)
        if kpad != kdim:
            x = torch.nn.functional.pad(x, (0, kpad - kdim))
            w = torch.nn.functional.pad(w, (0, kpad - kdim))
        hidden = ops.linear(x.contiguous(), w.contiguous(), self.patch_embedding.bias)
        hidden = hidden + self.position_embedding.weight[None, : gh * gw].to(hidden.dtype)
        for layer in self.layers:
            hidden = layer(hidden)
        y, _, _ = ops.layernorm(hidden, self.post_layernorm.weight, self.post_layernorm.bias, self.post_layernorm.eps)
        return y


class GemmaRMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.zeros(dim))

    def forward(self, x):
        return ops.rmsnorm(x, self.weight, self.eps, 1.0)


class GemmaMLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.gate_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.up_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.down_proj = nn.Linear(config.intermediate_size, config.hidden_size, bias=False)

    def forward(self, x, residual=None):
        w, _ = _packed([self.gate_proj, self.up_proj], self, "_gu")
        gu = ops.linear(x, w)                       # one GEMM for [gate | up]
        act = ops.gated_act(gu, ACT_GELU_TANH)      # gelu_tanh(gate) * up
        return ops.linear(act, self.down_proj.weight, residual=residual)


class GemmaAttention(nn.Module):
    def __init__(self, config, layer_idx: Optional[int] = None):
        super().__init__()
        self.layer_idx = layer_idx
        self.hidden_size, self.num_heads, self.head_dim = config.hidden_size, config.num_attention_heads, config.head_dim
        self.num_key_value_heads = config.num_key_value_heads
        if self.hidden_size % self.num_heads != 0:
            raise ValueError(f"hidden_size must be divisible by num_heads (got `hidden_size`: {self.hidden_size}"
                             f" and `num_heads`: {self.num_heads}).")
        bias = config.attention_bias
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=bias)
        self.k_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=bias)
        self.v_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=bias)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, self.hidden_size, bias=bias)


class GemmaDecoderLayer(nn.Module):
    """Pre-RMSNorm block with RoPE + GQA/MQA attention and a GeGLU MLP.  Notebook cells 11-13.
    `cache` is a (k, v) pair of (B, hk, cap, dh) buffers written in place at `start_pos`."""

    def __init__(self, config, layer_idx: int):
        super().__init__()
        self.self_attn = GemmaAttention(config=config, layer_idx=layer_idx)
        self.mlp = GemmaMLP(config)
        self.input_layernorm = GemmaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = GemmaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, hidden_states: torch.Tensor, rope: Optional[RopeTable], start_pos: int = 0,
                causal: bool = False, cache: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                pos_offset: int = 0) -> torch.Tensor:
        """start_pos: the cache slot of the first token; the rotary position of token t is
        start_pos + t + pos_offset (PaliGemma counts positions from 1: notebook cell 17, position_ids)."""
        a = self.self_attn
        B, L, _ = hidden_states.shape
        h, hk, dh = a.num_heads, a.num_key_value_heads, a.head_dim
        n = self.input_layernorm(hidden_states)
        w, b = _packed([a.q_proj, a.k_proj, a.v_proj], self, "_qkv")
        cos, sin = rope.on(n.device) if rope is not None else (None, None)
        q = torch.empty((B, h, L, dh), dtype=n.dtype, device=n.device)
        if cache is not None:
            kc, vc = cache
            kw, vw = kc[:B, :, start_pos:start_pos + L], vc[:B, :, start_pos:start_pos + L]
        else:
            kw = torch.empty((B, hk, L, dh), dtype=n.dtype, device=n.device)
            vw = torch.empty_like(kw)
        ops.qkv_rope(n, w, b, h, hk, dh, cos, sin, start_pos + pos_offset, q, kw, vw)
        if cache is not None:
            k_all, v_all = kc[:B, :, : start_pos + L], vc[:B, :, : start_pos + L]
        else:
            k_all, v_all = kw, vw
        if L == 1:
            o = ops.attention_decode(q, k_all, v_all, k_all.shape[2])
        else:
            o = ops.attention(q, k_all, v_all, causal=causal, start_pos=start_pos)
        x = ops.linear(o, a.o_proj.weight, a.o_proj.bias, residual=hidden_states)
        n = self.post_attention_layernorm(x)
        return self.mlp(n, residual=x)


@dataclass
class PaliGemmaShape:
    vision: SiglipVisionConfig
    text: object            # hidden_size, intermediate_size, num_hidden_layers, num_attention_heads, head_dim, ...
    projection_dim: int


class PaliGemmaForConditionalGeneration(nn.Module):
    """vision_tower -> multi_modal_projector -> [image tokens | text tokens] -> Gemma -> tied LM head.
    Prefill attends bidirectionally over the whole prefix (the notebook's inference mask,
    cell 17 `_update_causal_mask` with is_training=False); generated tokens attend to everything
    before them."""

    def __init__(self, shape: PaliGemmaShape):
        super().__init__()
        t = shape.text
        self.shape = shape
        self.vision_tower = SiglipVisionTransformer(shape.vision)
        self.multi_modal_projector = nn.Linear(shape.vision.hidden_size, shape.projection_dim, bias=True)
        self.embed_tokens = nn.Embedding(t.vocab_size, t.hidden_size)
        self.layers = nn.ModuleList([GemmaDecoderLayer(t, i) for i in range(t.num_hidden_layers)])
        self.norm = GemmaRMSNorm(t.hidden_size, eps=t.rms_norm_eps)
        self.rope = RopeTable(_angles(t.head_dim, t.max_position_embeddings, t.rope_theta))

    # PaliGemma's rotary positions are 1-indexed: position_ids = cache_position + 1 (notebook cell 17)
    ROPE_OFFSET = 1

    # our parameter names -> the notebook model's (cells 9, 15-17), for checkpoints laid out like its modules
    _REFERENCE_NAMES = (("vision_tower.patch_embedding.", "vision_tower.vision_model.embeddings.patch_embedding."),
                        ("vision_tower.position_embedding.", "vision_tower.vision_model.embeddings.position_embedding."),
                        ("vision_tower.layers.", "vision_tower.vision_model.encoder.layers."),
                        ("vision_tower.post_layernorm.", "vision_tower.vision_model.post_layernorm."),
                        ("multi_modal_projector.", "multi_modal_projector.linear."),
                        ("embed_tokens.", "language_model.model.embed_tokens."),
                        ("layers.", "language_model.model.layers."),
                        ("norm.", "language_model.model.norm."))

    @classmethod
    def reference_name(cls, name: str) -> str:
        """The notebook model's state_dict key of one of this module's parameters."""
        for ours, theirs in cls._REFERENCE_NAMES:
            if name.startswith(ours):
                return theirs + name[len(ours):]
        return name

    def load_reference_state_dict(self, state_dict) -> None:
        """Load a state_dict keyed like the notebook's PaliGemmaForConditionalGeneration (the tied
        language_model.lm_head.weight, if present, is the embedding table)."""
        own = self.state_dict()
        missing = [n for n in own if self.reference_name(n) not in state_dict]
        if missing:
            raise KeyError(f"reference state_dict lacks {missing[:4]}{' ...' if len(missing) > 4 else ''}")
        with torch.no_grad():
            for n, t in own.items():
                t.copy_(state_dict[self.reference_name(n)])

    def lm_head(self, hidden: torch.Tensor) -> torch.Tensor:
        return ops.linear(hidden, self.embed_tokens.weight)  # tied (GemmaForCausalLM.tie_weights)

    def _embed_scale(self, like: torch.Tensor) -> torch.Tensor:
        """sqrt(d) as a tensor of the activations' dtype (the notebook's `normalizer`, cell 13: rounded to bf16 BEFORE the
        multiply) -- kept on the device, so that no host-to-device copy sits inside a prefill (a hipGraph cannot hold one)."""
        c = getattr(self, "_scale_cache", None)
        if c is None or c.dtype != like.dtype or c.device != like.device:
            c = self._scale_cache = torch.tensor(self.shape.text.hidden_size ** 0.5, dtype=like.dtype, device=like.device)
        return c

    def _decoder(self, hidden, start_pos, caches):
        hidden = hidden * self._embed_scale(hidden)
        for i, layer in enumerate(self.layers):
            hidden = layer(hidden, self.rope, start_pos, causal=False, cache=None if caches is None else caches[i],
                           pos_offset=self.ROPE_OFFSET)
        return self.norm(hidden)

    def image_features(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """projector(vision tower) / sqrt(d) (notebook cell 17, get_image_features: the decoder multiplies every
        input embedding by sqrt(d), image rows included)."""
        dt = self.embed_tokens.weight.dtype
        img = ops.linear(self.vision_tower(pixel_values.to(dt)), self.multi_modal_projector.weight,
                         self.multi_modal_projector.bias)
        return img / (self.shape.text.hidden_size ** 0.5)

    @torch.no_grad()
    def prefill(self, pixel_values: torch.Tensor, input_ids: torch.Tensor, caches=None) -> torch.Tensor:
        """Hidden states (after the final norm) of the prefix [image tokens | text tokens]; K/V go into `caches`."""
        hidden = torch.cat([self.image_features(pixel_values), self.embed_tokens(input_ids)], dim=1)
        return self._decoder(hidden, 0, caches)

    # ---- generation state: static caches, the captured prefill, the decode plan ----------------------------------
    def _weights_key(self):
        return tuple(p._version for p in self.parameters()) + (self.embed_tokens.weight.data_ptr(),)

    def _generation_state(self, pixel_values, input_ids, max_cache_len: int):
        """Everything of generate() that does not depend on the VALUES of the inputs, built once per (batch, prompt length,
        cache length) and kept while the weights are untouched: the K/V buffers (slots past the prefix are written before
        they are read, so they need no clearing between calls), the decode plan over them (its RMSNorm-folded weight copies
        alone cost ~5 ms to build) and the prefill as ONE hipGraph -- vision tower, projector, the 264-row language-model
        pass and the vocabulary product of the last position are ~450 launches that the host needs ~9 ms to enqueue one
        by one, longer than the GPU needs to run them (VY_PREFILL_GRAPH=0: eager launches)."""
        t = self.shape.text
        dev, dt = pixel_values.device, self.embed_tokens.weight.dtype
        B, T = input_ids.shape
        key = (B, T, tuple(pixel_values.shape), max_cache_len, dt, dev)
        st = getattr(self, "_gen_state", None)
        wkey = self._weights_key()
        if st is not None and st["key"] == key and st["wkey"] == wkey:
            return st
        st = {"key": key, "wkey": wkey, "graph": None, "plan": None}
        st["caches"] = [(torch.zeros(B, t.num_key_value_heads, max_cache_len, t.head_dim, dtype=dt, device=dev),
                         torch.zeros(B, t.num_key_value_heads, max_cache_len, t.head_dim, dtype=dt, device=dev))
                        for _ in self.layers]
        st["no_stop"] = torch.tensor([-1], dtype=torch.long, device=dev)
        self._gen_state = st
        return st

    def _prefill_first_logits(self, pixel_values, input_ids, caches) -> torch.Tensor:
        hidden = torch.cat([self.image_features(pixel_values), self.embed_tokens(input_ids)], dim=1)
        out = self._decoder(hidden, 0, caches)
        return self.lm_head(out[:, -1:, :])[:, -1]

    def _run_prefill(self, st, pixel_values, input_ids) -> torch.Tensor:
        if os.environ.get("VY_PREFILL_GRAPH", "1") == "0":
            return self._prefill_first_logits(pixel_values, input_ids, st["caches"])
        if st["graph"] is None:
            st["px"], st["ids"] = pixel_values.clone(), input_ids.clone()
            # eager once (packs the sibling projections, warms the allocator), then record
            self._prefill_first_logits(st["px"], st["ids"], st["caches"])
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st["first"] = self._prefill_first_logits(st["px"], st["ids"], st["caches"])
            st["graph"] = g
        st["px"].copy_(pixel_values)
        st["ids"].copy_(input_ids)
        st["graph"].replay()
        return st["first"]

    @torch.no_grad()
    def generate(self, pixel_values: torch.Tensor, input_ids: torch.Tensor, max_new_tokens: int = 64,
                 max_cache_len: int = 384) -> torch.Tensor:
        t = self.shape.text
        dev = pixel_values.device
        B = input_ids.shape[0]
        n_img = self.vision_tower.num_patches
        pos = n_img + input_ids.shape[1]
        if pos + max_new_tokens - 1 > max_cache_len:
            raise ValueError(f"{pos} prefix tokens + {max_new_tokens} new tokens exceed max_cache_len={max_cache_len}")
        st = self._generation_state(pixel_values, input_ids, max_cache_len)
        first = self._run_prefill(st, pixel_values, input_ids)
        # greedy loop: the first token comes from the prefill; every later one is ONE native call through the
        # stack (vy_gemma_decoder_step) plus one for the pick (vy_greedy_step) -- no per-layer Python
        tokens = torch.zeros((B, max_new_tokens), dtype=torch.long, device=dev)
        done = torch.zeros(B, dtype=torch.bool, device=dev)       # (the notebook's loop has no early stop)
        no_stop = st["no_stop"]
        ops.greedy_step_(first, tokens, 0, None, no_stop, done)
        if max_new_tokens > 1:
            if st["plan"] is None:
                from ..decode_plan import GemmaDecodePlan
                st["plan"] = GemmaDecodePlan(self, st["caches"], B)
            plan = st["plan"]
            scale = self._embed_scale(self.embed_tokens.weight)
            for i in range(1, max_new_tokens):
                x = self.embed_tokens(tokens[:, i - 1]) * scale
                ops.greedy_step_(plan.step(x.contiguous(), pos), tokens, i, None, no_stop, done)
                pos += 1
        return tokens


def _angles(dim: int, max_pos: int, base: float) -> torch.Tensor:
    inv = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
    return torch.outer(torch.arange(max_pos).float(), inv)[None]
