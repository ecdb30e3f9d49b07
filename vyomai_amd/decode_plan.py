"""ctypes mirror of vy_decode_plan / vy_decode_layer (include/vyom_hip.h) and the builder that
turns a DecoderModel + StaticCacheOne into one.  One `vy_decoder_step` call then runs a whole
single-token step; the Python side only does the embedding lookup and the token pick."""
from __future__ import annotations

import ctypes as C
import os
from typing import List

import torch

from . import _lib
from .layers.attention import _shadow


class VyDecodeLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_w", "ln1_b", "w1", "b1", "w2", "b2",
                                          "ln2_w", "ln2_b", "kcache", "vcache")] + \
               [(n, C.c_int64) for n in ("c_sb", "c_sh", "c_sl")]


class VyDecodePlan(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_layers", "B", "d", "h", "hk", "dh", "ffn", "vocab", "act", "dtype")] + \
               [(n, C.c_float) for n in ("eps_attn", "eps_ffn", "eps_head")] + \
               [("cos_tab", C.c_void_p), ("sin_tab", C.c_void_p), ("layers", C.POINTER(VyDecodeLayer))] + \
               [(n, C.c_void_p) for n in ("head_wd", "head_bd", "head_ln_w", "head_ln_b", "head_wv", "head_bias")] + \
               [("ws", C.c_void_p), ("ws_bytes", C.c_int64)]


def _ptr(t):
    return None if t is None else t.data_ptr()


class DecodePlan:
    """Pointers of every layer of `model` (in the compute dtype) and of `cache` (StaticCacheOne)."""

    def __init__(self, model, cache, batch: int, dtype: torch.dtype, device):
        lib = _lib.load()
        cfg = model.config
        layers = list(model.all_layer)
        att0 = layers[0].attention
        h, hk, dh = att0.num_attention_heads, att0.num_key_value_heads, att0.head_dim
        d = cfg.hidden_size
        ffn = layers[0].feed_forward.intermediate.weight.shape[0]
        self.keep: List[torch.Tensor] = []  # everything the plan points at stays alive with it

        def sh(t):
            s = _shadow(t, dtype)
            if s is not None:
                if s.device != torch.device(device):
                    raise _lib.VyomHipError("model and cache must live on the same GPU")
                self.keep.append(s)
            return _ptr(s)

        arr = (VyDecodeLayer * len(layers))()
        for i, layer in enumerate(layers):
            att, ff = layer.attention, layer.feed_forward
            pw, pb = att._packed_shadow(dtype)
            for t_ in (pw, pb):
                if t_ is not None:
                    self.keep.append(t_)
            kc, vc = cache.key_cache[i], cache.value_cache[i]
            if kc.dtype != dtype or kc.device != torch.device(device):
                cache.key_cache[i] = kc = kc.to(device=device, dtype=dtype)
                cache.value_cache[i] = vc = vc.to(device=device, dtype=dtype)
            if kc.shape[1] != hk or kc.shape[0] < batch or kc.stride(3) != 1:
                raise ValueError(f"cache shape {tuple(kc.shape)} does not fit (B={batch}, kv heads={hk})")
            L = arr[i]
            L.wqkv, L.bqkv = _ptr(pw), _ptr(pb)
            L.wo, L.bo = sh(att.out.dense.weight), sh(att.out.dense.bias)
            L.ln1_w, L.ln1_b = sh(att.out.layernorm.weight), sh(att.out.layernorm.bias)
            L.w1, L.b1 = sh(ff.intermediate.weight), sh(ff.intermediate.bias)
            L.w2, L.b2 = sh(ff.out.weight), sh(ff.out.bias)
            L.ln2_w, L.ln2_b = sh(ff.layernorm.weight), sh(ff.layernorm.bias)
            L.kcache, L.vcache = kc.data_ptr(), vc.data_ptr()
            L.c_sb, L.c_sh, L.c_sl = kc.stride(0), kc.stride(1), kc.stride(2)
            self.keep += [kc, vc]
        self.layers = arr
        plan = VyDecodePlan()
        plan.num_layers, plan.B, plan.d, plan.h, plan.hk, plan.dh = len(layers), batch, d, h, hk, dh
        plan.ffn, plan.vocab, plan.act = ffn, cfg.vocab_size, layers[0].feed_forward.act
        plan.dtype = _lib.dtype_code(dtype)
        plan.eps_attn = att0.out.layernorm.eps
        plan.eps_ffn = layers[0].feed_forward.layernorm.eps
        head = model.lm_head
        plan.eps_head = head.layer_norm.eps
        if getattr(model, "_rope_table", None) is not None and model.position_embeddings is None:
            cos, sin = model._rope_table.on(device)
            self.keep += [cos, sin]
            plan.cos_tab, plan.sin_tab = cos.data_ptr(), sin.data_ptr()
        plan.layers = C.cast(arr, C.POINTER(VyDecodeLayer))
        plan.head_wd, plan.head_bd = sh(head.dense.weight), sh(head.dense.bias)
        plan.head_ln_w, plan.head_ln_b = sh(head.layer_norm.weight), sh(head.layer_norm.bias)
        plan.head_wv, plan.head_bias = sh(head.decoder.weight), sh(head.bias)
        nbytes = lib.vy_decode_ws_bytes(batch, d, h, hk, dh, ffn, plan.dtype)
        self.ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)   # (the fused kernels' tickets start at 0)
        plan.ws, plan.ws_bytes = self.ws.data_ptr(), nbytes
        self.plan = plan
        self.batch, self.d, self.vocab, self.dtype, self.device = batch, d, cfg.vocab_size, dtype, device
        self.ldv = (cfg.vocab_size + 7) // 8 * 8
        self.cache = cache
        self.capacity = min(int(cache.key_cache[i].shape[2]) for i in range(len(layers)))

    # ---- hipGraph replay -------------------------------------------------------------------
    def graph_capable(self) -> bool:
        """Device-side positions need the fused-RoPE QKV path (bf16, head_dim 64) or no RoPE."""
        # opt-in: at B=32 the step is bound by the ~86 short kernels themselves, not by launch gaps
        # (measured: replay 1.05 ms vs eager 0.96 ms per token step), so eager is the default
        if os.environ.get("VY_DECODE_GRAPH", "0") != "1":
            return False
        return (not self.plan.cos_tab) or (self.dtype == torch.bfloat16 and self.plan.dh == 64)

    def _launch(self, x_ptr, pos, pos_dev_ptr, hidden_ptr, logits_ptr):
        _lib.call("vy_decoder_step", C.byref(self.plan), x_ptr, pos, pos_dev_ptr, hidden_ptr, logits_ptr,
                  self.ldv, torch.cuda.current_stream().cuda_stream)

    def _capture(self, x: torch.Tensor, pos: int):
        """Run this step eagerly on a side stream (also the warm-up) and record the same launches
        into a hipGraph that reads the position from device memory; later steps only replay."""
        self.x_static = torch.empty_like(x)
        self.logits_static = torch.empty((self.batch, self.ldv), dtype=self.dtype, device=self.device)
        self.pos_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.x_static.copy_(x)
        self.pos_dev.fill_(pos)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._launch(self.x_static.data_ptr(), pos, self.pos_dev.data_ptr(), None, self.logits_static.data_ptr())
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._launch(self.x_static.data_ptr(), pos, self.pos_dev.data_ptr(), None, self.logits_static.data_ptr())

    def step(self, x: torch.Tensor, pos: int, want_hidden: bool = False):
        """x: (B, d) embeddings of the current token -> (logits (B, V) view, hidden (B, d) | None)."""
        assert x.shape == (self.batch, self.d) and x.dtype == self.dtype and x.is_contiguous()
        if not 0 <= pos < self.capacity:   # the kernels write K/V at `pos` unchecked
            raise ValueError(f"position {pos} exceeds the static cache size {self.capacity}")
        self.cache._seen_tokens = True
        if not want_hidden and self.graph_capable():
            if getattr(self, "graph", None) is None:
                self._capture(x, pos)  # the eager run inside produced this step's logits
            else:
                self.x_static.copy_(x)
                self.pos_dev.fill_(pos)
                self.graph.replay()
            return self.logits_static[:, : self.vocab], None
        logits = torch.empty((self.batch, self.ldv), dtype=self.dtype, device=self.device)
        hidden = torch.empty((self.batch, self.d), dtype=self.dtype, device=self.device) if want_hidden else None
        self._launch(x.data_ptr(), pos, None, _ptr(hidden), logits.data_ptr())
        return logits[:, : self.vocab], hidden


# ---- Gemma-style stack (PaliGemma-shape language model, BASELINE configs[4]) ---------------------------

class VyGemmaLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wqkv", "bqkv", "wo", "bo", "ln_in", "ln_post", "wgu", "wdown", "kcache",
                                          "vcache")] + [(n, C.c_int64) for n in ("c_sb", "c_sh", "c_sl")]


class VyGemmaPlan(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_layers", "B", "d", "h", "hk", "dh", "ffn", "vocab", "dtype")] + \
               [("eps", C.c_float), ("cos_tab", C.c_void_p), ("sin_tab", C.c_void_p),
                ("layers", C.POINTER(VyGemmaLayer)), ("norm_w", C.c_void_p), ("head_w", C.c_void_p),
                ("ws", C.c_void_p), ("ws_bytes", C.c_int64), ("flags", C.c_int32)]

GEMMA_PRESCALED = 1


class GemmaDecodePlan:
    """vy_gemma_plan of a models.paligemma.PaliGemmaForConditionalGeneration and its per-layer (k, v) cache
    buffers: `step(x, pos)` runs one token through all layers, the final norm and the tied vocabulary
    projection in ONE C call (vy_gemma_decoder_step)."""

    def __init__(self, model, caches, batch: int):
        from .models.paligemma import _packed
        lib = _lib.load()
        t = model.shape.text
        dt = model.embed_tokens.weight.dtype
        dev = model.embed_tokens.weight.device
        self.keep = []

        def ptr(x):
            if x is None:
                return None
            x = x.detach()
            assert x.dtype == dt and x.is_contiguous() and x.device == dev
            self.keep.append(x)
            return x.data_ptr()

        layers = list(model.layers)
        arr = (VyGemmaLayer * len(layers))()
        # single-sequence bf16 decode: RMSNorm folded into the weights (VY_GEMMA_PRESCALE=0: separate RMSNorm launches)
        # -- only where the step will run the chain that skips them (the same predicate as vy_gemma_decoder_step's `fused`;
        # the C side refuses a pre-scaled plan on any other chain)
        widths_ok = t.hidden_size % 8 == 0 and t.intermediate_size % 8 == 0 and (t.num_attention_heads * t.head_dim) % 8 == 0
        prescale = (dt == torch.bfloat16 and batch <= 4 and widths_ok and os.environ.get("VY_GEMMA_PRESCALE", "1") != "0"
                    and os.environ.get("VY_GEMMA_FUSED", "1") != "0")
        for i, layer in enumerate(layers):
            a, m = layer.self_attn, layer.mlp
            wqkv, bqkv = _packed([a.q_proj, a.k_proj, a.v_proj], layer, "_qkv")
            wgu, _ = _packed([m.gate_proj, m.up_proj], m, "_gu")
            if prescale:
                # the norms' (1 + w) folded into the K axis of the weights that follow them (done once; the step then
                # needs no RMSNorm launch: each product scales itself by the row's rsqrt(mean x^2 + eps))
                with torch.no_grad():
                    wqkv = (wqkv.float() * (1.0 + layer.input_layernorm.weight.detach().float())[None, :]).to(dt).contiguous()
                    wgu = (wgu.float() * (1.0 + layer.post_attention_layernorm.weight.detach().float())[None, :]).to(dt).contiguous()
            kc, vc = caches[i]
            if kc.dtype != dt or kc.shape[0] < batch or kc.stride(3) != 1 or kc.stride() != vc.stride():
                raise ValueError(f"cache buffers {tuple(kc.shape)} do not fit (B={batch})")
            L = arr[i]
            L.wqkv, L.bqkv = ptr(wqkv), ptr(bqkv)
            L.wo, L.bo = ptr(a.o_proj.weight), ptr(a.o_proj.bias)
            L.ln_in, L.ln_post = ptr(layer.input_layernorm.weight), ptr(layer.post_attention_layernorm.weight)
            L.wgu, L.wdown = ptr(wgu), ptr(m.down_proj.weight)
            L.kcache, L.vcache = kc.data_ptr(), vc.data_ptr()
            L.c_sb, L.c_sh, L.c_sl = kc.stride(0), kc.stride(1), kc.stride(2)
            self.keep += [kc, vc]
        self.layers = arr
        p = VyGemmaPlan()
        p.num_layers, p.B, p.d = len(layers), batch, t.hidden_size
        p.h, p.hk, p.dh = t.num_attention_heads, t.num_key_value_heads, t.head_dim
        p.ffn, p.vocab, p.dtype, p.eps = t.intermediate_size, t.vocab_size, _lib.dtype_code(dt), t.rms_norm_eps
        cos, sin = model.rope.on(dev)
        self.keep += [cos, sin]
        # rotary positions are cache slot + ROPE_OFFSET (PaliGemma counts from 1): the step indexes the tables
        # by cache slot, so hand it the tables shifted by that many rows
        off = int(getattr(model, "ROPE_OFFSET", 0)) * cos.shape[-1] * cos.element_size()
        assert cos.is_contiguous() and sin.is_contiguous() and cos.dim() >= 2
        p.cos_tab, p.sin_tab = cos.data_ptr() + off, sin.data_ptr() + off
        p.layers = C.cast(arr, C.POINTER(VyGemmaLayer))
        p.norm_w, p.head_w = ptr(model.norm.weight), ptr(model.embed_tokens.weight)
        lib.vy_gemma_ws_bytes.restype = C.c_int64
        lib.vy_gemma_ws_bytes.argtypes = [C.c_int32] * 6
        nbytes = lib.vy_gemma_ws_bytes(batch, p.d, p.h, p.dh, p.ffn, p.dtype)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        p.ws, p.ws_bytes = self.ws.data_ptr(), nbytes
        p.flags = GEMMA_PRESCALED if prescale else 0
        self.plan, self.batch, self.d, self.dtype, self.device = p, batch, p.d, dt, dev
        self.ldv = (t.vocab_size + 7) // 8 * 8
        self.vocab = t.vocab_size
        self.capacity = min(int(kc.shape[2]) for kc, _ in caches)

    def step(self, x: torch.Tensor, pos: int) -> torch.Tensor:
        """x: (B, d) scaled embeddings of the current token -> logits (B, vocab) view."""
        assert x.shape == (self.batch, self.d) and x.dtype == self.dtype and x.is_contiguous()
        if not 0 <= pos < self.capacity:   # the kernels write K/V at `pos` unchecked
            raise ValueError(f"position {pos} exceeds the cache size {self.capacity} (max_cache_len)")
        logits = torch.empty((self.batch, self.ldv), dtype=self.dtype, device=self.device)
        _lib.call("vy_gemma_decoder_step", C.byref(self.plan), x.data_ptr(), int(pos), logits.data_ptr(), self.ldv,
                  torch.cuda.current_stream().cuda_stream)
        return logits[:, : self.vocab]
