"""KV caches with the reference's class names and methods (VyomAI/layers/kv_cache.py).

MI355X layout: every cache is a preallocated (B, kv_heads, capacity, dh) buffer per layer in HBM.
``reserve()`` hands the fused QKV kernel the [start_pos, start_pos+L) window to write K/V into
directly, and attention reads the [0, start_pos+L) prefix in place -- no torch.cat, no repeat_kv.
The dynamic caches keep the reference's grow-as-you-go contract by doubling capacity.
``update()`` (the reference entry point) is kept for callers that already hold K/V tensors.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


def _kv_heads(config, is_gqa: bool, strict: bool) -> int:
    heads = None
    if is_gqa:
        heads = getattr(config, "num_key_value_heads", None)
        if heads is None and strict:
            raise ValueError("you are using is_gqa=True and config.num_key_value_heads is not available")
    return heads if heads is not None else config.num_attention_heads


class _LayerBuf:
    """One layer's K and V buffers with capacity growth."""

    def __init__(self, growable: bool):
        self.k: Optional[torch.Tensor] = None
        self.v: Optional[torch.Tensor] = None
        self.length = 0
        self.growable = growable

    def ensure(self, B, heads, need, dh, dtype, device):
        if self.k is None:
            cap = max(need, 64) if self.growable else need
            self.k = torch.zeros(B, heads, cap, dh, dtype=dtype, device=device)
            self.v = torch.zeros_like(self.k)
        if self.k.device != torch.device(device) or self.k.dtype != dtype:
            self.k, self.v = self.k.to(device=device, dtype=dtype), self.v.to(device=device, dtype=dtype)
        if need > self.k.shape[2]:
            if not self.growable:
                raise ValueError(f"{need} tokens exceed the static cache size {self.k.shape[2]}")
            cap = max(need, 2 * self.k.shape[2])
            nk = torch.zeros(self.k.shape[0], heads, cap, dh, dtype=dtype, device=device)
            nv = torch.zeros_like(nk)
            nk[:, :, : self.length] = self.k[:, :, : self.length]
            nv[:, :, : self.length] = self.v[:, :, : self.length]
            self.k, self.v = nk, nv

    def window(self, B, start, L):
        return self.k[:B, :, start:start + L], self.v[:B, :, start:start + L]

    def prefix(self, B, end):
        return self.k[:B, :, :end], self.v[:B, :, :end]


class DynamicCache:
    """Per-layer growing cache (reference :11-78).  update() appends and returns everything."""

    def __init__(self, config, is_gqa: Optional[bool] = False) -> None:
        self._buf = _LayerBuf(growable=True)
        self._seen_tokens = False

    @property
    def key_cache(self):
        return None if self._buf.k is None else self._buf.k[:, :, : self._buf.length]

    @property
    def value_cache(self):
        return None if self._buf.v is None else self._buf.v[:, :, : self._buf.length]

    def __len__(self) -> int:
        return self._buf.length

    def reserve(self, B, heads, L, dh, start_pos, dtype, device):
        start = self._buf.length  # dynamic caches append (the reference ignores start_pos here)
        self._buf.ensure(B, heads, start + L, dh, dtype, device)
        self._pending = (B, start, L)
        return self._buf.window(B, start, L)

    def commit(self) -> Tuple[torch.Tensor, torch.Tensor]:
        B, start, L = self._pending
        self._buf.length = start + L
        self._seen_tokens = True
        return self._buf.prefix(B, start + L)

    def update(self, key_states, value_states, start_pos: int = 0):
        B, heads, L, dh = key_states.shape
        kw, vw = self.reserve(B, heads, L, dh, start_pos, key_states.dtype, key_states.device)
        kw.copy_(key_states)
        vw.copy_(value_states)
        return self.commit()

    def get(self):
        if self._seen_tokens:
            return self.key_cache, self.value_cache
        raise ValueError("there is no token available in kv-cache")

    def get_seq_length(self, layer_idx: Optional[int] = 0) -> int:
        return self._buf.length

    def get_max_length(self) -> Optional[int]:
        return None


class StaticCache:
    """Per-layer fixed cache of max_position_embeddings slots, batch 1 (reference :81-168)."""

    def __init__(self, config, is_gqa: Optional[bool] = False) -> None:
        self.head_size = int(config.hidden_size // config.num_attention_heads)
        self.heads = _kv_heads(config, bool(is_gqa), strict=True)
        self.max_len = config.max_position_embeddings
        self._buf = _LayerBuf(growable=False)
        self._seen_tokens = False
        self.first_update_len = 0

    @property
    def key_cache(self):
        return self._buf.k

    @property
    def value_cache(self):
        return self._buf.v

    def reserve(self, B, heads, L, dh, start_pos, dtype, device):
        if L > self.max_len:
            raise ValueError(f"{(B, heads, L, dh)} is more than init k_cache size {self.max_len}")
        assert B == 1, "Only support batch size 1"
        self._buf.ensure(1, self.heads, self.max_len, self.head_size, dtype, device)
        if start_pos + L > self.max_len:
            raise ValueError(f"position {start_pos + L} exceeds the static cache size {self.max_len}")
        self._pending = (B, start_pos, L)
        return self._buf.window(B, start_pos, L)

    def commit(self):
        B, start, L = self._pending
        self._seen_tokens = True
        self.first_update_len = L
        self._buf.length = start + L
        return self._buf.prefix(B, start + L)

    def update(self, k, v, start_pos: int = 0):
        B, heads, L, dh = k.shape
        kw, vw = self.reserve(B, heads, L, dh, start_pos, k.dtype, k.device)
        kw.copy_(k)
        vw.copy_(v)
        return self.commit()

    def get(self):
        if self._seen_tokens:
            return (self._buf.k[:, :, : self.first_update_len], self._buf.v[:, :, : self.first_update_len])
        raise ValueError("there is no token available in kv-cache")

    def __len__(self) -> int:
        return 0 if not self._seen_tokens else self.max_len


class DynamicCacheOne:
    """Whole-model growing cache indexed by layer (reference :171-252)."""

    def __init__(self, config, is_gqa: bool = False) -> None:
        self.layers = config.num_hidden_layers
        self._bufs = [_LayerBuf(growable=True) for _ in range(self.layers)]
        self._pending = [None] * self.layers
        self._seen_tokens = False

    @property
    def key_cache(self) -> List:
        return [[] if b.k is None else b.k[:, :, : b.length] for b in self._bufs]

    @property
    def value_cache(self) -> List:
        return [[] if b.v is None else b.v[:, :, : b.length] for b in self._bufs]

    def __len__(self) -> int:
        return self._bufs[0].length if self._bufs else 0

    def reserve(self, index, B, heads, L, dh, start_pos, dtype, device):
        b = self._bufs[index]
        start = b.length
        b.ensure(B, heads, start + L, dh, dtype, device)
        self._pending[index] = (B, start, L)
        return b.window(B, start, L)

    def commit(self, index):
        B, start, L = self._pending[index]
        self._bufs[index].length = start + L
        self._seen_tokens = True
        return self._bufs[index].prefix(B, start + L)

    def update(self, index: int, key_states, value_states, start_pos: int = 0):
        B, heads, L, dh = key_states.shape
        kw, vw = self.reserve(index, B, heads, L, dh, start_pos, key_states.dtype, key_states.device)
        kw.copy_(key_states)
        vw.copy_(value_states)
        return self.commit(index)

    def get(self, index: int):
        if self._seen_tokens:
            b = self._bufs[index]
            return b.prefix(b.k.shape[0], b.length)
        raise ValueError("there is no token available in kv-cache")

    def get_seq_length(self, layer_idx: Optional[int] = 0) -> int:
        return self._bufs[layer_idx].length

    def get_max_length(self) -> Optional[int]:
        return None


class StaticCacheOne:
    """Whole-model static cache: (B, heads, max_cache_len, dh) per layer, slice-write at start_pos
    (reference :255-377).  Like the reference it reads config.num_key_value_heads whenever the
    attribute exists, GQA or not (:275-282)."""

    def __init__(self, config, max_cache_len: int = None, dtype: torch.dtype = torch.float32,
                 batch_size: int = 1, is_gqa: bool = False) -> None:
        self.head_size = int(config.hidden_size // config.num_attention_heads)
        self.batch_size = batch_size
        self.heads = getattr(config, "num_key_value_heads", None)
        if self.heads is None:
            self.heads = config.num_attention_heads
        self.max_cache_len = config.max_position_embeddings if max_cache_len is None else max_cache_len
        self.dtype = dtype
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.cache_shape = (self.batch_size, self.heads, self.max_cache_len, self.head_size)
        self.layers = config.num_hidden_layers
        self._seen_tokens = False
        self.key_cache: List[torch.Tensor] = []
        self.value_cache: List[torch.Tensor] = []
        for _ in range(self.layers):
            self.key_cache.append(torch.zeros(self.cache_shape, dtype=self.dtype, device=self.device))
            self.value_cache.append(torch.zeros(self.cache_shape, dtype=self.dtype, device=self.device))
        self._pending = [None] * self.layers

    def __len__(self) -> int:
        return self.max_cache_len

    def reserve(self, index, B, heads, L, dh, start_pos, dtype, device):
        kc = self.key_cache[index]
        if L > kc.size(2) or start_pos + L > kc.size(2):
            raise ValueError(f"{(B, heads, L, dh)} at {start_pos} is more than init k_cache size {tuple(kc.shape)}")
        if heads != kc.size(1) or B > kc.size(0):
            raise ValueError(f"cache shape {tuple(kc.shape)} does not fit K/V of shape {(B, heads, L, dh)}")
        if kc.dtype != dtype or kc.device != torch.device(device):
            # the reference allocates fp32 on the default device; follow the model instead
            self.key_cache[index] = kc = kc.to(device=device, dtype=dtype)
            self.value_cache[index] = self.value_cache[index].to(device=device, dtype=dtype)
        self._pending[index] = (B, start_pos, L)
        return kc[:B, :, start_pos:start_pos + L], self.value_cache[index][:B, :, start_pos:start_pos + L]

    def commit(self, index):
        B, start, L = self._pending[index]
        self._seen_tokens = True
        return self.key_cache[index][:B, :, : start + L], self.value_cache[index][:B, :, : start + L]

    def update(self, index: int, key_states, value_states, start_pos: int = 0):
        B, heads, L, dh = key_states.shape
        kw, vw = self.reserve(index, B, heads, L, dh, start_pos, key_states.dtype, key_states.device)
        kw.copy_(key_states)
        vw.copy_(value_states)
        return self.commit(index)

    def get(self, index: int):
        if self._seen_tokens:
            return self.key_cache[index], self.value_cache[index]
        raise ValueError("there is no token available in kv-cache")

    def get_seq_length(self, layer_idx: Optional[int] = 0) -> int:
        return self.key_cache[layer_idx].shape[-2]

    def get_max_length(self) -> Optional[int]:
        return None
