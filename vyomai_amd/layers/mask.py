"""Attention-mask descriptors.

The reference materialises additive float masks -- (B,1,L,S) built by create_mask_for_decoder
(VyomAI/models/decoder.py:360-362, 376-419) or (B,1,1,L) padding masks
(VyomAI/models/encoder.py:161-164) -- and hands them to SDPA.  On MI355X the kernels take a
*descriptor* instead (causal offset + key-padding bytes) and evaluate it in registers; a dense
tensor is only built when a caller asks for one (``.dense()``), which reproduces the reference
tensor exactly.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class AttnMask:
    causal: bool = False
    start_pos: int = 0
    keypad: Optional[torch.Tensor] = None  # uint8 (B, S), 1 = keep
    query_len: int = 0
    key_len: int = 0

    @classmethod
    def from_padding(cls, attention_mask: Optional[torch.Tensor], causal: bool, start_pos: int,
                     query_len: int) -> "AttnMask":
        """attention_mask: (B, S) of 0/1 in any dtype, or None (= all ones)."""
        key_len = start_pos + query_len
        kp = None
        if attention_mask is not None:
            if attention_mask.shape[-1] != key_len and causal:
                raise ValueError(
                    f"attention_mask has {attention_mask.shape[-1]} columns, expected start_pos+seq_len={key_len}")
            key_len = attention_mask.shape[-1]
            kp = (attention_mask != 0).to(torch.uint8).contiguous()
        return cls(causal=causal, start_pos=start_pos, keypad=kp, query_len=query_len, key_len=key_len)

    def dense(self, dtype=torch.float32, batch: Optional[int] = None) -> torch.Tensor:
        """The reference's additive mask tensor: (1 - causal*keep) * finfo(dtype).min."""
        dev = self.keypad.device if self.keypad is not None else "cpu"
        L, S = self.query_len, self.key_len
        keep = torch.ones(1 if batch is None else batch, 1, L if self.causal else 1, S, device=dev)
        if self.causal:
            i = torch.arange(L, device=dev)[:, None]
            j = torch.arange(S, device=dev)[None, :]
            keep = keep * (j <= i + self.start_pos).to(keep.dtype)[None, None]
        if self.keypad is not None:
            keep = keep * self.keypad[:, None, None, :].to(keep.dtype)
        return ((1.0 - keep) * torch.finfo(dtype).min).to(dtype)
