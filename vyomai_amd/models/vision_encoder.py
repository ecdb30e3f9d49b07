"""ViT patch encoder with the reference's API (VyomAI/models/vision_encoder.py)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from ..autograd_train import defer_residual_grads as _defer
from ..layers.attention import VisionAttention, _shadow
from ..layers.ffn import FeedForward
from ..layers.mask import AttnMask
from ..layers.positional_embeddings import VitAbsoluteEncoding

_position_embeddings = {"absolute": VitAbsoluteEncoding}


@dataclass
class EncoderOutput(object):
    logits: torch.Tensor


class EncoderLayer(nn.Module):
    """VisionAttention (fused qkv) -> FeedForward(out, layer_input).  Reference :20-53."""

    def __init__(self, config, layer_idx: int, attention_type: Optional[str] = None) -> None:
        super().__init__()
        self.attention = VisionAttention(config, layer_idx=layer_idx)
        self.feed_forward = FeedForward(config)
        self.layer_idx = layer_idx

    def forward(self, hidden_state, attention_mask, freqs=None) -> torch.Tensor:
        _defer(hidden_state)  # training: its residual-path gradients are added in the qkv dgrad epilogue
        out = self.attention(hidden_state=hidden_state, attention_mask=attention_mask, freqs=freqs)
        return self.feed_forward(out, hidden_state)


class Vit(nn.Module):
    """Reference :56-153.  The stride==kernel Conv2d patchify is evaluated as what it is -- one
    GEMM over unfolded patches (vy_linear_fwd with ``pixel_seq.weight`` viewed (d, C*p*p)) --
    so the parameter keeps its Conv2d name and shape.  cls_token / pos_embeddings have width
    C*p*p, which only works when C*p*p == hidden_size, as in the reference (:89-90)."""

    def __init__(self, config, pos_embedding_type: Optional[str] = "absolute") -> None:
        super().__init__()
        self.image_size = config.image_size
        self.patch_size = config.patch_size
        self.num_channels = config.num_channels
        self.num_patches = (self.image_size[0] // self.patch_size[0]) * (self.image_size[1] // self.patch_size[1])
        cls = _position_embeddings.get(pos_embedding_type, None)
        self.position_embeddings = cls(config) if cls is not None else None
        self.all_layer = nn.ModuleList([EncoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.pixel_seq = nn.Conv2d(in_channels=self.num_channels, out_channels=config.hidden_size,
                                   kernel_size=self.patch_size, stride=self.patch_size)
        patch_dim = config.num_channels * self.patch_size[0] * self.patch_size[1]
        self.cls_token = nn.Parameter(torch.randn(1, 1, patch_dim))

    def _patchify(self, pixel_values: torch.Tensor) -> torch.Tensor:
        b, c, hh, ww = pixel_values.shape
        ph, pw = self.patch_size
        gh, gw = hh // ph, ww // pw
        # (b, c, gh, ph, gw, pw) -> (b, gh*gw, c*ph*pw): the unfold of a stride==kernel conv
        x = pixel_values.reshape(b, c, gh, ph, gw, pw).permute(0, 2, 4, 1, 3, 5).reshape(b, gh * gw, c * ph * pw)
        w = self.pixel_seq.weight
        dt = pixel_values.dtype
        if torch.is_grad_enabled() and w.requires_grad:
            from ..autograd_train import PatchifyFn
            return PatchifyFn.apply(x.contiguous(), w, self.pixel_seq.bias)
        w2 = _shadow(w, dt).reshape(w.shape[0], -1)
        return ops.linear(x.contiguous(), w2, _shadow(self.pixel_seq.bias, dt))

    def forward(self, pixel_values: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> EncoderOutput:
        hidden_state = self._patchify(pixel_values)
        bsz, seqlen, _ = hidden_state.shape
        cls_tokens = self.cls_token.to(hidden_state.dtype).expand(bsz, 1, -1)
        hidden_state = torch.cat((cls_tokens, hidden_state), dim=1)
        freqs = None
        if self.position_embeddings is not None:
            # in-place add inside, then added to itself: 2*(tokens+pos), the reference's behaviour
            pos_info = self.position_embeddings(hidden_state)
            hidden_state = hidden_state + pos_info
        else:
            raise ValueError("Vit needs pos_embedding_type='absolute' (the reference has no other table)")
        mask = None
        if attention_mask is not None:
            mask = AttnMask.from_padding(attention_mask, causal=False, start_pos=0, query_len=seqlen + 1)
        for layer in self.all_layer:
            hidden_state = layer(hidden_state, mask, freqs)
        return EncoderOutput(hidden_state)

    @classmethod
    def from_config(cls, config, pos_embedding_type: Optional[str] = "absolute") -> nn.Module:
        return cls(config, pos_embedding_type)
