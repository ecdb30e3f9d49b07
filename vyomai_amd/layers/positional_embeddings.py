"""Positional encodings with the reference's class names and call signatures
(VyomAI/layers/positional_embeddings.py).  RoPE is evaluated by HIP kernels -- fused into the
QKV projection epilogue on the model path, or by ``vy_rope_fwd`` through
``apply_rotary_pos_emb``; the additive encodings are table lookups kept in PyTorch."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops


class AbsoluteEncoding(nn.Module):
    """Learned position table; forward(size) -> (1, size, d).  Reference :7-51."""

    def __init__(self, config) -> None:
        super().__init__()
        self.pos_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size,
                                           padding_idx=getattr(config, "pad_token_id", None))
        self.register_buffer("position_ids",
                             torch.arange(config.max_position_embeddings).expand((1, -1)),
                             persistent=False)
        self.max_size = config.max_position_embeddings

    def forward(self, size: int) -> torch.Tensor:
        if self.max_size < size:
            raise ValueError(
                f"The hidden size ({size }) is more than the config max_position_embeddings {self.max_size}")
        return self.pos_embeddings(self.position_ids[:, :size])


class SinusoidalEncoding(nn.Module):
    """Fixed sin/cos table (1, max_pos, d), even features sin, odd cos.  Reference :54-106."""

    def __init__(self, config) -> None:
        super().__init__()
        d = config.hidden_size
        if d % 2 != 0:
            raise ValueError(f"Cannot use SinusoidalEncoding with odd hidden dim got dim {d}")
        pos = torch.arange(0, config.max_position_embeddings).unsqueeze(1).float()
        div = torch.exp(torch.arange(0, d, 2, dtype=torch.float) * -(torch.log(torch.tensor(10000.0)) / d))
        table = torch.zeros(1, config.max_position_embeddings, d)
        table[:, :, 0::2] = torch.sin(pos * div)
        table[:, :, 1::2] = torch.cos(pos * div)
        self.positional_encoding = table  # plain attribute like the reference (not in state_dict)

    def forward(self, seq_len: int) -> torch.Tensor:
        return self.positional_encoding[:, :seq_len]


class RotaryEmbedding(nn.Module):
    """forward(seq_len) -> raw angles (1, seq_len, dh/2) fp32.  Reference :109-137."""

    def __init__(self, config):
        super().__init__()
        dim = int(config.hidden_size // config.num_attention_heads)
        self.register_buffer("inv_freq", 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim)))

    def forward(self, seq_len):
        t = torch.arange(seq_len, device=self.inv_freq.device).type_as(self.inv_freq)
        return torch.einsum("i, j -> i j", t, self.inv_freq)[None, :, :]


class RopeTable:
    """cos/sin of an angle table, fp32 on the device, built once per device.  The reference
    recomputes cos/sin from the angles in every layer of every forward (:173-175); here the host
    evaluates them once in fp32 with the same torch CPU ops and uploads them."""

    def __init__(self, angles: torch.Tensor):
        self.angles = angles[0] if angles.dim() == 3 else angles  # (P, dh/2) on the host
        self._dev: Dict[torch.device, Tuple[torch.Tensor, torch.Tensor]] = {}

    def on(self, device) -> Tuple[torch.Tensor, torch.Tensor]:
        device = torch.device(device)
        if device not in self._dev:
            a = self.angles.float().cpu()
            self._dev[device] = (a.cos().contiguous().to(device), a.sin().contiguous().to(device))
        return self._dev[device]


class RopeSlice:
    """What the MI355X models pass as ``freqs``: a table plus the position window."""

    def __init__(self, table: RopeTable, pos0: int, length: int):
        self.table, self.pos0, self.length = table, pos0, length


def resolve_freqs(freqs, device):
    """freqs: None | RopeSlice | raw angle tensor (1, L, dh/2) as in the reference API.
    -> (cos, sin, pos0) or (None, None, 0)."""
    if freqs is None:
        return None, None, 0
    if isinstance(freqs, RopeSlice):
        cos, sin = freqs.table.on(device)
        return cos, sin, freqs.pos0
    a = freqs[0] if freqs.dim() == 3 else freqs
    a = a.float().cpu()  # evaluate like the reference: fp32 on the host
    return a.cos().contiguous().to(device), a.sin().contiguous().to(device), 0


def apply_rotary_pos_emb(q, k, freqs, unsqueeze_dim=1) -> Tuple[torch.Tensor, torch.Tensor]:
    """q, k: (B, heads, L, dh); freqs: raw angles (1, L, dh/2).  Returns rotated copies
    (reference :155-182), computed by the vy_rope_fwd HIP kernel."""
    if unsqueeze_dim != 1:
        raise NotImplementedError("vyomai_amd RoPE kernels use the (B, heads, L, dh) layout")
    cos, sin, pos0 = resolve_freqs(freqs, q.device)
    qo = ops.rope_(q.contiguous().clone(), cos, sin, pos0)
    ko = ops.rope_(k.contiguous().clone(), cos, sin, pos0)
    return qo, ko


class VitAbsoluteEncoding(nn.Module):
    """Learned (1, patches+1, C*p*p) table.  forward(img_seq) adds IN PLACE and returns the same
    tensor, exactly like the reference (:222-226) -- Vit.forward then adds the result to itself,
    so the encoder input is 2*(tokens + pos); reproduced on purpose."""

    def __init__(self, config) -> None:
        super().__init__()
        ih, iw = config.image_size
        ph, pw = config.patch_size
        assert ih % ph == 0 and iw % pw == 0, "Image dimensions must be divisible by the patch size."
        n = (ih // ph) * (iw // pw)
        self.pos_embeddings = nn.Parameter(torch.randn(1, n + 1, config.num_channels * ph * pw))
        self.register_buffer("num_patches", torch.arange(n + 1).expand((1, -1)), persistent=False)

    def forward(self, img_seq: torch.Tensor) -> torch.Tensor:
        n = img_seq.shape[1]
        img_seq += self.pos_embeddings[:, : (n + 1)].to(img_seq.dtype)
        return img_seq
