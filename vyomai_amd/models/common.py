"""Pieces shared by the model files: LM head, position-embedding registry, RoPE window."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from .._lib import ACT_GELU_ERF
from ..layers.attention import _shadow
from ..layers.positional_embeddings import (AbsoluteEncoding, RopeSlice, RopeTable, RotaryEmbedding,
                                            SinusoidalEncoding)

POSITION_EMBEDDINGS = {"absolute": AbsoluteEncoding, "sinusoidal": SinusoidalEncoding}


class LMHead(nn.Module):
    """decoder(LN(gelu(dense(h)))) with the vocabulary bias tied to ``decoder.bias``
    (reference VyomAI/models/decoder.py:253-275, VyomAI/models/encoder.py:67-89).
    Two MFMA GEMM launches (GELU fused in the first) and one LayerNorm launch; the logits buffer
    has its row stride padded to 8 elements because vocab_size is odd."""

    def __init__(self, config) -> None:
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.layer_norm = nn.LayerNorm(config.hidden_size, eps=getattr(config, "layer_norm_eps", 1e-6))
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))
        self.decoder.bias = self.bias

    def forward(self, hidden_state: torch.Tensor) -> torch.Tensor:
        from ..autograd import _wants_grad
        if _wants_grad(hidden_state, self.dense.weight, self.decoder.weight):
            from ..autograd_train import LMHeadFn
            return LMHeadFn.apply(hidden_state, self.dense.weight, self.dense.bias, self.layer_norm.weight,
                                  self.layer_norm.bias, self.decoder.weight, self.bias, self.layer_norm.eps)
        dt = hidden_state.dtype
        x = ops.linear(hidden_state, _shadow(self.dense.weight, dt), _shadow(self.dense.bias, dt), act=ACT_GELU_ERF)
        x, _, _ = ops.layernorm(x, _shadow(self.layer_norm.weight, dt), _shadow(self.layer_norm.bias, dt),
                                self.layer_norm.eps)
        return ops.linear(x, _shadow(self.decoder.weight, dt), _shadow(self.bias, dt))


    def loss(self, hidden_state: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100,
             check_labels: bool = False) -> torch.Tensor:
        """Shifted CLM cross-entropy fused with the head (no fp32 logits copy): the training-side
        entry point; see autograd_train.LMHeadLossFn.  A label that is neither `ignore_index` nor a
        vocabulary id (torch.cross_entropy device-asserts on it) is never dereferenced: its row counts as
        ignored and the device flag `self.label_error` is raised -- `check_labels=True` (or
        `raise_on_label_error()` whenever convenient: it synchronises) turns that into a ValueError."""
        from ..autograd_train import LMHeadLossFn
        flag = getattr(self, "label_error", None)
        if flag is None or flag.device != hidden_state.device:
            flag = self.label_error = torch.zeros(1, dtype=torch.int32, device=hidden_state.device)
        out = LMHeadLossFn.apply(hidden_state, labels, ignore_index, self.dense.weight, self.dense.bias,
                                 self.layer_norm.weight, self.layer_norm.bias, self.decoder.weight, self.bias,
                                 self.layer_norm.eps, flag)
        if check_labels:
            self.raise_on_label_error()
        return out

    def raise_on_label_error(self) -> None:
        flag = getattr(self, "label_error", None)
        if flag is not None and int(flag.item()) != 0:
            flag.zero_()
            raise ValueError(f"labels contain ids outside [0, {self.decoder.weight.shape[0]}) other than "
                             "ignore_index (those rows were treated as ignored)")


class PositionMixin:
    """Builds position embeddings / the RoPE angle table exactly like the reference constructors
    (models/decoder.py:294-304) and serves per-forward windows."""

    def _init_positions(self, config, pos_embedding_type: Optional[str], who: str) -> None:
        cls = POSITION_EMBEDDINGS.get(pos_embedding_type, None)
        self.position_embeddings = cls(config) if cls is not None else None
        if pos_embedding_type == "rope":
            # plain tensor attribute, not a buffer, as in the reference (not in state_dict)
            self.emb_freq = RotaryEmbedding(config)(config.max_position_embeddings)
            self._rope_table = RopeTable(self.emb_freq)
            print(f"{who} Ignoring sinusoidal or absolute position embeddings because rope,is enable")

    # fp32 master weights with bf16 kernels: set by FlatTrainer (None = the parameters' dtype)
    compute_dtype = None

    def _embed(self, emb: nn.Embedding, input_ids: torch.Tensor) -> torch.Tensor:
        """Token embedding through vy_embedding_fwd/bwd (in the compute dtype when a trainer set one)."""
        w = emb.weight
        if not w.is_cuda:
            return emb(input_ids)  # host-side use only (state-dict tooling); kernels need the GPU
        from .. import ops
        from ..layers.attention import _shadow
        dt = self.compute_dtype or w.dtype
        if torch.is_grad_enabled() and w.requires_grad:
            from ..autograd_train import EmbeddingFn
            return EmbeddingFn.apply(input_ids, w, emb.padding_idx, dt)
        return ops.embedding(_shadow(w, dt), input_ids)

    def _cast(self, hidden_state: torch.Tensor) -> torch.Tensor:
        cd = self.compute_dtype
        return hidden_state if cd is None or hidden_state.dtype == cd else hidden_state.to(cd)

    def _positions(self, hidden_state: torch.Tensor, start_pos: int, seqlen: int):
        """-> (hidden_state [+ position info] in the compute dtype, freqs)."""
        if self.position_embeddings is not None:
            pos = self.position_embeddings(start_pos + seqlen)[:, start_pos:start_pos + seqlen, :]
            return self._cast(hidden_state + pos.to(device=hidden_state.device, dtype=hidden_state.dtype)), None
        hidden_state = self._cast(hidden_state)
        if start_pos + seqlen > self.emb_freq.shape[1]:
            raise ValueError(f"position {start_pos + seqlen} exceeds max_position_embeddings {self.emb_freq.shape[1]}")
        return hidden_state, RopeSlice(self._rope_table, start_pos, seqlen)
