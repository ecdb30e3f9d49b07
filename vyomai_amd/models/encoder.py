"""BERT-style encoder with the reference's API (VyomAI/models/encoder.py)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from ..layers.attention import EncoderAttention, EncoderAttentionGqa
from ..layers.ffn import FeedForward
from ..layers.mask import AttnMask
from ..autograd_train import defer_residual_grads as _defer
from .common import LMHead, PositionMixin


@dataclass
class EncoderOutput(object):
    logits: torch.Tensor  # holds the last hidden state, as in the reference (:168)


@dataclass
class MLMOutput(object):
    hidden_state: torch.Tensor
    logits: torch.Tensor


class EncoderLayer(nn.Module):
    """attention -> feed_forward(out, layer_input).  Reference :30-64."""

    def __init__(self, config, layer_idx: int, attention_type: str = None) -> None:
        super().__init__()
        self.attention = (EncoderAttentionGqa(config, layer_idx=layer_idx) if attention_type == "gqa"
                          else EncoderAttention(config, layer_idx=layer_idx))
        if attention_type == "gqa" and layer_idx == 0:
            print("Encoder Using GQA Attention")
        self.feed_forward = FeedForward(config)
        self.layer_idx = layer_idx

    def forward(self, hidden_state, attention_mask, freqs=None) -> torch.Tensor:
        _defer(hidden_state)  # training: its residual-path gradients are added in the QKV dgrad epilogue
        out = self.attention(hidden_state=hidden_state, attention_mask=attention_mask, freqs=freqs)
        return self.feed_forward(out, hidden_state)


class EncoderModel(nn.Module, PositionMixin):
    """Reference :92-177.  forward(input_ids, attention_mask) -> EncoderOutput(hidden states)."""

    def __init__(self, config, pos_embedding_type: Optional[str] = "absolute", attention_type: str = None) -> None:
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size,
                                            padding_idx=getattr(config, "pad_token_id", None))
        self._init_positions(config, pos_embedding_type, "Encoder")
        self.all_layer = nn.ModuleList(
            [EncoderLayer(config, i, attention_type) for i in range(config.num_hidden_layers)])

    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> EncoderOutput:
        _, seqlen = input_ids.shape
        hidden_state = self._embed(self.word_embeddings, input_ids)
        hidden_state, freqs = self._positions(hidden_state, 0, seqlen)
        # the reference builds (1-mask)*finfo.min of shape (B,1,1,L) (:161-164); same information
        # as a key-padding descriptor
        mask = AttnMask.from_padding(attention_mask, causal=False, start_pos=0, query_len=seqlen)
        if mask.keypad is None:
            mask = None
        for layer in self.all_layer:
            hidden_state = layer(hidden_state, mask, freqs)
        return EncoderOutput(hidden_state)

    @classmethod
    def from_config(cls, config, pos_embedding_type: Optional[str] = "absolute", attention_type: str = None):
        return cls(config, pos_embedding_type, attention_type)


class EncoderForMaskedLM(nn.Module):
    """Reference :180-217."""

    def __init__(self, config, pos_embedding_type: Optional[str] = "absolute", attention_type: str = None) -> None:
        super().__init__()
        self.encoder = EncoderModel(config, pos_embedding_type=pos_embedding_type, attention_type=attention_type)
        self.lm_head = LMHead(config=config)

    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> MLMOutput:
        out = self.encoder(input_ids=input_ids, attention_mask=attention_mask)
        return MLMOutput(hidden_state=out.logits, logits=self.lm_head(out.logits))

    @classmethod
    def from_config(cls, config, pos_embedding_type: Optional[str] = "absolute", attention_type: str = None):
        return cls(config, pos_embedding_type, attention_type)
