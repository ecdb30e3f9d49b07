"""Config + small helpers with the reference's names (VyomAI/utils.py:9-40, 89-100)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import torch.nn as nn


@dataclass
class EncoderConfig:
    """Same fields and defaults as the reference dataclass (VyomAI/utils.py:89-100); any
    duck-typed object with these attributes works (the reference tests define their own)."""

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


def model_size(model: nn.Module) -> float:
    """Parameters + buffers in MiB."""
    n = sum(p.nelement() * p.element_size() for p in model.parameters())
    n += sum(b.nelement() * b.element_size() for b in model.buffers())
    return n / 1024 ** 2


def model_parameters(model: nn.Module) -> Dict[str, int]:
    ps = list(model.parameters())
    return {"total_params": sum(p.numel() for p in ps),
            "trainable_params": sum(p.numel() for p in ps if p.requires_grad)}


def init_weights(module: nn.Module) -> None:
    """N(0, 0.02) linear/embedding weights, zero biases, unit LayerNorm."""
    if isinstance(module, nn.Linear):
        module.weight.data.normal_(mean=0.0, std=0.02)
        if module.bias is not None:
            module.bias.data.zero_()
    elif isinstance(module, nn.Embedding):
        module.weight.data.normal_(mean=0.0, std=0.02)
        if module.padding_idx is not None:
            module.weight.data[module.padding_idx].zero_()
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)
