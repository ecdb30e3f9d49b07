"""FeedForward with the reference's signature and parameter names (VyomAI/layers/ffn.py:18-40):
LN(dropout(out(act(intermediate(x)))) + input_tensor), width multiplier*hidden (NOT
config.intermediate_size, like the reference :19-23).

MI355X: two MFMA GEMM launches -- bias+GELU in the first epilogue, bias+dropout+residual in the
second -- and one LayerNorm launch."""
from __future__ import annotations

from typing import Union

import torch
import torch.nn as nn

from .._lib import ACT_GELU_ERF, ACT_GELU_TANH

# activations with a fused HIP epilogue; the reference's table (:7-15) also lists leaky_relu,
# relu6, sigmoid, silu/swish, tanh, which no shipped model or test selects.
_FUSED_ACT = {"gelu": ACT_GELU_ERF, "gelu_tanh": ACT_GELU_TANH, "gelu_pytorch_tanh": ACT_GELU_TANH}
_REFERENCE_ACTS = {"gelu", "leaky_relu", "relu6", "sigmoid", "silu", "swish", "tanh"}


class FeedForward(nn.Module):
    def __init__(self, config, multiplier: Union[int, float] = 4) -> None:
        super().__init__()
        inner = int(multiplier) * config.hidden_size
        self.intermediate = nn.Linear(config.hidden_size, inner)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.layernorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        name = getattr(config, "hidden_act", None)
        if name in _FUSED_ACT:
            self.act = _FUSED_ACT[name]
        elif name in _REFERENCE_ACTS:
            raise NotImplementedError(f"hidden_act={name!r} has no fused MI355X epilogue yet (gelu only)")
        else:
            self.act = ACT_GELU_ERF  # unknown names fall back to GELU like the reference (:26-29)
        self.out = nn.Linear(inner, config.hidden_size)

    def forward(self, hidden_state: torch.Tensor, input_tensor: torch.Tensor) -> torch.Tensor:
        from ..autograd import ffn_block
        return ffn_block(hidden_state, input_tensor, self.intermediate.weight, self.intermediate.bias,
                         self.out.weight, self.out.bias, self.layernorm.weight, self.layernorm.bias,
                         self.layernorm.eps, self.act, self.dropout.p, self.training)
