"""Token loops with the reference's signatures (VyomAI/generation_utils.py).  Host control flow
only: each step calls the model forward (HIP kernels) and picks the next token."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops


def _pick(logits: torch.Tensor, temperature: float, do_sample: bool) -> torch.Tensor:
    """softmax(logits / temperature) (vy_sampling_probs), then a draw or the most probable token."""
    probs = ops.sampling_probs(logits, temperature)
    if do_sample:
        return torch.multinomial(probs, num_samples=1)
    return probs.argmax(dim=-1, keepdim=True)


@torch.no_grad()
def generate(model: nn.Module, tokenize_text: torch.Tensor, max_new_tokens: Optional[int] = 3,
             temperature: Optional[float] = 1.0, do_sample: Optional[bool] = False,
             use_cache: Optional[bool] = False) -> torch.Tensor:
    """Reference :6-51: re-feeds the whole sequence (no cache) or only the newest token with
    ``start_pos`` (cache attached to the model)."""
    idx = nxt = tokenize_text
    index = 0
    first = True
    for _ in range(max_new_tokens):
        if not use_cache:
            logits = model(input_ids=idx).logits
        else:
            logits = model(input_ids=nxt if not first else idx, start_pos=index, use_cache=use_cache).logits
        first = False
        nxt = _pick(logits[:, -1], temperature, do_sample)
        idx = torch.cat((idx, nxt), dim=1)
        index = idx.size(1) - 1
    return idx


@torch.no_grad()
def generate_multimodel(model: nn.Module, encoder_output: torch.Tensor, encoder_attention_mask: torch.Tensor,
                        decoder_start: torch.Tensor, max_new_tokens=24, temperature=1.0, do_sample=False,
                        top_k=10, use_cache=False) -> torch.Tensor:
    """Reference :128-197.  ``index`` counts the image token too (:195)."""
    idx = nxt = decoder_start
    index = 0
    for _ in range(max_new_tokens):
        if use_cache:
            logits = model(encoder_output=encoder_output, decoder_input_ids=nxt, use_cache=use_cache,
                           start_pos=index).logits
        else:
            logits = model(encoder_output=encoder_output, decoder_input_ids=idx).logits
        nxt = _pick(logits[:, -1], temperature, do_sample)
        idx = torch.cat((idx, nxt), dim=1)
        index = idx.size(1)
    return idx


@torch.no_grad()
def generate_seq2seq(model: nn.Module, encoder_output: torch.Tensor, encoder_attention_mask: torch.Tensor,
                     decoder_start: torch.Tensor, max_new_tokens: Optional[int] = 5,
                     temperature: Optional[float] = 1.0, do_sample: Optional[bool] = False,
                     top_k: Optional[int] = 10, use_cache: Optional[bool] = False) -> torch.Tensor:
    """Reference :54-125.  With a cache the first call feeds the whole decoder start (start_pos 0),
    later calls only the newest token with start_pos = tokens already cached."""
    idx = nxt = decoder_start
    index = 0
    for _ in range(max_new_tokens):
        if use_cache:
            logits = model(encoder_output=encoder_output, attention_mask=encoder_attention_mask,
                           decoder_input_ids=nxt, use_cache=use_cache, start_pos=index).logits
        else:
            logits = model(encoder_output=encoder_output, attention_mask=encoder_attention_mask,
                           decoder_input_ids=idx, use_cache=use_cache).logits
        nxt = _pick(logits[:, -1], temperature, do_sample)
        idx = torch.cat((idx, nxt), dim=1)
        index = idx.size(1) - 1
    return idx

