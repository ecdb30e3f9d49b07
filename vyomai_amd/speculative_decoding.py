"""Speculative decoding (Leviathan et al., arXiv 2211.17192) -- the reference's ``speculative_generate``
(VyomAI/speculative_decoding.py:86-245) over vyomai_amd ``DecoderModel``s.

The reference drives HF-style models (``model(input_ids=, past_key_values=, use_cache=)`` with a
``transformers`` cache); here drafter and target are the native decoders and the KV caches are
``DynamicCacheOne`` objects that are trimmed in place after a rejection.  Every forward runs on the HIP
path: with ``use_cache`` only the tokens a model has not seen go through the layers, and the vocabulary
projection is evaluated only at the positions the algorithm reads (the last one for a draft, the
gamma + 1 last ones for the verification).  Control flow, return values and the order of the random
draws follow the reference line by line; ``rand_fn`` (default ``torch.rand`` on the target's device) only
exists so that tests can fix the acceptance draws.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple, Union

import torch
from torch import Tensor

from .layers.kv_cache import DynamicCache, DynamicCacheOne, StaticCache, StaticCacheOne
from .logits_processors import LogitsProcessor, NucleusProcessor


def trim_cache(cache, num_tokens_to_discard: int):
    """Drop the last ``num_tokens_to_discard`` tokens (reference :9-71).  Dynamic caches shrink in place;
    static caches are written at ``start_pos`` on the next call, so their discarded slots are only
    cleared."""
    if cache is None:
        return None
    n = int(num_tokens_to_discard)
    if n <= 0:
        return cache
    if isinstance(cache, DynamicCacheOne):
        for b in cache._bufs:
            b.length = max(b.length - n, 0)
    elif isinstance(cache, DynamicCache):
        cache._buf.length = max(cache._buf.length - n, 0)
    elif isinstance(cache, (StaticCacheOne, StaticCache)):
        pass
    else:
        raise ValueError("Unsupported cache type.")
    return cache


def norm_fn(x: Tensor) -> Tensor:
    """norm(max(0, x)) (reference :73-83)."""
    x_max = torch.where(x > 0, x, torch.zeros_like(x))
    return x_max / torch.sum(x_max, dim=-1, keepdim=True)


class _Stepper:
    """A decoder and its cache: logits at positions [first, L) of a prefix of L tokens."""

    def __init__(self, model, use_cache: bool):
        self.model = model
        self.cache = DynamicCacheOne(model.config) if use_cache else None
        self.device = next(model.parameters()).device

    def logits(self, ids: Tensor, first: int) -> Tensor:
        ids = ids.to(self.device)
        L = ids.shape[1]
        mask = torch.ones((1, L), dtype=torch.long, device=self.device)
        if self.cache is not None:
            seen = len(self.cache)
            if seen > first:
                raise RuntimeError(f"cache holds {seen} tokens, logits wanted from position {first}")
            hidden, self.cache = self.model.forward_hidden(ids[:, seen:], mask, True, self.cache, seen)
            first -= seen
        else:
            hidden, _ = self.model.forward_hidden(ids, mask)
        return self.model.lm_head(hidden[:, first:, :].contiguous())

    def trim(self, n: int) -> None:
        if self.cache is not None:
            trim_cache(self.cache, n)


@torch.no_grad()
def speculative_generate(
    inputs: Tensor,
    drafter: torch.nn.Module,
    target: torch.nn.Module,
    gamma: int = 5,
    logits_processor: Optional[LogitsProcessor] = None,
    max_gen_len: Optional[int] = 128,
    eos_tokens_id: Optional[Union[int, List[int]]] = 2,
    pad_token_id: int = 2,
    use_cache: Optional[bool] = False,
    skip_sample_adjustment: Optional[bool] = False,
    first_target: Optional[bool] = True,
    rand_fn: Optional[Callable[[int], Tensor]] = None,
) -> Tuple[List[int], float]:
    """-> (generated ids, accepted drafts / speculated drafts).  Batch size 1 (reference :130)."""
    if logits_processor is None:
        logits_processor = NucleusProcessor(temperature=0.2, top_p=0.9)   # the reference's default (:91)
    tgt, drf = _Stepper(target, bool(use_cache)), _Stepper(drafter, bool(use_cache))
    dev = tgt.device
    if rand_fn is None:
        rand_fn = lambda n: torch.rand(n, device=dev)   # noqa: E731

    list_tokens_id = eos_tokens_id if isinstance(eos_tokens_id, list) else [eos_tokens_id]
    stop_tokens = torch.tensor(list_tokens_id, dtype=torch.long, device=dev).unsqueeze(1)
    assert inputs.shape[0] == 1, "Speculative decoding only supports batch size 1."
    assert drafter.config.vocab_size == target.config.vocab_size, \
        "Drafter and target models should have the same vocabulary size."
    drafts_accepted, drafts_speculated = .0, .0
    vocabulary_size = target.config.vocab_size

    prompt_len = len(inputs[0])
    cfg = target.config
    max_seq_length = cfg.max_position_embeddings if hasattr(cfg, "max_position_embeddings") else (
        cfg.max_context_length if hasattr(cfg, "max_context_length") else 512)
    total_len = min(max_seq_length, prompt_len + max_gen_len)
    input_ids = torch.full((1, total_len), pad_token_id, dtype=torch.long, device=dev)
    input_ids[0, :prompt_len] = inputs.to(dev)
    current_position = prompt_len

    if first_target:
        # prefill the target's cache and take a first token from it (reference :148-162)
        p_p = logits_processor(tgt.logits(input_ids[..., :current_position], current_position - 1)[..., -1, :])
        t = logits_processor.sample(p_p)
        input_ids[0, current_position] = t
        current_position += 1
        if torch.isin(t, stop_tokens):
            return input_ids[0, prompt_len:current_position].tolist(), 0

    while current_position < total_len:
        corrected_gamma = min(gamma, total_len - current_position - 1)
        q = torch.zeros((1, corrected_gamma, vocabulary_size), device=dev)

        for k in range(corrected_gamma):   # gamma drafts (reference :172-185)
            draft_logits = drf.logits(input_ids[..., :current_position + k], current_position + k - 1)[..., -1, :]
            draft_probs = logits_processor(draft_logits)
            q[0, k] = draft_probs.to(dev)
            xi = logits_processor.sample(draft_probs)
            input_ids[0, current_position + k] = xi
        drafts_speculated += corrected_gamma

        # the target on the drafts: logits of positions current-1 .. current+gamma-1 (reference :189-197)
        mp = tgt.logits(input_ids[..., :current_position + corrected_gamma], current_position - 1)
        p = logits_processor(mp[..., :corrected_gamma, :])

        r = rand_fn(corrected_gamma)
        fractions = p / q
        n = corrected_gamma
        for i in range(corrected_gamma):   # rejection sampling (reference :200-206)
            if r[i] > fractions[0, i, input_ids[0, current_position + i]]:
                n = i
                break
        drafts_accepted += n

        stop_locations = torch.nonzero(torch.eq(input_ids[..., current_position:current_position + n], stop_tokens))
        if stop_locations.shape[0] > 0:
            stop_location = stop_locations[0, 1].item()
            return (input_ids[0, prompt_len:current_position + stop_location + 1].tolist(),
                    drafts_accepted / drafts_speculated)

        if n == corrected_gamma:
            p_p = logits_processor(mp[..., corrected_gamma, :])
        else:
            if use_cache:
                drf.trim(corrected_gamma - n)
                tgt.trim(corrected_gamma - n + 1)
            p_p = p[..., n, :] if skip_sample_adjustment else norm_fn(p[..., n, :] - q[0, n, :])
        x = logits_processor.sample(p_p)

        input_ids[0, current_position + n:current_position + corrected_gamma] = pad_token_id
        input_ids[0, current_position + n] = x
        current_position += n + 1
        if torch.isin(x, stop_tokens):
            return input_ids[0, prompt_len:current_position].tolist(), drafts_accepted / drafts_speculated

    return input_ids[0, prompt_len:].tolist(), drafts_accepted / drafts_speculated
