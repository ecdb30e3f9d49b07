"""Speculative decoding (Leviathan et al., arXiv 2211.17192) over vyomai_amd ``DecoderModel``s -- the
algorithm, arguments, return values and order of random draws of the reference's ``speculative_generate``
(VyomAI/speculative_decoding.py:86-245), which drives HF-style models with ``transformers`` caches.

Here drafter and target are the native decoders.  Each owns a ``DynamicCacheOne`` that is trimmed in place
after a rejection; with ``use_cache`` only the tokens a model has not seen go through its layers, and the
vocabulary projection is evaluated only where the algorithm reads it (the last position for a draft, the
gamma + 1 last positions for the verification).  ``rand_fn`` (default ``torch.rand`` on the target's
device) exists so that tests can fix the acceptance draws.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple, Union

import torch
from torch import Tensor

from .layers.kv_cache import DynamicCache, DynamicCacheOne, StaticCache, StaticCacheOne
from .logits_processors import LogitsProcessor, NucleusProcessor


def trim_cache(cache, num_tokens_to_discard: int):
    """Forget the last ``num_tokens_to_discard`` tokens of a cache (reference :9-71).  Dynamic caches shrink
    in place (their buffers stay, the next write lands on the forgotten slots); static caches are written
    at an explicit ``start_pos`` anyway and are returned as they are."""
    if cache is None:
        return None
    drop = max(int(num_tokens_to_discard), 0)
    if isinstance(cache, DynamicCacheOne):
        bufs = cache._bufs
    elif isinstance(cache, DynamicCache):
        bufs = [cache._buf]
    elif isinstance(cache, (StaticCacheOne, StaticCache)):
        bufs = []
    else:
        raise ValueError("Unsupported cache type.")
    for buf in bufs:
        buf.length = max(buf.length - drop, 0)
    return cache


def norm_fn(x: Tensor) -> Tensor:
    """The positive part of x, normalised to sum 1 (reference :73-83)."""
    pos = torch.where(x > 0, x, torch.zeros_like(x))   # (a NaN counts as 0, as in the reference)
    return pos / pos.sum(dim=-1, keepdim=True)


class _Decoder:
    """One model of the pair with its cache: ``logits_from(tokens, first)`` -> (1, len - first, V) logits of the
    positions first.. of the token prefix, running only what the cache does not hold yet."""

    def __init__(self, model, cached: bool):
        self.model = model
        self.device = next(model.parameters()).device
        self.cache = DynamicCacheOne(model.config) if cached else None

    def logits_from(self, tokens: Tensor, first: int) -> Tensor:
        tokens = tokens.to(self.device)
        keys = torch.ones_like(tokens)
        if self.cache is None:
            hidden, _ = self.model.forward_hidden(tokens, keys)
        else:
            have = len(self.cache)
            if have > first:
                raise RuntimeError(f"the cache holds {have} tokens but logits are wanted from position {first}")
            hidden, self.cache = self.model.forward_hidden(tokens[:, have:], keys, True, self.cache, have)
            first -= have
        return self.model.lm_head(hidden[:, first:, :].contiguous())

    def forget(self, n: int) -> None:
        trim_cache(self.cache, n)


def _first_rejected(accept_ratio: Tensor, draws: Tensor) -> int:
    """Index of the first draft whose draw exceeds p/q, or the number of drafts (reference :200-206)."""
    for i in range(draws.shape[0]):
        if draws[i] > accept_ratio[i]:
            return i
    return draws.shape[0]


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
    """-> (generated token ids, accepted drafts / proposed drafts).  One sequence at a time (reference :130)."""
    proc = logits_processor if logits_processor is not None else NucleusProcessor(temperature=0.2, top_p=0.9)
    if inputs.shape[0] != 1:
        raise AssertionError("Speculative decoding only supports batch size 1.")
    if drafter.config.vocab_size != target.config.vocab_size:
        raise AssertionError("Drafter and target models should have the same vocabulary size.")
    big, small = _Decoder(target, bool(use_cache)), _Decoder(drafter, bool(use_cache))
    dev = big.device
    draw = rand_fn if rand_fn is not None else (lambda n: torch.rand(n, device=dev))
    stops = torch.tensor(eos_tokens_id if isinstance(eos_tokens_id, list) else [eos_tokens_id],
                         dtype=torch.long, device=dev).unsqueeze(1)
    cfg = target.config
    limit = getattr(cfg, "max_position_embeddings", None) or getattr(cfg, "max_context_length", 512)
    start = inputs.shape[1]
    end = min(limit, start + max_gen_len)
    seq = torch.full((1, end), pad_token_id, dtype=torch.long, device=dev)
    seq[0, :start] = inputs.to(dev)[0]
    pos = start
    accepted = proposed = 0.0

    def finished(upto: int):
        return seq[0, start:upto].tolist(), (accepted / proposed if proposed else 0)

    if first_target:   # the target prefills its cache and contributes the first token (reference :148-162)
        tok = proc.sample(proc(big.logits_from(seq[:, :pos], pos - 1)[:, -1, :]))
        seq[0, pos] = tok
        pos += 1
        if torch.isin(tok, stops):
            return seq[0, start:pos].tolist(), 0

    while pos < end:
        g = min(gamma, end - pos - 1)
        q = torch.zeros((1, g, cfg.vocab_size), device=dev)
        for j in range(g):   # the drafter proposes g tokens (reference :172-185)
            q[0, j] = proc(small.logits_from(seq[:, :pos + j], pos + j - 1)[:, -1, :]).to(dev)
            seq[0, pos + j] = proc.sample(q[:, j])
        proposed += g

        # one target pass scores them all: logits of positions pos-1 .. pos+g-1 (reference :189-197)
        scores = big.logits_from(seq[:, :pos + g], pos - 1)
        p = proc(scores[:, :g, :])
        ratio = (p / q)[0, torch.arange(g, device=dev), seq[0, pos:pos + g]]
        n = _first_rejected(ratio, draw(g))
        accepted += n

        hit = torch.nonzero(torch.eq(seq[:, pos:pos + n], stops))   # a stop token among the accepted (:211-216)
        if hit.shape[0] > 0:
            return finished(pos + int(hit[0, 1]) + 1)

        if n == g:
            nxt = proc.sample(proc(scores[:, g, :]))
        else:
            if use_cache:   # both caches go back to the last accepted token (reference :224-226)
                small.forget(g - n)
                big.forget(g - n + 1)
            resid = p[:, n, :] if skip_sample_adjustment else norm_fn(p[:, n, :] - q[0, n, :])
            nxt = proc.sample(resid)
        seq[0, pos + n:pos + g] = pad_token_id
        seq[0, pos + n] = nxt
        pos += n + 1
        if torch.isin(nxt, stops):
            return finished(pos)

    return finished(end)
