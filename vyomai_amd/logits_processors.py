"""Logits processors for sampling: the reference's five classes (VyomAI/logits_processors.py) on the HIP path.

A processor is three numbers -- temperature, top_k, top_p -- and a way to pick a token.
``processor(logits)`` is ``softmax(mask(logits) / temperature)``: ONE kernel, ``vy_sampling_probs``, finds the
k-th largest value and the nucleus cut by radix selection (no sort) and writes the probabilities;
``processor.sample(probs)`` is arg-max (greedy) or ``torch.multinomial``.

Kept from the reference because callers can observe it: the top-k processors overwrite the CALLER's logits
below the k-th largest value with -1e20 (reference :59-63, :92-95); the nucleus processor does not touch its
input (it masks a sorted copy, :73-81); probabilities come back in the dtype of the logits.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import ops

REMOVED = -1e20   # what the reference writes into a removed logit (:62, :79)


class LogitsProcessor:
    """Base of the five processors (reference :7-23, where ``_process`` and ``sample`` are abstract)."""

    stochastic = True    # sample() draws; False: arg-max

    def __init__(self, temperature: float, top_k=None, top_p=None):
        if type(self) is LogitsProcessor:
            raise TypeError("LogitsProcessor is abstract: use Greedy/Multinomial/TopK/Nucleus/TopKNucleusProcessor")
        self.temperature = temperature
        if top_k is not None:
            self.top_k = top_k
        if top_p is not None:
            self.top_p = top_p

    # ---- the two numbers the kernel needs ------------------------------------------------------
    def _k(self, logits: Tensor) -> int:
        k = getattr(self, "top_k", 0)
        return min(k, logits.size(-1)) if k else 0

    def _p(self) -> float:
        return float(getattr(self, "top_p", 0.0))

    def _support(self, logits: Tensor, k: int, p: float) -> Tensor:
        """True where a token survives the masking (its probability at temperature 1 is not 0)."""
        return ops.sampling_probs(logits, 1.0, k, p) > 0

    def _mask_topk_in_place(self, logits: Tensor) -> None:
        k = self._k(logits)
        if k and k < logits.size(-1):
            logits.masked_fill_(~self._support(logits, k, 0.0), REMOVED)

    # ---- reference interface -------------------------------------------------------------------
    def __call__(self, logits: Tensor) -> Tensor:
        probs = ops.sampling_probs(logits, self.temperature, self._k(logits), self._p())
        self._mask_topk_in_place(logits)          # after the kernel read them (masking twice changes nothing)
        return probs.to(logits.dtype)

    def _process(self, logits: Tensor) -> Tensor:
        """The masked logits (removed entries = -1e20); identity for greedy / multinomial."""
        k, p = self._k(logits), self._p()
        if not k and not p:
            return logits
        if p:
            out = torch.where(self._support(logits, k, p), logits, torch.full_like(logits, REMOVED))
            self._mask_topk_in_place(logits)
            return out
        self._mask_topk_in_place(logits)
        return logits

    def sample(self, probs: Tensor) -> Tensor:
        if self.stochastic:
            return torch.multinomial(probs.float(), num_samples=1)
        return probs.argmax(dim=-1, keepdim=True)


class GreedyProcessor(LogitsProcessor):
    """Most probable token (reference :26-36)."""

    stochastic = False

    def __init__(self, temperature: float = 1):
        LogitsProcessor.__init__(self, temperature)


class MultinomialProcessor(LogitsProcessor):
    """Random sampling from the whole distribution (reference :39-49)."""

    def __init__(self, temperature: float):
        LogitsProcessor.__init__(self, temperature)


class TopKProcessor(MultinomialProcessor):
    """Random sampling among the k most probable tokens (reference :52-63)."""

    def __init__(self, temperature: float, top_k: int):
        LogitsProcessor.__init__(self, temperature, top_k=top_k)


class NucleusProcessor(MultinomialProcessor):
    """Random sampling inside the smallest set of tokens whose mass exceeds top_p (reference :66-81)."""

    def __init__(self, temperature: float, top_p: float):
        LogitsProcessor.__init__(self, temperature, top_p=top_p)


class TopKNucleusProcessor(MultinomialProcessor):
    """Top-k first, then the nucleus of what is left (reference :84-102)."""

    def __init__(self, temperature: float, top_k: int, top_p: float):
        LogitsProcessor.__init__(self, temperature, top_k=top_k, top_p=top_p)
