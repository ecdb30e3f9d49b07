"""Logits processors for sampling -- the reference's classes (VyomAI/logits_processors.py) on the HIP path.

``processor(logits)`` returns ``softmax(_process(logits) / temperature)``; the masking (top-k threshold,
nucleus cut) and the softmax are ONE kernel, ``vy_sampling_probs`` (radix selection in registers, no sort).
What the reference does as a side effect is kept: TopKProcessor / TopKNucleusProcessor write -1e20 into the
caller's ``logits`` below the k-th largest value (reference :59-63, :92-95); NucleusProcessor leaves its input
alone (it masks a sorted copy, :73-81).  Probabilities come back in the dtype of ``logits`` (computed in fp32).
"""
from __future__ import annotations

import abc

import torch
from torch import Tensor

from . import ops

_MASKED = -1e20   # reference :62, :79


class LogitsProcessor(abc.ABC):
    """Reference :7-23."""

    top_k: int = 0
    top_p: float = 0.0

    def __init__(self, temperature: float):
        self.temperature = temperature

    def __call__(self, logits: Tensor) -> Tensor:
        probs = ops.sampling_probs(logits, self.temperature, self._k(logits), self.top_p)
        self._side_effect(logits)
        return probs.to(logits.dtype)

    def _k(self, logits: Tensor) -> int:
        return min(self.top_k, logits.size(-1)) if self.top_k else 0

    def _side_effect(self, logits: Tensor) -> None:
        pass

    @abc.abstractmethod
    def _process(self, logits: Tensor) -> Tensor:
        pass

    @abc.abstractmethod
    def sample(self, probs: Tensor) -> Tensor:
        pass

    def _masked(self, logits: Tensor, top_k: int, top_p: float) -> Tensor:
        """logits with -1e20 where the processor removes a token (its support at temperature 1)."""
        support = ops.sampling_probs(logits, 1.0, top_k, top_p) > 0
        return torch.where(support, logits, torch.full_like(logits, _MASKED))


class GreedyProcessor(LogitsProcessor):
    """Greedy: most probable token (reference :26-36)."""

    def __init__(self, temperature: float = 1):
        super().__init__(temperature)

    def _process(self, logits: Tensor) -> Tensor:
        return logits

    def sample(self, probs: Tensor) -> Tensor:
        return torch.argmax(probs, dim=-1).unsqueeze(-1)


class MultinomialProcessor(LogitsProcessor):
    """Multinomial: random sampling (reference :39-49)."""

    def __init__(self, temperature: float):
        super().__init__(temperature)

    def _process(self, logits: Tensor) -> Tensor:
        return logits

    def sample(self, probs: Tensor) -> Tensor:
        return torch.multinomial(probs.float(), num_samples=1)


class TopKProcessor(MultinomialProcessor):
    """Top-k sampling (reference :52-63)."""

    def __init__(self, temperature: float, top_k: int):
        super().__init__(temperature)
        self.top_k = top_k

    def _side_effect(self, logits: Tensor) -> None:
        logits.copy_(self._masked(logits, self._k(logits), 0.0))   # the reference masks in place

    def _process(self, logits: Tensor) -> Tensor:
        self._side_effect(logits)
        return logits


class NucleusProcessor(MultinomialProcessor):
    """Nucleus (top-p) sampling (reference :66-81)."""

    def __init__(self, temperature: float, top_p: float):
        super().__init__(temperature)
        self.top_p = top_p

    def _process(self, logits: Tensor) -> Tensor:
        return self._masked(logits, 0, self.top_p)


class TopKNucleusProcessor(MultinomialProcessor):
    """Top-k, then nucleus over what is left (reference :84-102)."""

    def __init__(self, temperature: float, top_k: int, top_p: float):
        super().__init__(temperature)
        self.top_k = top_k
        self.top_p = top_p

    def _side_effect(self, logits: Tensor) -> None:
        logits.copy_(self._masked(logits, self._k(logits), 0.0))

    def _process(self, logits: Tensor) -> Tensor:
        out = self._masked(logits, self._k(logits), self.top_p)
        self._side_effect(logits)
        return out
