"""Seed / offset bookkeeping of the dropout masks (vy_linear_dropout_fwd / vy_dropout).

A mask is a pure function of (seed, offset, row, column): every dropout site draws a fresh `offset`
per forward call and hands (p, seed, offset) to its backward, which regenerates the same mask instead
of storing it.  The seed defaults to torch's (torch.manual_seed controls it, as it controls nn.Dropout
in the reference); ranks of a data-parallel job get different masks through `rank_offset`."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

_MASK64 = (1 << 64) - 1
_state = {"seed": None, "offset": 0}


def manual_seed(seed: int) -> None:
    _state["seed"] = int(seed) & _MASK64
    _state["offset"] = 0


def _rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def next_dropout(p: float) -> Optional[Tuple[float, int, int]]:
    """-> (p, seed, offset) for one dropout call, or None when p == 0."""
    if p <= 0.0:
        return None
    seed = _state["seed"]
    if seed is None:
        seed = torch.initial_seed() & _MASK64
    _state["offset"] += 1
    return float(p), seed, (_rank() << 40) + _state["offset"]
