"""Deterministic weight / input recipe shared by the golden-vector generator, the
parity tests and bench.py.

Every tensor is a pure function of (name, shape): a counter-based integer hash
(splitmix64) of ``crc32(name) + flat_index`` mapped to uniform values.  No torch
RNG, no weight files.  The values are exactly representable in fp32 (24-bit
mantissa draws), so the reference (in the build container) and this package
(on the GPU box) see bit-identical parameters.

numpy only — this module must stay importable without a GPU and without the
HIP extension.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(name: str, shape: Tuple[int, ...], scale: float = 1.0, offset: float = 0.0) -> np.ndarray:
    """fp32 array of ``shape`` with values offset + scale * u, u uniform in [-1, 1)."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(name.encode("utf-8"))) << np.uint64(32)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + seed
    bits = _splitmix64(idx) >> np.uint64(40)  # 24 random bits
    u = bits.astype(np.float64) / float(1 << 23) - 1.0  # [-1, 1), exact in fp32
    return (offset + scale * u).astype(np.float32).reshape(shape)


def token_ids(name: str, shape: Tuple[int, ...], low: int, high: int) -> np.ndarray:
    """int64 ids in [low, high)."""
    n = int(np.prod(shape))
    seed = np.uint64(zlib.crc32(name.encode("utf-8"))) << np.uint64(32)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + seed
    r = _splitmix64(idx) >> np.uint64(11)
    return (low + (r % np.uint64(high - low)).astype(np.int64)).reshape(shape)


def param_value(name: str, shape: Tuple[int, ...]) -> np.ndarray:
    """Recipe for a parameter called ``name`` in a reference-compatible state_dict.

    * LayerNorm scales (``layernorm.weight`` / ``layer_norm.weight``): 1 + 0.1 u
    * biases and LayerNorm shifts: 0.02 u
    * embeddings / cls_token / pos_embeddings: 1.0 u (torch default init is N(0,1))
    * matrices: u / sqrt(fan_in) * 0.8  (keeps post-LN activations O(1) like default init)
    """
    # LMHead ties ``decoder.bias`` to ``bias`` (reference models/decoder.py:263-265):
    # both state_dict keys must resolve to one value.
    name = name.replace("lm_head.decoder.bias", "lm_head.bias")
    # the seq2seq LMHead ties ``vocab.bias`` to ``bias`` the same way (models/encoder_decoder.py:97-99)
    name = name.replace("lm_head.vocab.bias", "lm_head.bias")
    leaf = name.rsplit(".", 1)[-1]
    is_ln = ("layernorm" in name) or ("layer_norm" in name) or ("norm." in name)
    if is_ln and leaf == "weight":
        return uniform(name, shape, 0.1, 1.0)
    if leaf == "bias" or len(shape) <= 1:
        return uniform(name, shape, 0.02)
    if "embeddings" in name or "cls_token" in name:
        return uniform(name, shape, 1.0)
    fan_in = int(np.prod(shape[1:]))
    return uniform(name, shape, 0.8 / np.sqrt(fan_in))


def state_dict_values(named_shapes: Iterable[Tuple[str, Tuple[int, ...]]]) -> Dict[str, np.ndarray]:
    return {n: param_value(n, tuple(s)) for n, s in named_shapes}


def load_recipe_(module) -> None:
    """Fill every entry of ``module.state_dict()`` in place from the recipe (torch module)."""
    import torch

    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            if not t.is_floating_point():
                continue
            v = torch.from_numpy(param_value(name, tuple(t.shape)))
            t.copy_(v.to(t.dtype))
