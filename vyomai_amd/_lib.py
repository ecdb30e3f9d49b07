"""ctypes binding of libvyom_hip.so (C ABI in include/vyom_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950) into
``vyomai_amd/lib/``.  There is no fallback: if it is missing or a call fails, the op raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# VY_LIB_PATH: developer knob for A/B runs of two builds of the same ABI on one box (tools/ab_lib.sh)
LIB_PATH = os.environ.get("VY_LIB_PATH") or os.path.join(_HERE, "lib", "libvyom_hip.so")

VY_F32, VY_BF16 = 0, 1
ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH = 0, 1, 2
ACT_SAVE_DERIV = 0x100   # saved tensor = act'(pre) instead of pre (include/vyom_hip.h)
MASK_NONE, MASK_CAUSAL, MASK_KEYPAD, MASK_ADDITIVE = 0, 1, 2, 4

_p, _i64, _i, _f, _u64 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_uint64

# name -> argtypes, exactly the prototypes of include/vyom_hip.h
PROTOTYPES = {
    "vy_linear_fwd": [_p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i, _i, _p],
    "vy_qkv_rope_fwd": [_p, _i64, _p, _i64, _p, _p, _p, _i64,
                        _p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _p, _i64, _i64, _i64,
                        _i64, _i64, _i64, _i, _i, _i, _i, _p],
    "vy_attn_fwd": [_p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _p, _i64, _i64, _i64,
                    _p, _i64, _i64, _p, _i, _i64, _p, _i64, _p, _i64, _i64,
                    _i64, _i, _i, _i64, _i64, _i, _f, _i, _p],
    "vy_attn_decode": [_p, _i64, _i64, _p, _i64, _i64, _i64, _p, _i64, _i64, _i64,
                       _p, _i64, _i64, _i, _i, _i64, _i, _f, _i, _p],
    "vy_layernorm_fwd": [_p, _i64, _p, _p, _p, _i64, _p, _p, _i64, _i64, _f, _i, _p],
    "vy_rmsnorm_fwd": [_p, _i64, _p, _p, _i64, _i64, _i64, _f, _f, _i, _p],
    "vy_gated_act_fwd": [_p, _i64, _p, _i64, _i64, _i64, _i, _i, _p],
    "vy_rope_fwd": [_p, _i64, _i64, _i64, _p, _p, _i64, _i64, _i, _i64, _i, _i, _i, _p],
    "vy_linear_dgrad": [_p, _i64, _p, _i64, _p, _i64, _i, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _i, _p],
    "vy_linear_wgrad": [_p, _i64, _p, _i64, _p, _i64, _p, _f, _p, _i64, _i64, _i64, _i, _p],
    "vy_linear_wgrad_grouped": [_p, _i, _i, _p],
    "vy_layernorm_bwd": [_p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _p, _p, _f, _p, _i64, _i64, _i, _p],
    "vy_attn_bwd": [_p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _p, _i64, _i64, _i64,
                    _p, _p, _i64, _i64, _p, _p,
                    _p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _p, _i64, _i64, _i64,
                    _i, _i64, _p, _i64, _p, _p, _i64, _i64, _i, _i, _i64, _i64, _i, _f, _i, _p],
    "vy_adamw_step": [_p, _p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _i64, _f, _p, _p],
    "vy_adamw_step_gated": [_p, _p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _i64, _f, _p, _p, _p],
    "vy_sumsq": [_p, _i64, _p, _p, _p],
    "vy_linear_dropout_fwd": [_p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _f, _u64, _u64, _i, _p],
    "vy_dropout": [_p, _i64, _p, _i64, _i64, _i64, _f, _u64, _u64, _i, _p],
    "vy_act_bwd": [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i, _i, _p],
    "vy_xent_fwd": [_p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _p, _i, _p],
    "vy_xent_bwd": [_p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _i, _p],
    "vy_xent_fused": [_p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _i64, _p, _i, _p],
    "vy_transpose_batched": [_p, _i, _i, _i, _p],
    "vy_embedding_fwd": [_p, _i64, _p, _p, _i64, _i64, _i64, _i64, _p, _i, _p],
    "vy_embedding_bwd": [_p, _i64, _p, _p, _i64, _i64, _i64, _i64, _i64, _i, _p],
    "vy_transpose": [_p, _i64, _p, _i64, _i64, _i64, _i, _p],
    "vy_greedy_step": [_p, _i64, _i64, _i64, _i, _p, _i64, _i64, _p, _i64, _p, _i, _p, _p, _p],
    "vy_sampling_probs": [_p, _i64, _i64, _i64, _i, _f, _i, _f, _p, _i64, _p],
    "vy_decoder_step": [_p, _p, _i64, _p, _p, _p, _i64, _p],
    "vy_gemma_decoder_step": [_p, _p, _i64, _p, _i64, _p],
    "vy_cast": [_p, _p, _i64, _i, _i, _p],
    "vy_ddp_unique_id": [_p],
    "vy_ddp_init": [_p, _i, _i],
    "vy_ddp_all_reduce_async": [_p, _i64, _i, _p],
    "vy_ddp_destroy": [],
    "vy_set_concurrent_chains": [_i],
    "vy_workspace_set": [_p, _p, _i64],
}
OTHER_SYMBOLS = ["vy_last_error", "vy_abi_version", "vy_layernorm_bwd_ws_rows", "vy_decode_ws_bytes",
                 "vy_gemma_ws_bytes", "vy_ddp_world", "vy_ddp_rank"]
ALL_SYMBOLS = list(PROTOTYPES) + OTHER_SYMBOLS


class VyWgradDesc(C.Structure):
    """vy_wgrad_desc of include/vyom_hip.h."""
    _fields_ = [("dy", _p), ("lddy", _i64), ("x", _p), ("ldx", _i64), ("dw", _p), ("lddw", _i64), ("db", _p),
                ("M", _i64), ("N", _i64), ("K", _i64)]


class VyomHipError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the kernel library; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VyomHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  vyomai_amd has no CPU or PyTorch fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.vy_last_error.restype = C.c_char_p
    lib.vy_last_error.argtypes = []
    lib.vy_abi_version.restype = C.c_int
    lib.vy_layernorm_bwd_ws_rows.restype = C.c_int64
    lib.vy_layernorm_bwd_ws_rows.argtypes = [_i64]
    lib.vy_decode_ws_bytes.restype = C.c_int64
    lib.vy_decode_ws_bytes.argtypes = [C.c_int32] * 7
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise VyomHipError(f"{name} failed ({rc}): {lib.vy_last_error().decode()}")


def dtype_code(torch_dtype) -> int:
    import torch

    if torch_dtype == torch.bfloat16:
        return VY_BF16
    if torch_dtype == torch.float32:
        return VY_F32
    raise VyomHipError(f"vyomai_amd kernels support float32 and bfloat16, got {torch_dtype}")
