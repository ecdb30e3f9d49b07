"""Tensor-level wrappers over the C ABI (include/vyom_hip.h).

PyTorch is used for device memory and the current HIP stream only; every computation below is a
call into libvyom_hip.so.  All functions require CUDA (ROCm) tensors and raise otherwise -- there
is no CPU or eager fallback.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import call, dtype_code

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------
# Two lanes: the training forward as two batch halves on two HIP streams.
#
# Every kernel of the block forward is row-separable (GEMM rows, LayerNorm rows, attention batches), so the
# first B0 sequences can run through the whole stack on the current stream while the remaining ones run through
# it on a side stream, both writing into the SAME full-batch tensors (which is all the full-batch backward needs).
# What it buys on MI355X: a K = 768 GEMM launch is a k-loop bound by what the CUs take in from L2 followed by an
# epilogue in which all 256 workgroups store at once (HBM-bound, matrix pipe idle) -- a third of the launch.  Two
# independent half-size launch chains drift apart, so one chain's epilogue / LayerNorm / attention softmax runs
# under the other's MFMA loop.
#
# Ordering rules (correct by construction):
#   * lane 1 waits, before each of its launches, for everything the host has enqueued on the main stream so far
#     EXCEPT lane 0's launch of the same op (event recorded first): torch ops on the main stream (mask bytes,
#     casts) are visible to it; it can lag behind lane 0 but never run ahead of the host's program order;
#   * any op that is not lane-aware, and the end of the region, makes the main stream wait for lane 1 first;
#   * the region is entered only where every tensor produced inside is kept until after the join (training
#     forward: saved for backward), so the caching allocator -- which sees main-stream use only -- never hands a
#     block that lane 1 still touches to someone else.
# ------------------------------------------------------------------------------------------


class _Lanes:
    active = False
    side: Optional[torch.cuda.Stream] = None
    fork_ev: Optional[torch.cuda.Event] = None
    join_ev: Optional[torch.cuda.Event] = None
    B = L = B0 = 0
    dirty = False        # lane 1 has work the main stream has not waited for yet
    keep: list = []      # every tensor a split launch touched: alive until the join, whatever the caller does with it
    streams = {}


class lanes:
    """with ops.lanes(B, L, enabled): ... -- see the block comment above.  No-op when not enabled or B < 2."""

    def __init__(self, B: int, L: int, enabled: bool = True):
        self.on = bool(enabled) and B >= 2 and not _Lanes.active and torch.cuda.is_available()
        self.B, self.L = B, L

    def __enter__(self):
        if self.on:
            dev = torch.cuda.current_device()
            st = _Lanes.streams.get(dev)
            if st is None:
                st = _Lanes.streams[dev] = (torch.cuda.Stream(), torch.cuda.Event(), torch.cuda.Event())
            _Lanes.side, _Lanes.fork_ev, _Lanes.join_ev = st
            _Lanes.B, _Lanes.L, _Lanes.B0 = self.B, self.L, self.B // 2
            _Lanes.active, _Lanes.dirty, _Lanes.keep = True, False, []
            call("vy_set_concurrent_chains", 2)   # (grids are sized on the host, in program order with the launches)
        return self

    def __exit__(self, *exc):
        if self.on:
            _lanes_join()
            _Lanes.active = False
            call("vy_set_concurrent_chains", 1)
            if _Lanes.keep:
                # the allocator may reuse these blocks for main-stream work from here on: that work is ordered behind
                # the join just enqueued
                _Lanes.keep = []
        return False


def _lanes_join() -> None:
    if _Lanes.active and _Lanes.dirty:
        _Lanes.join_ev.record(_Lanes.side)
        torch.cuda.current_stream().wait_event(_Lanes.join_ev)
        _Lanes.dirty = False


def _lane_plan(rows: Optional[int] = None, batches: Optional[int] = None, keep=()):
    """-> [(first, count, stream)] over rows (of a (B*L, .) view) or batches: one entry when the op runs whole, two
    when it is split over the lanes."""
    main = torch.cuda.current_stream()
    if _Lanes.active:
        if rows is not None and rows == _Lanes.B * _Lanes.L:
            cut, n = _Lanes.B0 * _Lanes.L, rows
        elif batches is not None and batches == _Lanes.B:
            cut, n = _Lanes.B0, batches
        else:
            cut = None
        if cut is not None:
            _Lanes.fork_ev.record(main)
            _Lanes.side.wait_event(_Lanes.fork_ev)
            _Lanes.dirty = True
            if keep:
                _Lanes.keep.extend(t for t in keep if t is not None)
            return [(0, cut, main.cuda_stream), (cut, n - cut, _Lanes.side.cuda_stream)]
        _lanes_join()
    n = rows if rows is not None else batches
    return [(0, n, main.cuda_stream)]


def _stream() -> int:
    """The stream of an op that always runs whole (inside a lane region: after the main stream has joined lane 1)."""
    _lanes_join()
    return torch.cuda.current_stream().cuda_stream


_WS = {}
_WS_BYTES = 64 << 20


def _mid_ws(rows: int) -> None:
    """Mid-size GEMMs (32 < rows <= 4096) split K over workgroups and need scratch for their fp32 partial tiles
    (vy_workspace_set: one buffer per stream, registered once; the library never allocates device memory)."""
    if rows <= 32 or rows > 4096 or _Lanes.active:
        return
    st = torch.cuda.current_stream()
    key = (st.device_index, st.cuda_stream)
    if key not in _WS:
        buf = torch.empty(_WS_BYTES, dtype=torch.uint8, device=torch.device("cuda", st.device_index))
        call("vy_workspace_set", st.cuda_stream, buf.data_ptr(), _WS_BYTES)
        _WS[key] = buf


def _at(t: Optional[Tensor], first: int):
    """data pointer of t[first:] (None stays None)."""
    if t is None:
        return None
    return t.data_ptr() + first * t.stride(0) * t.element_size()


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _need_gpu(*ts: Optional[Tensor]) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.VyomHipError(
                "vyomai_amd ops run on MI355X only: got a CPU tensor (no CPU fallback exists; "
                "move the module and its inputs to 'cuda')"
            )


def _rows(x: Tensor) -> Tensor:
    """View (..., K) as (M, K) without copying when the rows are uniformly strided."""
    if x.dim() == 2:
        return x if x.stride(1) == 1 else x.contiguous()
    if x.stride(-1) != 1:
        x = x.contiguous()
    try:
        return x.view(-1, x.shape[-1])
    except RuntimeError:
        return x.reshape(-1, x.shape[-1])


def linear(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, act: int = _lib.ACT_NONE,
           residual: Optional[Tensor] = None, pre_out: Optional[Tensor] = None,
           out: Optional[Tensor] = None, dropout: Optional[Tuple[float, int, int]] = None) -> Tensor:
    """act(x @ w.T + bias) + residual   (vy_linear_fwd).
    dropout = (p, seed, offset): dropout(x @ w.T + bias) / (1 - p) + residual  (vy_linear_dropout_fwd)."""
    _need_gpu(x, w, bias, residual)
    x2 = _rows(x)
    M, K = x2.shape
    N = w.shape[0]
    assert w.shape[1] == K and w.stride(1) == 1
    r2 = _rows(residual) if residual is not None else None
    if out is None:
        # row stride padded to 8 elements so every row is 16-byte aligned (N = vocab is odd)
        ldy = (N + 7) // 8 * 8
        buf = torch.empty((M, ldy), dtype=x.dtype, device=x.device)
        y2 = buf[:, :N]
        ret = buf.view(*x.shape[:-1], ldy)[..., :N]
    else:
        y2 = _rows(out)
        ret = out
    p2 = _rows(pre_out) if pre_out is not None else None
    if p2 is not None:
        assert p2.stride(0) == y2.stride(0)
    if dropout is not None and dropout[0] > 0.0:
        # (the mask is a function of the global row index: this launch is never split over the lanes)
        assert act == _lib.ACT_NONE and p2 is None
        _mid_ws(M)
        call("vy_linear_dropout_fwd", x2.data_ptr(), x2.stride(0), w.data_ptr(), w.stride(0), _ptr(bias),
             _ptr(r2), r2.stride(0) if r2 is not None else 0, y2.data_ptr(), y2.stride(0), M, N, K,
             float(dropout[0]), int(dropout[1]), int(dropout[2]), dtype_code(x.dtype), _stream())
        return ret
    _mid_ws(M)
    whole = _Lanes.active and (x2.data_ptr() != x.data_ptr() or (r2 is not None and r2.data_ptr() != residual.data_ptr()))
    for m0, m, st in ([(0, M, _stream())] if whole else _lane_plan(rows=M, keep=(x2, r2, y2, p2, w, bias))):
        call("vy_linear_fwd", _at(x2, m0), x2.stride(0), w.data_ptr(), w.stride(0), _ptr(bias),
             _at(r2, m0), r2.stride(0) if r2 is not None else 0, _at(y2, m0), y2.stride(0), _at(p2, m0),
             m, N, K, act, dtype_code(x.dtype), st)
    return ret


def qkv_rope(x: Tensor, w_packed: Tensor, b_packed: Optional[Tensor], h: int, hk: int, dh: int,
             cos: Optional[Tensor], sin: Optional[Tensor], pos0: int,
             q: Tensor, k: Tensor, v: Tensor) -> None:
    """Fused projection + RoPE + head split; q/k/v are (B, heads, L, dh) *views* that are written
    in place (k/v may be slices of a static KV cache).  (vy_qkv_rope_fwd)"""
    _need_gpu(x, w_packed, q, k, v)
    B, L, K = x.shape
    x2 = _rows(x)
    for t in (q, k, v):
        assert t.stride(3) == 1
    _mid_ws(B * L)
    whole = _Lanes.active and (x2.data_ptr() != x.data_ptr() or L != _Lanes.L)
    for b0, nb, st in ([(0, B, _stream())] if whole else _lane_plan(batches=B, keep=(x2, q, k, v, w_packed, b_packed))):
        call("vy_qkv_rope_fwd", _at(x2, b0 * L), x2.stride(0), w_packed.data_ptr(), w_packed.stride(0),
             _ptr(b_packed), _ptr(cos), _ptr(sin), pos0,
             _at(q, b0), q.stride(0), q.stride(1), q.stride(2),
             _at(k, b0), k.stride(0), k.stride(1), k.stride(2),
             _at(v, b0), v.stride(0), v.stride(1), v.stride(2),
             nb, L, K, h, hk, dh, dtype_code(x.dtype), st)


def attention(q: Tensor, k: Tensor, v: Tensor, *, causal: bool = False, start_pos: int = 0,
              keypad: Optional[Tensor] = None, addmask: Optional[Tensor] = None,
              scale: Optional[float] = None, lse: Optional[Tensor] = None,
              out: Optional[Tensor] = None) -> Tensor:
    """q (B,h,L,dh), k/v (B,hk,S,dh) -> (B, L, h*dh).  (vy_attn_fwd)"""
    _need_gpu(q, k, v, keypad, addmask)
    B, h, L, dh = q.shape
    hk, S = k.shape[1], k.shape[2]
    if out is None:
        out = torch.empty((B, L, h * dh), dtype=q.dtype, device=q.device)
    kind = 0
    if causal:
        kind |= _lib.MASK_CAUSAL
    if keypad is not None:
        assert keypad.dtype == torch.uint8 and keypad.shape == (B, S) and keypad.stride(1) == 1
        kind |= _lib.MASK_KEYPAD
    am_sb = am_sl = 0
    if addmask is not None:
        assert addmask.dtype == torch.float32 and addmask.dim() == 4 and addmask.stride(3) == 1
        assert addmask.shape[3] == S
        kind |= _lib.MASK_ADDITIVE
        am_sb = addmask.stride(0) if addmask.shape[0] > 1 else 0
        am_sl = addmask.stride(2) if addmask.shape[2] > 1 else 0
    if scale is None:
        scale = 1.0 / math.sqrt(dh)
    if lse is not None:
        assert lse.is_contiguous() and lse.shape == (B, h, L)
    whole = addmask is not None or (_Lanes.active and L != _Lanes.L)
    for b0, nb, st in ([(0, B, _stream())] if whole else _lane_plan(batches=B, keep=(q, k, v, out, lse, keypad))):
        call("vy_attn_fwd", _at(q, b0), q.stride(0), q.stride(1), q.stride(2),
             _at(k, b0), k.stride(0), k.stride(1), k.stride(2),
             _at(v, b0), v.stride(0), v.stride(1), v.stride(2),
             _at(out, b0), out.stride(0), out.stride(1), _at(lse, b0), kind, start_pos,
             _at(keypad, b0), keypad.stride(0) if keypad is not None else 0,
             _ptr(addmask), am_sb, am_sl, nb, h, hk, L, S, dh, float(scale), dtype_code(q.dtype), st)
    return out


def attention_decode(q: Tensor, k: Tensor, v: Tensor, S: int, scale: Optional[float] = None) -> Tensor:
    """q (B,h,1,dh) against cache k/v (B,hk,>=S,dh): attends keys [0,S).  (vy_attn_decode)"""
    _need_gpu(q, k, v)
    B, h, _, dh = q.shape
    hk = k.shape[1]
    out = torch.empty((B, 1, h * dh), dtype=q.dtype, device=q.device)
    if scale is None:
        scale = 1.0 / math.sqrt(dh)
    call("vy_attn_decode", q.data_ptr(), q.stride(0), q.stride(1),
         k.data_ptr(), k.stride(0), k.stride(1), k.stride(2),
         v.data_ptr(), v.stride(0), v.stride(1), v.stride(2),
         out.data_ptr(), out.stride(0), B, h, hk, S, dh, float(scale), dtype_code(q.dtype), _stream())
    return out


def layernorm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float,
              save_stats: bool = False) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor]]:
    _need_gpu(x, gamma, beta)
    x2 = _rows(x)
    M, N = x2.shape
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    whole = _Lanes.active and x2.data_ptr() != x.data_ptr()
    for m0, m, st in ([(0, M, _stream())] if whole else _lane_plan(rows=M, keep=(x2, y, mean, rstd, gamma, beta))):
        call("vy_layernorm_fwd", _at(x2, m0), x2.stride(0), gamma.data_ptr(), beta.data_ptr(),
             _at(y, m0), y.stride(0), _at(mean, m0), _at(rstd, m0), m, N, float(eps), dtype_code(x.dtype), st)
    return y.view(x.shape), mean, rstd


def rope_(x: Tensor, cos: Tensor, sin: Tensor, pos0: int = 0, inverse: bool = False) -> Tensor:
    """In-place rotary embedding on (B, heads, L, dh).  (vy_rope_fwd)"""
    _need_gpu(x, cos, sin)
    B, heads, L, dh = x.shape
    assert x.stride(3) == 1
    call("vy_rope_fwd", x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), cos.data_ptr(),
         sin.data_ptr(), pos0, B, heads, L, dh, 1 if inverse else 0, dtype_code(x.dtype), _stream())
    return x


def rope_tables(head_dim: int, max_pos: int, device) -> Tuple[Tensor, Tensor]:
    """fp32 cos/sin of the reference's angle table (VyomAI/layers/positional_embeddings.py:
    127-137, 173-175), half width (the reference concatenates the angles with themselves).
    Evaluated on the host in fp32 exactly like the reference, then uploaded once."""
    inv_freq = 1.0 / (10000 ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(max_pos).type_as(inv_freq)
    ang = torch.einsum("i,j->ij", t, inv_freq)
    return ang.cos().contiguous().to(device), ang.sin().contiguous().to(device)


# ------------------------------------------------------------------------------------------
# backward / training entry points
# ------------------------------------------------------------------------------------------


def linear_dgrad(dy: Tensor, wt: Tensor, pre: Optional[Tensor] = None, act: int = _lib.ACT_NONE,
                 add_to: Optional[Tensor] = None, out: Optional[Tensor] = None,
                 add_to2: Optional[Tensor] = None) -> Tensor:
    """dX = (dY @ W) * act'(pre) + add_to + add_to2, with W^T given ([K, N] row-major).  (vy_linear_dgrad)"""
    _need_gpu(dy, wt, pre, add_to, add_to2)
    if add_to is None and add_to2 is not None:
        add_to, add_to2 = add_to2, None
    d2 = _rows(dy)
    M, N = d2.shape
    K = wt.shape[0]
    assert wt.shape[1] == N and wt.stride(1) == 1
    if out is None:
        out = torch.empty((*dy.shape[:-1], K), dtype=dy.dtype, device=dy.device)
    o2 = _rows(out)
    p2 = _rows(pre) if pre is not None else None
    a2 = _rows(add_to) if add_to is not None else None
    b2 = _rows(add_to2) if add_to2 is not None else None
    _mid_ws(M)
    call("vy_linear_dgrad", d2.data_ptr(), d2.stride(0), wt.data_ptr(), wt.stride(0), _ptr(p2),
         p2.stride(0) if p2 is not None else 0, act, _ptr(a2), a2.stride(0) if a2 is not None else 0,
         _ptr(b2), b2.stride(0) if b2 is not None else 0,
         o2.data_ptr(), o2.stride(0), M, N, K, dtype_code(dy.dtype), _stream())
    return out


def linear_wgrad(dy: Tensor, x: Tensor, dw: Tensor, db: Optional[Tensor], accumulate: bool,
                 alpha: Optional[Tensor] = None) -> None:
    """dw (fp32 [N,K]) (+)= alpha * dy^T @ x ; db (fp32 [N]) (+)= alpha * colsum(dy).  (vy_linear_wgrad)
    alpha: optional fp32 device scalar (read by the kernel, no host sync)."""
    _need_gpu(dy, x, dw, db, alpha)
    assert alpha is None or (alpha.dtype == torch.float32 and alpha.numel() == 1)
    d2, x2 = _rows(dy), _rows(x)
    M, N = d2.shape
    K = x2.shape[1]
    assert dw.dtype == torch.float32 and dw.shape == (N, K) and dw.stride(1) == 1
    call("vy_linear_wgrad", d2.data_ptr(), d2.stride(0), x2.data_ptr(), x2.stride(0), dw.data_ptr(),
         dw.stride(0), _ptr(db), 1.0 if accumulate else 0.0, _ptr(alpha), M, N, K, dtype_code(dy.dtype), _stream())


def linear_wgrad_grouped(items) -> None:
    """items: up to 8 tuples (dy, x, dw, db|None): dw += dy^T @ x, db += colsum(dy), ONE launch
    (vy_linear_wgrad_grouped).  The tensors must stay alive until the launch has been enqueued (they do:
    the caller holds them)."""
    import ctypes as C
    n = len(items)
    assert 1 <= n <= 8
    arr = (_lib.VyWgradDesc * n)()
    dt = None
    for i, (dy, x, dw, db) in enumerate(items):
        _need_gpu(dy, x, dw, db)
        d2, x2 = _rows(dy), _rows(x)
        M, N = d2.shape
        K = x2.shape[1]
        assert x2.shape[0] == M and dw.dtype == torch.float32 and dw.shape == (N, K) and dw.stride(1) == 1
        assert db is None or (db.dtype == torch.float32 and db.numel() == N)
        assert dt is None or dt == dy.dtype
        dt = dy.dtype
        a = arr[i]
        a.dy, a.lddy, a.x, a.ldx = d2.data_ptr(), d2.stride(0), x2.data_ptr(), x2.stride(0)
        a.dw, a.lddw, a.db = dw.data_ptr(), dw.stride(0), _ptr(db)
        a.M, a.N, a.K = M, N, K
    call("vy_linear_wgrad_grouped", C.cast(arr, C.c_void_p), n, dtype_code(dt), _stream())


def layernorm_bwd(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, dgamma: Tensor,
                  dbeta: Tensor, accumulate: bool) -> Tensor:
    _need_gpu(dy, x, gamma, mean, rstd, dgamma, dbeta)
    d2, x2 = _rows(dy), _rows(x)
    M, N = x2.shape
    dx = torch.empty((M, N), dtype=x.dtype, device=x.device)
    W = _lib.load().vy_layernorm_bwd_ws_rows(M)
    ws = torch.empty((2 * W * N,), dtype=torch.float32, device=x.device)
    call("vy_layernorm_bwd", d2.data_ptr(), d2.stride(0), x2.data_ptr(), x2.stride(0), gamma.data_ptr(),
         mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dx.stride(0), dgamma.data_ptr(), dbeta.data_ptr(),
         1.0 if accumulate else 0.0, ws.data_ptr(), M, N, dtype_code(x.dtype), _stream())
    return dx.view(x.shape)


def attention_bwd(q, k, v, out, dout, lse, dq, dk, dv, *, causal: bool, start_pos: int = 0,
                  keypad: Optional[Tensor] = None, scale: Optional[float] = None,
                  cos: Optional[Tensor] = None, sin: Optional[Tensor] = None, rope_pos0: int = 0) -> None:
    """Flash attention backward; dq/dk/dv are (B, heads, L|S, dh) views written in place."""
    _need_gpu(q, k, v, out, dout, lse, dq, dk, dv, keypad)
    B, h, L, dh = q.shape
    hk, S = k.shape[1], k.shape[2]
    kind = (_lib.MASK_CAUSAL if causal else 0) | (_lib.MASK_KEYPAD if keypad is not None else 0)
    if scale is None:
        scale = 1.0 / math.sqrt(dh)
    assert out.stride() == dout.stride() and out.stride(2) == 1
    delta = torch.empty((B, h, L), dtype=torch.float32, device=q.device)
    call("vy_attn_bwd", q.data_ptr(), q.stride(0), q.stride(1), q.stride(2),
         k.data_ptr(), k.stride(0), k.stride(1), k.stride(2),
         v.data_ptr(), v.stride(0), v.stride(1), v.stride(2),
         out.data_ptr(), dout.data_ptr(), out.stride(0), out.stride(1), lse.data_ptr(), delta.data_ptr(),
         dq.data_ptr(), dq.stride(0), dq.stride(1), dq.stride(2),
         dk.data_ptr(), dk.stride(0), dk.stride(1), dk.stride(2),
         dv.data_ptr(), dv.stride(0), dv.stride(1), dv.stride(2),
         kind, start_pos, _ptr(keypad), keypad.stride(0) if keypad is not None else 0,
         _ptr(cos), _ptr(sin), rope_pos0,
         B, h, hk, L, S, dh, float(scale), dtype_code(q.dtype), _stream())


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, p_bf16: Optional[Tensor], lr: float,
               beta1: float, beta2: float, eps: float, weight_decay: float, step: int,
               grad_scale: float = 1.0, scale_dev: Optional[Tensor] = None, gate: Optional[Tensor] = None) -> None:
    """scale_dev: optional fp32 device scalar multiplied into grad_scale by the kernel (clip coefficient).
    gate: optional fp32 device scalar; when it is 0 the launch changes nothing (vy_adamw_step_gated)."""
    _need_gpu(p, g, m, v, p_bf16, scale_dev, gate)
    assert scale_dev is None or (scale_dev.dtype == torch.float32 and scale_dev.numel() == 1)
    if gate is None:
        call("vy_adamw_step", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(p_bf16), p.numel(),
             lr, beta1, beta2, eps, weight_decay, step, grad_scale, _ptr(scale_dev), _stream())
    else:
        assert gate.dtype == torch.float32 and gate.numel() == 1
        call("vy_adamw_step_gated", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(p_bf16), p.numel(),
             lr, beta1, beta2, eps, weight_decay, step, grad_scale, _ptr(scale_dev), gate.data_ptr(), _stream())


def sumsq(x: Tensor) -> Tensor:
    """sum(x^2) of a contiguous fp32 tensor as a device scalar (vy_sumsq; deterministic)."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = torch.empty(1024, dtype=torch.float32, device=x.device)
    call("vy_sumsq", x.data_ptr(), x.numel(), out.data_ptr(), ws.data_ptr(), _stream())
    return out[0]


def dropout(x: Tensor, p: float, seed: int, offset: int, out: Optional[Tensor] = None) -> Tensor:
    """x * keep(seed, offset, row, column) / (1 - p) over the (rows, last dim) view of x (vy_dropout): the
    mask vy_linear_dropout_fwd applied in its epilogue for the same (seed, offset)."""
    _need_gpu(x, out)
    x2 = _rows(x)
    if out is None:
        out = torch.empty_like(x2)
        ret = out.view(x.shape)
    else:
        ret = out
    o2 = _rows(out)
    call("vy_dropout", x2.data_ptr(), x2.stride(0), o2.data_ptr(), o2.stride(0), x2.shape[0], x2.shape[1],
         float(p), int(seed), int(offset), dtype_code(x.dtype), _stream())
    return ret


def transpose(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    _need_gpu(x)
    R, C = x.shape
    if out is None:
        out = torch.empty((C, R), dtype=x.dtype, device=x.device)
    call("vy_transpose", x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), R, C, dtype_code(x.dtype), _stream())
    return out


class TransposeBatch:
    """Descriptor table for vy_transpose_batched: many (src [R,C] -> dst [C,R]) pairs, one launch."""

    def __init__(self, pairs) -> None:
        import ctypes as C
        import numpy as np

        class Desc(C.Structure):
            _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("ldin", C.c_int64), ("ldout", C.c_int64),
                        ("R", C.c_int32), ("C", C.c_int32), ("tile0", C.c_int32), ("tiles_c", C.c_int32)]
        assert C.sizeof(Desc) == 48
        arr = (Desc * len(pairs))()
        tile0 = 0
        self.dtype = pairs[0][0].dtype
        for i, (src, dst) in enumerate(pairs):
            _need_gpu(src, dst)
            R, Cc = src.shape
            assert dst.shape == (Cc, R) and src.stride(1) == 1 and dst.stride(1) == 1 and src.dtype == self.dtype
            tc = (Cc + 63) // 64
            arr[i] = Desc(src.data_ptr(), dst.data_ptr(), src.stride(0), dst.stride(0), R, Cc, tile0, tc)
            tile0 += tc * ((R + 63) // 64)
        self.n, self.total = len(pairs), tile0
        self.keys = [(s_.data_ptr(), d_.data_ptr()) for s_, d_ in pairs]
        host = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy())
        self.table = host.to(pairs[0][0].device)

    def run(self) -> None:
        call("vy_transpose_batched", self.table.data_ptr(), self.n, self.total, dtype_code(self.dtype), _stream())


def cast(src: Tensor, dst: Tensor) -> Tensor:
    _need_gpu(src, dst)
    call("vy_cast", src.data_ptr(), dst.data_ptr(), src.numel(), dtype_code(src.dtype), dtype_code(dst.dtype), _stream())
    return dst


def act_bwd(dy: Tensor, pre: Tensor, act: int) -> Tensor:
    """dy * act'(pre).  (vy_act_bwd)"""
    _need_gpu(dy, pre)
    d2, p2 = _rows(dy), _rows(pre)
    M, N = p2.shape
    out = torch.empty((M, N), dtype=dy.dtype, device=dy.device)
    call("vy_act_bwd", d2.data_ptr(), d2.stride(0), p2.data_ptr(), p2.stride(0), out.data_ptr(), out.stride(0),
         M, N, act, dtype_code(dy.dtype), _stream())
    return out.view(pre.shape)


def xent_fwd(logits2d: Tensor, labels: Tensor, ignore_index: int, lse: Tensor, loss_sum: Tensor, count: Tensor,
             err_flag: Optional[Tensor] = None) -> None:
    """logits2d: (M, V) view with 16-byte aligned, padded rows; labels int64 (M,).  err_flag: device int32,
    set to 1 when a label is neither ignore_index nor in [0, V) (such rows count as ignored)."""
    _need_gpu(logits2d, labels, lse, loss_sum, count, err_flag)
    M, V = logits2d.shape
    call("vy_xent_fwd", logits2d.data_ptr(), logits2d.stride(0), labels.data_ptr(), ignore_index, lse.data_ptr(),
         loss_sum.data_ptr(), count.data_ptr(), M, V, _ptr(err_flag), dtype_code(logits2d.dtype), _stream())


def xent_bwd_(logits2d: Tensor, labels: Tensor, ignore_index: int, lse: Tensor, gscale: Tensor, count: Tensor) -> None:
    """Overwrite logits with d loss / d logits (in place)."""
    _need_gpu(logits2d, labels, lse, gscale, count)
    M, V = logits2d.shape
    call("vy_xent_bwd", logits2d.data_ptr(), logits2d.stride(0), labels.data_ptr(), ignore_index, lse.data_ptr(),
         gscale.data_ptr(), count.data_ptr(), M, V, dtype_code(logits2d.dtype), _stream())


def embedding(table: Tensor, ids: Tensor, err_flag: Optional[Tensor] = None) -> Tensor:
    """out[..., :] = table[ids[...], :]  (vy_embedding_fwd).  table: (V, d) bf16/fp32; ids int64."""
    _need_gpu(table, ids, err_flag)
    assert ids.dtype == torch.long and table.dim() == 2 and table.stride(1) == 1
    idc = ids.contiguous()
    V, d = table.shape
    out = torch.empty((*ids.shape, d), dtype=table.dtype, device=table.device)
    call("vy_embedding_fwd", table.data_ptr(), table.stride(0), idc.data_ptr(), out.data_ptr(), d, idc.numel(), d, V,
         _ptr(err_flag), dtype_code(table.dtype), _stream())
    return out


def greedy_step_(logits: Tensor, tokens: Tensor, cur_pos: int, text_mask: Optional[Tensor], eos_ids: Tensor,
                 eos_reached: Tensor, not_done: Optional[Tensor] = None) -> None:
    """One generated position of the greedy loop (vy_greedy_step): tokens[:, cur_pos] <- forced prompt token or
    argmax(logits); eos_reached |= new EOS; not_done (one int32) += rows still running."""
    _need_gpu(logits, tokens, text_mask, eos_ids, eos_reached, not_done)
    assert logits.dim() == 2 and logits.stride(1) == 1 and tokens.dtype == torch.long and tokens.stride(1) == 1
    assert eos_reached.dtype in (torch.bool, torch.uint8) and eos_reached.is_contiguous()
    assert eos_ids.dtype == torch.long and eos_ids.is_contiguous()
    B, V = logits.shape
    assert tokens.shape[0] == B and eos_reached.numel() == B and 0 <= cur_pos < tokens.shape[1]
    if text_mask is not None:
        assert text_mask.dtype in (torch.bool, torch.uint8) and text_mask.stride(1) == 1 and text_mask.shape == tokens.shape
    if not_done is not None:
        assert not_done.dtype == torch.int32 and not_done.numel() == 1
    call("vy_greedy_step", logits.data_ptr(), logits.stride(0), B, V, dtype_code(logits.dtype), tokens.data_ptr(),
         tokens.stride(0), int(cur_pos), _ptr(text_mask), 0 if text_mask is None else text_mask.stride(0),
         eos_ids.data_ptr(), eos_ids.numel(), eos_reached.data_ptr(), _ptr(not_done), _stream())


def sampling_probs(logits: Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 0.0) -> Tensor:
    """fp32 softmax(mask(logits) / temperature) over the last dimension (vy_sampling_probs):
    top_k > 0 and 0 < top_p < 1 mask as TopKProcessor / NucleusProcessor of the reference do."""
    _need_gpu(logits)
    V = logits.shape[-1]
    l2 = logits.reshape(-1, V)
    if l2.stride(1) != 1:
        l2 = l2.contiguous()
    probs = torch.empty((l2.shape[0], V), dtype=torch.float32, device=logits.device)
    if l2.shape[0] == 0:   # no rows (the last round of speculative decoding verifies zero drafts)
        return probs.view(*logits.shape)
    call("vy_sampling_probs", l2.data_ptr(), l2.stride(0), l2.shape[0], V, dtype_code(l2.dtype), float(temperature),
         int(top_k), float(top_p), probs.data_ptr(), probs.stride(0), _stream())
    return probs.view(*logits.shape)


def embedding_bwd_(dout: Tensor, ids: Tensor, dw: Tensor, padding_idx: Optional[int]) -> None:
    """dw[ids[m], :] += dout[m, :] in fp32, skipping padding_idx  (vy_embedding_bwd)."""
    _need_gpu(dout, ids, dw)
    d2 = _rows(dout)
    idc = ids.contiguous()
    assert dw.dtype == torch.float32 and dw.stride(1) == 1 and idc.numel() == d2.shape[0]
    call("vy_embedding_bwd", d2.data_ptr(), d2.stride(0), idc.data_ptr(), dw.data_ptr(), dw.stride(0),
         -1 if padding_idx is None else int(padding_idx), d2.shape[0], d2.shape[1], dw.shape[0],
         dtype_code(dout.dtype), _stream())


def xent_fused_(logits2d: Tensor, labels: Tensor, ignore_index: int, lse: Tensor, loss_sum: Tensor,
                count: Tensor, gscale: Tensor, err_flag: Optional[Tensor] = None) -> None:
    """One pass: lse / loss_sum as xent_fwd, then logits <- d loss / d logits in place (vy_xent_fused).
    count (#rows with label != ignore) is an input here."""
    _need_gpu(logits2d, labels, lse, loss_sum, count, gscale, err_flag)
    M, V = logits2d.shape
    call("vy_xent_fused", logits2d.data_ptr(), logits2d.stride(0), labels.data_ptr(), ignore_index, lse.data_ptr(),
         loss_sum.data_ptr(), count.data_ptr(), gscale.data_ptr(), M, V, _ptr(err_flag), dtype_code(logits2d.dtype),
         _stream())


def rmsnorm(x: Tensor, w: Tensor, eps: float, w_offset: float = 1.0) -> Tensor:
    """x * rsqrt(mean x^2 + eps) * (w_offset + w)  (vy_rmsnorm_fwd; Gemma: w_offset = 1)."""
    _need_gpu(x, w)
    x2 = _rows(x)
    M, N = x2.shape
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    call("vy_rmsnorm_fwd", x2.data_ptr(), x2.stride(0), w.data_ptr(), y.data_ptr(), y.stride(0), M, N, float(eps),
         float(w_offset), dtype_code(x.dtype), _stream())
    return y.view(x.shape)


def gated_act(gate_up: Tensor, act: int) -> Tensor:
    """act(gate_up[..., :I]) * gate_up[..., I:]  (vy_gated_act_fwd)."""
    _need_gpu(gate_up)
    g2 = _rows(gate_up)
    M, two_i = g2.shape
    I = two_i // 2
    out = torch.empty((M, I), dtype=gate_up.dtype, device=gate_up.device)
    call("vy_gated_act_fwd", g2.data_ptr(), g2.stride(0), out.data_ptr(), out.stride(0), M, I, act,
         dtype_code(gate_up.dtype), _stream())
    return out.view(*gate_up.shape[:-1], I)
