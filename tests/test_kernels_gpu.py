"""Kernel-level parity: every C-ABI entry point vs the CPU oracle / a plain fp32-fp64 torch
restatement of the same op, on the same seeded inputs.  Runs on the MI355X box (-m gpu).

Tolerances: f32 kernels 1e-5 (north_star); bf16 kernels are compared with the oracle evaluated on
the bf16-rounded inputs, bar = a few bf16 ulps of the output scale.
"""
import math

import numpy as np
import pytest
import torch

from oracle import vyom_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from vyomai_amd import ops, _lib
    return ops, _lib


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed + 1000 * len(shape) + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def check(got, want, atol, rtol=0.0, what=""):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - want).abs()
    bound = atol + rtol * want.abs()
    bad = err > bound
    if bad.any():
        idx = bad.nonzero()[0].tolist()
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.numel()} off; max err {err.max():.3e} (atol {atol}); "
            f"first at {idx}: got {got[tuple(idx)]:.6f} want {want[tuple(idx)]:.6f}")


ACTS = {0: lambda x: x, 1: O.gelu_erf, 2: O.gelu_tanh}


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 768), (51, 1003, 768), (300, 768, 72),
                                   (16, 768, 768), (1, 256, 3072), (1024, 3072, 768), (640, 768, 3072),
                                   (32, 3072, 768), (32, 1003, 768), (7, 768, 3072), (32, 64, 40),
                                   # <= 4 rows: the matrix-vector kernel (one wave per output column)
                                   (1, 2048, 16384), (2, 1003, 768), (3, 770, 264), (4, 768, 3072), (1, 5, 8),
                                   # > 1024 rows: the 256 x 192 tile kernel (M / N / K tails)
                                   (2000, 768, 768), (4096, 1003, 768), (1500, 384, 72), (3072, 3072, 768),
                                   # mid-size M with few tiles: K split over workgroups (fp32 partial tiles + finish launch):
                                   # PaliGemma-shape prefill (264 rows; Gemma out-projection, SigLIP fc1 / fc2 with a K tail) and
                                   # the 2112 rows of the captioning decoder (configs[3])
                                   (264, 2048, 2048), (264, 2048, 16384), (256, 4304, 1152), (256, 1152, 4304), (2112, 768, 3072),
                                   (2112, 768, 768)])
@pytest.mark.parametrize("act", [0, 1])
def test_linear_bf16(M, N, K, act):
    ops, _ = _ops()
    x = rnd(M, K, seed=1).bfloat16()
    w = rnd(N, K, seed=2, scale=1 / math.sqrt(K)).bfloat16()
    b = rnd(N, seed=3, scale=0.1).bfloat16()
    r = rnd(M, N, seed=4).bfloat16()
    pre_ref = x.double() @ w.double().t() + b.double()
    want = ACTS[act](pre_ref) + r.double()
    ldy = (N + 7) // 8 * 8
    pre = torch.zeros(M, ldy, dtype=torch.bfloat16, device=DEV)
    y = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), act=act, residual=r.to(DEV), pre_out=pre[:, :N])
    torch.cuda.synchronize()
    check(y, want, 3e-2, 1e-2, f"linear bf16 {M}x{N}x{K} act{act}")
    check(pre[:, :N], pre_ref, 3e-2, 1e-2, "pre_out")
    # no bias / no residual / no pre path
    y2 = ops.linear(x.to(DEV), w.to(DEV))
    check(y2, x.double() @ w.double().t(), 3e-2, 1e-2, "linear bf16 plain")


def test_linear_bf16_wide_mid_m_and_split_k_equals_unsplit():
    """(264, 32768, 2048): Gemma's gate/up projection at prefill (768 tiles: not split).  And the split launches against
    the same GEMM without a workspace (one workgroup per tile walks all of K): equal up to the fp32 summation order."""
    ops, lib = _ops()
    M, N, K = 264, 32768, 2048
    x = rnd(M, K, seed=1).bfloat16().to(DEV)
    w = rnd(N, K, seed=2, scale=1 / math.sqrt(K)).bfloat16().to(DEV)
    y = ops.linear(x, w)
    cols = torch.arange(0, N, 37, device=DEV)
    check(y[:, cols], x.double() @ w[cols].double().t(), 3e-2, 1e-2, "gate/up shape")
    from vyomai_amd._lib import call
    for (M, N, K) in [(264, 2048, 16384), (256, 1152, 4304), (2112, 768, 3072)]:
        x = rnd(M, K, seed=3).bfloat16().to(DEV)
        w = rnd(N, K, seed=4, scale=1 / math.sqrt(K)).bfloat16().to(DEV)
        b = rnd(N, seed=5, scale=0.1).bfloat16().to(DEV)
        r = rnd(M, N, seed=6).bfloat16().to(DEV)
        split = ops.linear(x, w, b, act=1, residual=r)
        st = torch.cuda.current_stream()
        call("vy_workspace_set", st.cuda_stream, None, 0)            # no workspace: the one-launch kernel
        ops._WS.pop((st.device_index, st.cuda_stream), None)
        try:
            import vyomai_amd.ops as O_
            saved, O_._mid_ws = O_._mid_ws, lambda rows: None
            whole = ops.linear(x, w, b, act=1, residual=r)
        finally:
            O_._mid_ws = saved
        d = (split.float() - whole.float()).abs().max().item()
        assert d <= 2 ** -6 * max(1.0, whole.float().abs().max().item()), (M, N, K, d)   # a bf16 rounding step at most


@pytest.mark.parametrize("M,N,K", [(51, 1003, 768), (130, 64, 16), (64, 64, 4), (257, 192, 260)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_f32(M, N, K, act):
    ops, _ = _ops()
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K))
    b, r = rnd(N, seed=3, scale=0.1), rnd(M, N, seed=4)
    want = ACTS[act](x.double() @ w.double().t() + b.double()) + r.double()
    y = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), act=act, residual=r.to(DEV))
    check(y, want, 1e-5, 1e-5, f"linear f32 {M}x{N}x{K}")


def _qkv_ref(x, w, b, h, hk, dh, pos0, rope, dtype):
    """oracle: three projections -> split heads -> apply_rotary in `dtype` like the reference."""
    B, L, _ = x.shape
    y = (x.double() @ w.double().t() + b.double()).to(dtype)
    q, k, v = y.split([h * dh, hk * dh, hk * dh], dim=-1)
    q, k, v = O.split_heads(q, dh), O.split_heads(k, dh), O.split_heads(v, dh)
    if rope:
        fr = O.rotary_angles(dh, pos0 + L + 1)[:, pos0:pos0 + L]
        q, k = O.apply_rotary(q, k, fr)
    return q, k, v


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,L,K,h,hk,dh,rope", [(2, 96, 768, 12, 4, 64, True), (2, 17, 64, 4, 2, 16, True),
                                               (3, 50, 768, 12, 12, 64, False), (2, 1, 768, 12, 4, 64, True),
                                               (1, 40, 256, 2, 1, 128, True), (4, 300, 768, 12, 4, 64, True),
                                               (32, 1, 768, 12, 12, 64, True), (5, 1, 768, 12, 4, 64, False),
                                               (3, 700, 768, 12, 12, 64, True),
                                               # <= 4 rows without the fused rotary epilogue: the matrix-vector kernel
                                               (1, 1, 2048, 8, 1, 256, True), (3, 1, 256, 2, 1, 128, False),
                                               (4, 1, 768, 12, 4, 64, False), (2, 2, 1152, 16, 16, 72, False)])
def test_qkv_rope(dtype, B, L, K, h, hk, dh, rope):
    ops, _ = _ops()
    pos0 = 5
    N = (h + 2 * hk) * dh
    x = rnd(B, L, K, seed=1).to(dtype)
    w = rnd(N, K, seed=2, scale=1 / math.sqrt(K)).to(dtype)
    b = rnd(N, seed=3, scale=0.1).to(dtype)
    qr, kr, vr = _qkv_ref(x, w, b, h, hk, dh, pos0, rope, dtype)
    cos = sin = None
    if rope:
        cos, sin = ops.rope_tables(dh, 64 + L, DEV)
    q = torch.zeros(B, h, L, dh, dtype=dtype, device=DEV)
    # k/v written straight into a (B, hk, maxlen, dh) static cache at pos0
    kc = torch.zeros(B, hk, pos0 + L + 3, dh, dtype=dtype, device=DEV)
    vc = torch.zeros_like(kc)
    ops.qkv_rope(x.to(DEV), w.to(DEV), b.to(DEV), h, hk, dh, cos, sin, pos0, q,
                 kc[:, :, pos0:pos0 + L], vc[:, :, pos0:pos0 + L])
    tol = 4e-2 if dtype == torch.bfloat16 else 1e-5
    check(q, qr, tol, tol, "q")
    check(kc[:, :, pos0:pos0 + L], kr, tol, tol, "k (cache slice)")
    check(vc[:, :, pos0:pos0 + L], vr, tol, tol, "v (cache slice)")
    assert float(kc[:, :, :pos0].abs().max()) == 0 and float(kc[:, :, pos0 + L:].abs().max()) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N", [(51, 768), (4, 64), (1000, 2048), (33, 1152)])
def test_layernorm(dtype, M, N):
    ops, _ = _ops()
    x = (rnd(M, N, seed=1) * 2 + 0.5).to(dtype)
    g = (1 + 0.1 * rnd(N, seed=2)).to(dtype)
    b = (0.1 * rnd(N, seed=3)).to(dtype)
    want = torch.nn.functional.layer_norm(x.double(), (N,), g.double(), b.double(), 1e-5)
    y, mean, rstd = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, save_stats=True)
    tol = 3e-2 if dtype == torch.bfloat16 else 2e-6
    check(y, want, tol, tol, "layernorm")
    check(mean, x.double().mean(-1), 1e-5, 1e-5, "mean")
    check(rstd, (x.double().var(-1, unbiased=False) + 1e-5).rsqrt(), 1e-5, 1e-5, "rstd")


def _dense_mask(B, L, S, causal, start_pos, keypad, addmask):
    m = torch.zeros(B, 1, L, S)
    fmin = torch.finfo(torch.float32).min
    if causal:
        i = torch.arange(L)[:, None]
        j = torch.arange(S)[None, :]
        m = m + torch.where(j <= i + start_pos, 0.0, fmin)[None, None]
    if keypad is not None:
        m = m + (1.0 - keypad[:, None, None, :].float()) * fmin
    m = m.clamp_min(fmin)  # (1-c*k)*min in the reference is one factor: never -inf
    if addmask is not None:
        m = m + addmask
    return m


ATTN_CASES = [
    # B, h, hk, L, S, dh, causal, start, keypad, additive
    (2, 4, 2, 200, 200, 64, True, 0, False, False),
    (2, 4, 4, 128, 128, 64, False, 0, False, False),
    (2, 4, 2, 130, 130, 64, True, 0, True, False),
    (2, 12, 4, 40, 100, 64, True, 60, True, False),
    (1, 2, 2, 70, 70, 64, False, 0, True, False),
    (2, 2, 1, 65, 65, 64, False, 0, False, True),
    (2, 3, 3, 512, 512, 64, True, 0, False, False),
    (1, 2, 1, 150, 150, 128, True, 0, False, False),
    (2, 2, 2, 64, 64, 128, False, 0, True, False),
    (2, 4, 2, 17, 17, 16, True, 0, True, False),   # row-wise kernel (dh 16)
    (3, 12, 12, 197, 197, 64, False, 0, False, False),  # ViT shape
    # other head widths: the general MFMA kernel (bf16) / the row-wise kernel (fp32)
    (1, 16, 16, 256, 256, 72, False, 0, False, False),   # SigLIP-So400m (Examples/paligemma.ipynb cell 9)
    (2, 8, 1, 264, 264, 256, True, 0, False, False),     # Gemma-2B prefill, MQA (cell 12)
    (2, 8, 1, 70, 110, 256, True, 40, True, False),      # chunked prefill with padding (left padding: dead rows)
    (2, 3, 3, 100, 100, 96, False, 0, True, False),
    (1, 2, 2, 130, 130, 160, True, 0, False, False),
    (2, 4, 2, 33, 33, 72, True, 0, True, False),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_fwd(dtype, case):
    ops, _ = _ops()
    B, h, hk, L, S, dh, causal, start, use_kp, use_add = case
    q = rnd(B, h, L, dh, seed=1).to(dtype)
    k = rnd(B, hk, S, dh, seed=2).to(dtype)
    v = rnd(B, hk, S, dh, seed=3).to(dtype)
    keypad = None
    if use_kp:
        keypad = torch.ones(B, S, dtype=torch.uint8)
        keypad[0, S - S // 3:] = 0            # right padding
        if B > 1:
            keypad[1, : min(5, S - 1)] = 0    # LEFT padding: with a causal mask rows 0..4 are fully masked
    addmask = None
    if use_add:
        addmask = rnd(B, 1, L, S, seed=5)
        addmask[:, :, :, ::7] = torch.finfo(torch.float32).min
    mask = _dense_mask(B, L, S, causal, start, keypad, addmask)
    want = O.merge_heads(O.sdpa(q.float(), O.repeat_kv(k.float(), h // hk), O.repeat_kv(v.float(), h // hk), mask))
    lse = torch.zeros(B, h, L, dtype=torch.float32, device=DEV)
    got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), causal=causal, start_pos=start,
                        keypad=keypad.to(DEV) if keypad is not None else None,
                        addmask=addmask.to(DEV) if addmask is not None else None, lse=lse)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    check(got, want, tol, tol, f"attention {case}")
    s = (q.float() @ O.repeat_kv(k.float(), h // hk).transpose(-1, -2)) / math.sqrt(dh) + mask
    live = (mask > torch.finfo(torch.float32).min / 2).any(-1).expand(B, h, L)  # skip fully masked rows
    ref_lse = torch.logsumexp(s.double(), -1)
    check(torch.where(live, lse.cpu().double(), 0.0), torch.where(live, ref_lse, 0.0),
          2e-2 if dtype == torch.bfloat16 else 1e-4, 1e-4, "lse")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,h,hk,S,dh", [(3, 12, 12, 37, 64), (32, 12, 4, 640, 64), (2, 8, 1, 300, 256), (1, 4, 2, 1, 16)])
def test_attention_decode(dtype, B, h, hk, S, dh):
    ops, _ = _ops()
    q = rnd(B, h, 1, dh, seed=1).to(dtype)
    kc = rnd(B, hk, S + 9, dh, seed=2).to(dtype)
    vc = rnd(B, hk, S + 9, dh, seed=3).to(dtype)
    want = O.merge_heads(O.sdpa(q.float(), O.repeat_kv(kc[:, :, :S].float(), h // hk),
                                O.repeat_kv(vc[:, :, :S].float(), h // hk), None))
    got = ops.attention_decode(q.to(DEV), kc.to(DEV), vc.to(DEV), S)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    check(got, want, tol, tol, "decode")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_rope_standalone_and_inverse(dtype):
    ops, _ = _ops()
    B, H, L, dh, pos0 = 2, 3, 19, 64, 7
    x = rnd(B, H, L, dh, seed=1).to(dtype)
    cos, sin = ops.rope_tables(dh, 64, DEV)
    fr = O.rotary_angles(dh, 64)[:, pos0:pos0 + L]
    want, _ = O.apply_rotary(x, x, fr)
    y = ops.rope_(x.to(DEV).clone(), cos, sin, pos0)
    # the standalone kernel mirrors the reference's per-op rounding
    check(y, want, 1e-6 if dtype == torch.float32 else 1e-2, 0, "rope")
    back = ops.rope_(y.clone(), cos, sin, pos0, inverse=True)
    # two roundings per direction in bf16: allow 3 ulps
    check(back, x, 1e-2 if dtype == torch.bfloat16 else 1e-6, 1.2e-2 if dtype == torch.bfloat16 else 0,
          "rope inverse round trip")
