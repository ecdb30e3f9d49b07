"""Backward kernels vs torch autograd on the CPU oracle functions (fp64/fp32), bf16 tolerances."""
import math

import pytest
import torch

from oracle import vyom_oracle as O
from tests.test_kernels_gpu import check, rnd, _dense_mask

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _ops():
    from vyomai_amd import ops
    return ops


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (1000, 768, 768), (4096, 3072, 768), (777, 768, 3072), (64, 2304, 768)])
def test_wgrad(M, N, K):
    ops = _ops()
    dy, x = rnd(M, N, seed=1).to(BF), rnd(M, K, seed=2).to(BF)
    dw = torch.full((N, K), 7.0, dtype=torch.float32, device=DEV)
    db = torch.full((N,), 7.0, dtype=torch.float32, device=DEV)
    ops.linear_wgrad(dy.to(DEV), x.to(DEV), dw, db, accumulate=False)
    want = dy.double().t() @ x.double()
    tol = 2e-3 * math.sqrt(M)
    check(dw, want, tol, 1e-3, "dW")
    check(db, dy.double().sum(0), tol, 1e-3, "db")
    ops.linear_wgrad(dy.to(DEV), x.to(DEV), dw, db, accumulate=True)
    check(dw, 2 * want, 2 * tol, 1e-3, "dW accumulate")


def test_wgrad_grouped():
    """vy_linear_wgrad_grouped: several accumulating weight gradients in one launch (256 x 256 tiles), ragged
    N / K / M, with and without a bias gradient, rows of the operands strided."""
    ops = _ops()
    shapes = [(4096, 768, 768, True), (4096, 3072, 768, True), (4096, 768, 3072, False), (4100, 2304, 768, True),
              (1000, 520, 264, True), (300, 56, 8, False)]
    items, wants = [], []
    for i, (M, N, K, bias) in enumerate(shapes):
        dy_full = rnd(M, N + 8, seed=10 + i).to(BF).to(DEV)       # row stride N + 8
        dy, x = dy_full[:, :N], rnd(M, K, seed=30 + i).to(BF).to(DEV)
        dw = torch.full((N, K), 3.0, dtype=torch.float32, device=DEV)
        db = torch.full((N,), 3.0, dtype=torch.float32, device=DEV) if bias else None
        items.append((dy, x, dw, db))
        wants.append((dy.double().t() @ x.double(), dy.double().sum(0), M))
    ops.linear_wgrad_grouped(items)
    for (dy, x, dw, db), (w_dw, w_db, M) in zip(items, wants):
        tol = 2e-3 * math.sqrt(M)
        check(dw, w_dw.cpu() + 3.0, tol, 1e-3, "grouped dW (accumulated onto 3.0)")
        if db is not None:
            check(db, w_db.cpu() + 3.0, tol, 1e-3, "grouped db")
    ops.linear_wgrad_grouped(items[:1])      # a group of one
    check(items[0][2], 2 * wants[0][0].cpu() + 3.0, 4e-3 * math.sqrt(4096), 1e-3, "second accumulation")
    from vyomai_amd._lib import VyomHipError
    with pytest.raises(VyomHipError):
        ops.linear_wgrad_grouped([(items[0][0][:, :767], items[0][1][:, :767], items[0][2][:767, :767], None)])   # K % 8


@pytest.mark.parametrize("M,N,K,act", [(300, 768, 3072, 1), (1024, 768, 768, 0), (51, 3072, 768, 0),
                                       # split-K launches (mid-size M, few tiles): the decoder rows of configs[3], a 264-row prefill
                                       (2112, 3072, 768, 0), (2112, 768, 3072, 1), (264, 2048, 2048, 0), (264, 16384, 2048, 0)])
def test_dgrad(M, N, K, act):
    """dX[M,K] = dY[M,N] @ W[N,K] (* gelu'(pre)) (+ add)."""
    ops = _ops()
    dy = rnd(M, N, seed=1).to(BF)
    w = (rnd(N, K, seed=2) / math.sqrt(N)).to(BF)
    pre = rnd(M, K, seed=3).to(BF)
    add = rnd(M, K, seed=4).to(BF)
    want = dy.double() @ w.double()
    if act:
        p = pre.double().requires_grad_(True)
        O.gelu_erf(p).sum().backward()
        want = want * p.grad
    want = want + add.double()
    wt = ops.transpose(w.to(DEV))
    check(wt, w.t(), 0, 0, "transpose")
    got = ops.linear_dgrad(dy.to(DEV), wt, pre.to(DEV) if act else None, act, add.to(DEV))
    check(got, want, 4e-2, 1e-2, "dgrad")


@pytest.mark.parametrize("M,N", [(1000, 768), (64, 64), (5000, 768)])
def test_layernorm_bwd(M, N):
    ops = _ops()
    x = (rnd(M, N, seed=1) * 2 + 0.3).to(BF)
    g = (1 + 0.1 * rnd(N, seed=2)).to(BF)
    b = (0.1 * rnd(N, seed=3)).to(BF)
    dy = rnd(M, N, seed=4).to(BF)
    xd, gd, bd = x.double().requires_grad_(True), g.double().requires_grad_(True), b.double().requires_grad_(True)
    (torch.nn.functional.layer_norm(xd, (N,), gd, bd, 1e-5) * dy.double()).sum().backward()
    y, mean, rstd = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, save_stats=True)
    dg = torch.zeros(N, dtype=torch.float32, device=DEV)
    dbt = torch.zeros(N, dtype=torch.float32, device=DEV)
    dx = ops.layernorm_bwd(dy.to(DEV), x.to(DEV), g.to(DEV), mean, rstd, dg, dbt, accumulate=False)
    check(dx, xd.grad, 3e-2, 1e-2, "dx")
    check(dg, gd.grad, 2e-3 * math.sqrt(M), 1e-3, "dgamma")
    check(dbt, bd.grad, 2e-3 * math.sqrt(M), 1e-3, "dbeta")


BWD_CASES = [
    # B, h, hk, L, S, causal, start, keypad
    (2, 4, 4, 128, 128, True, 0, False),
    (2, 4, 2, 200, 200, True, 0, False),
    (1, 2, 2, 96, 96, False, 0, False),
    (2, 4, 2, 130, 130, True, 0, True),
    (2, 3, 1, 70, 70, False, 0, True),
    (2, 12, 12, 512, 512, True, 0, False),
    # ViT-B/16: 197 tokens, non-causal (VisionAttention, reference layers/attention.py:591-624)
    (3, 12, 12, 197, 197, False, 0, False),
    # odd length, non-causal, key padding (encoder attention under a padding mask)
    (2, 12, 4, 131, 131, False, 0, True),
    # the other head widths (general kernels: SigLIP 72, 128, Gemma 256, and odd multiples of 8)
    (2, 4, 2, 130, 130, True, 0, True, 72),
    (1, 16, 16, 256, 256, False, 0, False, 72),
    (2, 4, 4, 128, 128, True, 0, False, 128),
    (2, 4, 2, 200, 200, True, 0, True, 128),
    (2, 3, 1, 70, 70, False, 0, True, 128),
    (1, 8, 1, 264, 264, True, 0, False, 256),
    (2, 2, 2, 97, 97, False, 0, True, 256),
    (2, 4, 2, 150, 150, True, 0, False, 96),
    (1, 2, 1, 65, 65, True, 0, False, 160),
    (1, 2, 2, 64, 64, False, 0, False, 40),
    # queries behind a cached prefix (L != S, start_pos): tuned and general kernels
    (2, 12, 4, 40, 100, True, 60, True),
    (1, 4, 4, 200, 328, True, 128, False),
    (2, 4, 2, 40, 100, True, 60, True, 128),
    (1, 4, 1, 70, 200, True, 130, False, 256),
    (2, 2, 2, 33, 77, False, 0, True, 72),
]


@pytest.mark.parametrize("case", BWD_CASES)
def test_attention_bwd(case):
    ops = _ops()
    B, h, hk, L, S, causal, start, use_kp = case[:8]
    dh = case[8] if len(case) > 8 else 64
    q = rnd(B, h, L, dh, seed=1).to(BF)
    k = rnd(B, hk, S, dh, seed=2).to(BF)
    v = rnd(B, hk, S, dh, seed=3).to(BF)
    do = rnd(B, L, h * dh, seed=4).to(BF)
    keypad = None
    if use_kp:
        keypad = torch.ones(B, S, dtype=torch.uint8)
        keypad[0, S - S // 3:] = 0
        if B > 1:
            keypad[1, S - 7:] = 0
    mask = _dense_mask(B, L, S, causal, start, keypad, None)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    s = qd @ O.repeat_kv(kd, h // hk).transpose(-1, -2) / math.sqrt(dh) + mask.double()
    out_ref = O.merge_heads(torch.softmax(s, -1) @ O.repeat_kv(vd, h // hk))
    (out_ref * do.double()).sum().backward()
    qg, kg, vg = q.to(DEV), k.to(DEV), v.to(DEV)
    lse = torch.zeros(B, h, L, dtype=torch.float32, device=DEV)
    out = ops.attention(qg, kg, vg, causal=causal, start_pos=start,
                        keypad=keypad.to(DEV) if keypad is not None else None, lse=lse)
    # dq/dk/dv written into one packed (B, L, (h+2hk)*dh) buffer, the layout the QKV dgrad consumes
    W = (h + 2 * hk) * dh
    packed = torch.zeros(B, max(L, S), W, dtype=BF, device=DEV)
    dq = packed[:, :L, : h * dh].view(B, L, h, dh).permute(0, 2, 1, 3)
    dk = packed[:, :S, h * dh:(h + hk) * dh].view(B, S, hk, dh).permute(0, 2, 1, 3)
    dv = packed[:, :S, (h + hk) * dh:].view(B, S, hk, dh).permute(0, 2, 1, 3)
    ops.attention_bwd(qg, kg, vg, out, do.to(DEV), lse, dq, dk, dv, causal=causal, start_pos=start,
                      keypad=keypad.to(DEV) if keypad is not None else None)
    check(out, out_ref, 2e-2, 2e-2, "fwd out")
    check(dq, qd.grad, 4e-2, 3e-2, "dq")
    check(dk, kd.grad, 4e-2, 3e-2, "dk")
    check(dv, vd.grad, 4e-2, 3e-2, "dv")


def test_adamw_matches_torch():
    ops = _ops()
    n = 100003
    p0, g = rnd(n, seed=1), rnd(n, seed=2)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    p = p0.clone().to(DEV)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    pb = torch.zeros(n, dtype=BF, device=DEV)
    for step in range(1, 4):
        pt.grad = g.clone() * step
        opt.step()
        ops.adamw_step(p, (g * step).to(DEV), m, v, pb, 1e-3, 0.9, 0.999, 1e-8, 0.01, step)
    check(p, pt.detach(), 1e-6, 1e-5, "adamw")
    check(pb, pt.detach().to(BF), 1e-2, 1e-2, "adamw bf16 shadow")


@pytest.mark.parametrize("dh", [64, 128, 256])
def test_attention_bwd_fused_rope_inverse(dh):
    """dq/dk with the RoPE backward inside vy_attn_bwd (fused into the epilogues at dh = 64, rotation launches after
    the general kernels otherwise) == separate inverse rotation."""
    ops = _ops()
    B, h, hk, L = 2, 4, 2, 96
    q, k, v = rnd(B, h, L, dh, seed=1).to(BF), rnd(B, hk, L, dh, seed=2).to(BF), rnd(B, hk, L, dh, seed=3).to(BF)
    do = rnd(B, L, h * dh, seed=4).to(BF)
    cos, sin = ops.rope_tables(dh, 128, DEV)
    qg, kg, vg = q.to(DEV), k.to(DEV), v.to(DEV)
    lse = torch.zeros(B, h, L, dtype=torch.float32, device=DEV)
    out = ops.attention(qg, kg, vg, causal=True, lse=lse)
    def run(fused):
        dq = torch.zeros(B, h, L, dh, dtype=BF, device=DEV)
        dk = torch.zeros(B, hk, L, dh, dtype=BF, device=DEV)
        dv = torch.zeros_like(dk)
        if fused:
            ops.attention_bwd(qg, kg, vg, out, do.to(DEV), lse, dq, dk, dv, causal=True, cos=cos, sin=sin, rope_pos0=3)
        else:
            ops.attention_bwd(qg, kg, vg, out, do.to(DEV), lse, dq, dk, dv, causal=True)
            ops.rope_(dq, cos, sin, 3, inverse=True)
            ops.rope_(dk, cos, sin, 3, inverse=True)
        return dq, dk, dv
    a, b = run(True), run(False)
    check(a[0], b[0], 2e-2, 2e-2, "dq")
    check(a[1], b[1], 2e-2, 2e-2, "dk")
    check(a[2], b[2], 0, 0, "dv")


# ---- fp32: the parity path has a backward too (plain-FMA kernels) ---------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (1000, 768, 520), (77, 50265, 64), (4100, 72, 3072)])
def test_wgrad_fp32(M, N, K):
    ops = _ops()
    ld = (N + 7) // 8 * 8
    dy_full = torch.zeros(M, ld)
    dy_full[:, :N] = rnd(M, N, seed=1)
    x = rnd(M, K, seed=2)
    dyg = dy_full.to(DEV)[:, :N]
    dw = torch.full((N, K), 7.0, dtype=torch.float32, device=DEV)
    db = torch.full((N,), 7.0, dtype=torch.float32, device=DEV)
    ops.linear_wgrad(dyg, x.to(DEV), dw, db, accumulate=False)
    want = (dy_full[:, :N].double().t() @ x.double())
    tol = 2e-6 * math.sqrt(M) * 4
    check(dw, want, tol, 1e-5, "dW fp32")
    check(db, dy_full[:, :N].double().sum(0), tol, 1e-5, "db fp32")
    alpha = torch.tensor([0.5], dtype=torch.float32, device=DEV)
    ops.linear_wgrad(dyg, x.to(DEV), dw, db, accumulate=True, alpha=alpha)
    check(dw, 1.5 * want, 2 * tol, 1e-5, "dW accumulate with alpha")
    ops.linear_wgrad_grouped([(dyg, x.to(DEV), dw, None)])
    check(dw, 2.5 * want, 3 * tol, 1e-5, "grouped entry point, fp32")


BWD_F32_CASES = [
    # B, h, hk, L, S, causal, start, keypad ('right' / 'left' = rows without a visible key), dh
    (2, 4, 4, 128, 128, True, 0, None, 64),
    (2, 4, 2, 130, 130, True, 0, "right", 64),
    (2, 3, 1, 70, 70, False, 0, "right", 64),
    (2, 4, 2, 100, 100, True, 0, "left", 64),
    (1, 2, 2, 50, 50, False, 0, "left", 72),
    (2, 4, 2, 90, 90, True, 0, "right", 128),
    (1, 2, 1, 65, 65, True, 0, None, 256),
    (2, 12, 4, 40, 100, True, 60, "right", 64),
    (2, 2, 2, 33, 77, False, 0, None, 40),
]


@pytest.mark.parametrize("case", BWD_F32_CASES)
def test_attention_bwd_fp32(case):
    """fp32 attention backward against fp64 autograd through the reference formulation (scores + additive finfo.min
    mask, softmax, @ v: layers/attention.py:127-140) at 1e-5 -- including rows without a visible key (left padding under
    a causal mask; every key padded), whose uniform softmax autograd differentiates like any other."""
    ops = _ops()
    B, h, hk, L, S, causal, start, kpk, dh = case
    q, k, v = rnd(B, h, L, dh, seed=1), rnd(B, hk, S, dh, seed=2), rnd(B, hk, S, dh, seed=3)
    do = rnd(B, L, h * dh, seed=4)
    keypad = None
    if kpk is not None:
        keypad = torch.ones(B, S, dtype=torch.uint8)
        if kpk == "right":
            keypad[0, S - S // 3:] = 0
            if B > 1:
                keypad[1, S - 7:] = 0
        else:
            keypad[0, : S // 4] = 0          # causal: the first rows of sequence 0 see nothing
            if not causal:
                keypad[0, :] = 0             # non-causal: every row of sequence 0 is dead
    mask = _dense_mask(B, L, S, causal, start, keypad, None)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    s = qd @ O.repeat_kv(kd, h // hk).transpose(-1, -2) / math.sqrt(dh) + mask.double()
    out_ref = O.merge_heads(torch.softmax(s, -1) @ O.repeat_kv(vd, h // hk))
    (out_ref * do.double()).sum().backward()
    qg, kg, vg = q.to(DEV), k.to(DEV), v.to(DEV)
    kpg = keypad.to(DEV) if keypad is not None else None
    lse = torch.zeros(B, h, L, dtype=torch.float32, device=DEV)
    out = ops.attention(qg, kg, vg, causal=causal, start_pos=start, keypad=kpg, lse=lse)
    W = (h + 2 * hk) * dh
    packed = torch.zeros(B, max(L, S), W, dtype=torch.float32, device=DEV)
    dq = packed[:, :L, : h * dh].view(B, L, h, dh).permute(0, 2, 1, 3)
    dk = packed[:, :S, h * dh:(h + hk) * dh].view(B, S, hk, dh).permute(0, 2, 1, 3)
    dv = packed[:, :S, (h + hk) * dh:].view(B, S, hk, dh).permute(0, 2, 1, 3)
    ops.attention_bwd(qg, kg, vg, out, do.to(DEV), lse, dq, dk, dv, causal=causal, start_pos=start, keypad=kpg)
    check(out, out_ref, 2e-5, 1e-5, "fwd out fp32")
    check(dq, qd.grad, 3e-5, 1e-4, "dq fp32")
    check(dk, kd.grad, 3e-5, 1e-4, "dk fp32")
    check(dv, vd.grad, 3e-5, 1e-4, "dv fp32")


def test_attention_bwd_fp32_rope_inverse():
    ops = _ops()
    B, h, hk, L, dh = 2, 4, 2, 48, 64
    q, k, v = rnd(B, h, L, dh, seed=1).to(DEV), rnd(B, hk, L, dh, seed=2).to(DEV), rnd(B, hk, L, dh, seed=3).to(DEV)
    do = rnd(B, L, h * dh, seed=4).to(DEV)
    cos, sin = ops.rope_tables(dh, 64, DEV)
    lse = torch.zeros(B, h, L, dtype=torch.float32, device=DEV)
    out = ops.attention(q, k, v, causal=True, lse=lse)
    res = []
    for fused in (True, False):
        dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        if fused:
            ops.attention_bwd(q, k, v, out, do, lse, dq, dk, dv, causal=True, cos=cos, sin=sin, rope_pos0=3)
        else:
            ops.attention_bwd(q, k, v, out, do, lse, dq, dk, dv, causal=True)
            ops.rope_(dq, cos, sin, 3, inverse=True)
            ops.rope_(dk, cos, sin, 3, inverse=True)
        res.append((dq, dk, dv))
    for a, b_ in zip(*res):
        assert torch.equal(a, b_)
