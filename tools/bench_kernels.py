"""Micro-benchmark of the block's kernels at BASELINE config-2 shapes (B=32, L=512, d=768, h=12).
Run on the GPU box: python tools/bench_kernels.py [--iters N]"""
import argparse
import math
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops, _lib  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--L", type=int, default=512)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    B, L, d, h, dh = a.B, a.L, 768, 12, 64
    M = B * L
    dev, bf = "cuda", torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *s: (torch.randn(*s, generator=g)).to(bf).to(dev)
    x = r(M, d)
    rows = []
    for name, N, K, act, res in (("qkv(plain)", 2304, 768, 0, False), ("out+res", 768, 768, 0, True),
                                 ("ffn1+gelu", 3072, 768, 1, False), ("ffn2+res", 768, 3072, 0, True),
                                 ("lm_dense", 768, 768, 1, False)):
        if a.only and a.only not in name:
            continue
        xx = r(M, K)
        w = (r(N, K) / math.sqrt(K)).contiguous()
        b = r(N)
        rs = r(M, N) if res else None
        out = torch.empty(M, N, dtype=bf, device=dev)
        t = timeit(lambda: ops.linear(xx, w, b, act=act, residual=rs, out=out), a.iters)
        fl = 2.0 * M * N * K
        rows.append((name, t, fl / t * 1e-6))
    if a.only:
        for n, t, f in rows:
            print(f"{n:40s} {t:10.1f} us   {f:10.1f} TFLOP/s|GB/s")
        return
    # lm head vocab projection
    w = (r(50265, 768) / 27.7).contiguous()
    b = r(50265)
    out = torch.empty(M, 50272, dtype=bf, device=dev)[:, :50265]
    t = timeit(lambda: ops.linear(x, w, b, out=out), max(3, a.iters // 4))
    rows.append(("lm_vocab", t, 2.0 * M * 50265 * 768 / t * 1e-6))
    # fused qkv + rope
    w = (r(2304, 768) / 27.7).contiguous()
    b = r(2304)
    cos, sin = ops.rope_tables(dh, 1024, dev)
    q = torch.empty(B, h, L, dh, dtype=bf, device=dev)
    k = torch.empty_like(q)
    v = torch.empty_like(q)
    x3 = x.view(B, L, d)
    t = timeit(lambda: ops.qkv_rope(x3, w, b, h, h, dh, cos, sin, 0, q, k, v), a.iters)
    rows.append(("qkv+rope fused", t, 2.0 * M * 2304 * 768 / t * 1e-6))
    # attention
    q, k, v = r(B, h, L, dh), r(B, h, L, dh), r(B, h, L, dh)
    o = torch.empty(B, L, d, dtype=bf, device=dev)
    t = timeit(lambda: ops.attention(q, k, v, causal=True, out=o), a.iters)
    rows.append(("attn causal (causal-counted flops)", t, 2.0 * B * h * L * L * dh / t * 1e-6))
    t = timeit(lambda: ops.attention(q, k, v, causal=False, out=o), a.iters)
    rows.append(("attn full", t, 4.0 * B * h * L * L * dh / t * 1e-6))
    # attention backward (delta + dq + dkdv), causal
    o4 = torch.empty(B, L, h, dh, dtype=bf, device=dev)
    lse = torch.empty(B, h, L, dtype=torch.float32, device=dev)
    ops.attention(q, k, v, causal=True, out=o4.view(B, L, d), lse=lse)
    do = r(B, L, h, dh)
    dq_, dk_, dv_ = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    t = timeit(lambda: ops.attention_bwd(q, k, v, o4.view(B, L, d), do.view(B, L, d), lse, dq_, dk_, dv_, causal=True), a.iters)
    rows.append(("attn bwd causal (causal-counted flops)", t, 5.0 * B * h * L * L * dh / t * 1e-6))
    # layernorm
    gma, bta = r(d), r(d)
    t = timeit(lambda: ops.layernorm(x, gma, bta, 1e-5), a.iters)
    rows.append(("layernorm (GB/s in last col)", t, 2 * M * d * 2 / t * 1e-3))
    # decode attention B=32 S=640
    qd = r(B, h, 1, dh)
    kc, vc = r(B, h, 640, dh), r(B, h, 640, dh)
    t = timeit(lambda: ops.attention_decode(qd, kc, vc, 640), a.iters)
    rows.append(("decode attn S=640 (GB/s)", t, 2 * B * h * 640 * dh * 2 / t * 1e-3))
    for n, t, f in rows:
        print(f"{n:40s} {t:10.1f} us   {f:10.1f} TFLOP/s|GB/s")
    blk = sum(t for n, t, _ in rows if n in ("qkv+rope fused", "out+res", "ffn1+gelu", "ffn2+res",
                                            "attn causal (causal-counted flops)")) + 2 * [t for n, t, _ in rows if n.startswith("layernorm")][0]
    fl = (24 * d * d + 2 * L * d) * M
    print(f"block forward (sum of kernels): {blk:.1f} us -> {fl / blk * 1e-6:.1f} TFLOP/s = {fl / blk * 1e-6 / 2500 * 100:.1f}% of 2.5 PF")


if __name__ == "__main__":
    main()
