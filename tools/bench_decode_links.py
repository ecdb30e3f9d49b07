"""What does each kernel of the decode step cost as a link of a hipGraph chain, on its own?  Chains of ONE op
repeated (same stream, so every launch waits for the previous one), with the weights either the same every time
(hot) or rotating through a pool larger than the caches (cold, as in a real step), B=32 rows, d=768:
  python tools/bench_decode_links.py
Compare with tools/probe/cold_chain_probe (a bare streaming kernel: 1.8 us hot, 2.7 us cold at 24 KiB per
workgroup) to see how much of a link is the kernel's own instruction path."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops  # noqa: E402

B, d, h, dh, ffn = 32, 768, 12, 64, 3072
dev, bf = "cuda", torch.bfloat16
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(bf).to(dev)
CHAIN, POOL = 120, 60


def graph_time(fn_i, chain=CHAIN):
    """fn_i(i) enqueues link i.  Returns us per link under graph replay."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(4):
            fn_i(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(chain):
            fn_i(i)
    gr.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        gr.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (5 * chain)


def main():
    x = r(B, d)
    x3 = x.view(B, 1, d)
    mid = r(B, ffn)
    cos, sin = ops.rope_tables(dh, 1024, dev)
    wq = [(r(3 * d, d) / math.sqrt(d)).contiguous() for _ in range(POOL)]
    w1 = [(r(ffn, d) / math.sqrt(d)).contiguous() for _ in range(POOL)]
    wo = [(r(d, d) / math.sqrt(d)).contiguous() for _ in range(POOL)]
    w2 = [(r(d, ffn) / math.sqrt(ffn)).contiguous() for _ in range(POOL // 2)]
    bq, b1, bo = r(3 * d), r(ffn), r(d)
    gam, bet = r(d), r(d)
    S = 577
    cap = 640
    kc = [r(B, h, cap, dh) for _ in range(12)]
    vc = [r(B, h, cap, dh) for _ in range(12)]
    q = torch.empty(B, h, 1, dh, dtype=bf, device=dev)
    ao = torch.empty(B, d, dtype=bf, device=dev)
    y1 = torch.empty(B, ffn, dtype=bf, device=dev)
    yd = torch.empty(B, d, dtype=bf, device=dev)
    from vyomai_amd import _lib
    import ctypes as C
    lib = _lib.load()

    def qkv(i, hot):
        w = wq[0 if hot else i % POOL]
        kk = kc[i % 12][:, :, 576:577]
        vv = vc[i % 12][:, :, 576:577]
        ops.qkv_rope(x3, w, bq, h, h, dh, cos, sin, 576, q, kk, vv)

    def ffn1(i, hot):
        ops.linear(x, w1[0 if hot else i % POOL], b1, act=1, out=y1)

    def ffn1_noact(i, hot):
        ops.linear(x, w1[0 if hot else i % POOL], b1, act=0, out=y1)

    def attn(i, hot):
        j = 0 if hot else i % 12
        ops.attention_decode(q, kc[j], vc[j], S)

    def ln(i, hot):
        ops.layernorm(x, gam, bet, 1e-5)

    rows = [("qkv+rope+cache (skinny16)", qkv), ("ffn1+gelu (skinny16)", ffn1), ("ffn1 no act (skinny16)", ffn1_noact),
            ("layernorm 32 rows", ln)]
    try:
        ops.attention_decode(q, kc[0], vc[0], S)
        rows.append(("attention decode S=577", attn))
    except Exception as ex:  # noqa: BLE001
        print("attention_decode signature differs:", ex)
    for name, fn in rows:
        th = graph_time(lambda i: fn(i, True))
        tc = graph_time(lambda i: fn(i, False))
        print(f"{name:32s} hot {th:6.2f} us   cold {tc:6.2f} us per link")


if __name__ == "__main__":
    main()
