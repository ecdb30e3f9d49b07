"""Same-process A/B of the large-M bf16 GEMM variants on the four forward launches of one layer
(B=32 x 512 rows, d=768): correctness against the default selection, then interleaved timing rounds.
  python tools/exp_gemm.py [variants...]      e.g.  python tools/exp_gemm.py -1 8 9
Also times the same work as two half-batches on two streams (chip-level de-phasing)."""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops, _lib  # noqa: E402

lib = _lib.load()
lib.vy_debug_set_gemm_variant.argtypes = [C.c_int]


def setvar(v):
    lib.vy_debug_set_gemm_variant(int(v))


def timeit(fn, iters=20, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    variants = [int(v) for v in sys.argv[1:]] or [-1, 8, 9]
    B, L, d, h, dh = 32, 512, 768, 12, 64
    M = B * L
    dev, bf = "cuda", torch.bfloat16
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(bf).to(dev)
    x = r(M, d)
    x3 = x.view(B, L, d)
    wqkv, bqkv = (r(3 * d, d) / math.sqrt(d)).contiguous(), r(3 * d)
    wo, bo = (r(d, d) / math.sqrt(d)).contiguous(), r(d)
    w1, b1 = (r(4 * d, d) / math.sqrt(d)).contiguous(), r(4 * d)
    w2, b2 = (r(d, 4 * d) / math.sqrt(4 * d)).contiguous(), r(d)
    cos, sin = ops.rope_tables(dh, 1024, dev)
    hm_in = r(M, 4 * d)

    def mk():
        q = torch.empty(B, h, L, dh, dtype=bf, device=dev)
        return dict(q=q, k=torch.empty_like(q), v=torch.empty_like(q), s1=torch.empty(M, d, dtype=bf, device=dev),
                    hm=torch.empty(M, 4 * d, dtype=bf, device=dev), s2=torch.empty(M, d, dtype=bf, device=dev),
                    hm2=torch.empty(M, 4 * d, dtype=bf, device=dev), pre=torch.empty(M, 4 * d, dtype=bf, device=dev),
                    qkvp=torch.empty(M, 3 * d, dtype=bf, device=dev))

    launches = {
        "qkv+rope": lambda o: ops.qkv_rope(x3, wqkv, bqkv, h, h, dh, cos, sin, 0, o["q"], o["k"], o["v"]),
        "out+res": lambda o: ops.linear(x, wo, bo, residual=x, out=o["s1"]),
        "ffn1+gelu": lambda o: ops.linear(x, w1, b1, act=1, out=o["hm"]),
        "ffn1+gelu+pre": lambda o: ops.linear(x, w1, b1, act=1, out=o["hm2"], pre_out=o["pre"]),
        "qkv plain": lambda o: ops.linear(x, wqkv, bqkv, out=o["qkvp"]),
        "ffn2+res": lambda o: ops.linear(hm_in, w2, b2, residual=x, out=o["s2"]),
    }
    flops = {"qkv+rope": 2.0 * M * 3 * d * d, "out+res": 2.0 * M * d * d, "ffn1+gelu": 2.0 * M * 4 * d * d,
             "ffn2+res": 2.0 * M * 4 * d * d, "ffn1+gelu+pre": 2.0 * M * 4 * d * d, "qkv plain": 2.0 * M * 3 * d * d}
    # correctness: every variant against the default selection, bit for bit (same arithmetic order per
    # output element: one accumulation chain over k in slices, fp32)
    setvar(-1)
    ref = mk()
    for f in launches.values():
        f(ref)
    torch.cuda.synchronize()
    for v in variants:
        if v == -1:
            continue
        setvar(v)
        o = mk()
        for f in launches.values():
            f(o)
        torch.cuda.synchronize()
        for kname in ref:
            a, b = ref[kname].float(), o[kname].float()
            md = (a - b).abs().max().item()
            print(f"variant {v:3d} {kname:3s}: max |diff| vs default {md:.3e}  equal={torch.equal(ref[kname], o[kname])}")
    # timing: interleaved rounds
    outs = mk()
    res = {v: {n: [] for n in launches} for v in variants}
    allt = {v: [] for v in variants}
    for rnd in range(4):
        for v in variants:
            setvar(v)
            for n, f in launches.items():
                res[v][n].append(timeit(lambda: f(outs), 20))
            allt[v].append(timeit(lambda: [f(outs) for f in launches.values()], 10))
    for v in variants:
        line = [f"variant {v:3d}:"]
        for n in launches:
            t = sorted(res[v][n])[len(res[v][n]) // 2]
            line.append(f"{n} {t:6.1f}us {flops[n] / t * 1e-6:6.0f}TF")
        t = sorted(allt[v])[len(allt[v]) // 2]
        line.append(f"| 4 launches {t:6.1f}us = {sum(flops.values()) / t * 1e-6:6.0f} TF ({sum(flops.values()) / t * 1e-6 / 25:.1f}%)")
        print("  ".join(line))
    # two half-batches on two streams, default selection per half
    for v in variants:
        setvar(v)
        Mh = M // 2
        halves = []
        for hb in range(2):
            xs = x[hb * Mh:(hb + 1) * Mh]
            q = torch.empty(B // 2, h, L, dh, dtype=bf, device=dev)
            halves.append(dict(x=xs, x3=xs.view(B // 2, L, d), q=q, k=torch.empty_like(q), v=torch.empty_like(q),
                               s1=torch.empty(Mh, d, dtype=bf, device=dev), hm=torch.empty(Mh, 4 * d, dtype=bf, device=dev),
                               hin=hm_in[hb * Mh:(hb + 1) * Mh], s2=torch.empty(Mh, d, dtype=bf, device=dev)))
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]

        def two():
            cur = torch.cuda.current_stream()
            for s_, hh in zip(streams, halves):
                s_.wait_stream(cur)
                with torch.cuda.stream(s_):
                    ops.qkv_rope(hh["x3"], wqkv, bqkv, h, h, dh, cos, sin, 0, hh["q"], hh["k"], hh["v"])
                    ops.linear(hh["x"], wo, bo, residual=hh["x"], out=hh["s1"])
                    ops.linear(hh["x"], w1, b1, act=1, out=hh["hm"])
                    ops.linear(hh["hin"], w2, b2, residual=hh["x"], out=hh["s2"])
            for s_ in streams:
                cur.wait_stream(s_)
        t = timeit(two, 10)
        print(f"variant {v:3d}: two half-batch streams, 8 launches {t:6.1f}us = {sum(flops.values()) / t * 1e-6:6.0f} TF")


if __name__ == "__main__":
    main()
