"""Cycles of workgroup 0 in prologue / k-loop / epilogue (VY_GEMM_ROT=16) for the four forward launches of a layer
as bench.py's roofline probe issues them (QKV + RoPE + head split, out + residual, FFN1 + GELU, FFN2 + residual)."""
import os, sys, math, torch
os.environ.setdefault("VY_GEMM_ROT", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops, _lib
from tools.bench_kernels import timeit
import ctypes as C


def clock():
    out = (C.c_ulonglong * 6)()
    torch.cuda.synchronize()
    _lib.load().vy_debug_gemm_clock(out)
    n = max(out[2], 1)
    return (out[0] / (out[1] / 100.0) if out[1] else 0.0), (out[0] / n), [out[3] / n, out[4] / n, out[5] / n]


B, L, d, h, dh = 32, 512, 768, 12, 64
M = B * L
bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(bf).cuda()
x = r(M, d); x3 = x.view(B, L, d)
wqkv, bqkv = (r(3 * d, d) / math.sqrt(d)).contiguous(), r(3 * d)
wo, bo = (r(d, d) / math.sqrt(d)).contiguous(), r(d)
w1, b1 = (r(4 * d, d) / math.sqrt(d)).contiguous(), r(4 * d)
w2, b2 = (r(d, 4 * d) / math.sqrt(4 * d)).contiguous(), r(d)
cos, sin = ops.rope_tables(dh, 1024, "cuda")
q = torch.empty(B, h, L, dh, dtype=bf, device="cuda"); k = torch.empty_like(q); v = torch.empty_like(q)
s1 = torch.empty(M, d, dtype=bf, device="cuda"); hm = torch.empty(M, 4 * d, dtype=bf, device="cuda"); hin = r(M, 4 * d)
pre = torch.empty(M, 4 * d, dtype=bf, device="cuda")
cases = {
    "qkv+rope": lambda: ops.qkv_rope(x3, wqkv, bqkv, h, h, dh, cos, sin, 0, q, k, v),
    "qkv no rope": lambda: ops.qkv_rope(x3, wqkv, bqkv, h, h, dh, None, None, 0, q, k, v),
    "qkv plain linear": lambda: ops.linear(x, wqkv, bqkv, out=torch.empty(M, 3 * d, dtype=bf, device="cuda")),
    "out+res": lambda: ops.linear(x, wo, bo, residual=x, out=s1),
    "ffn1+gelu": lambda: ops.linear(x, w1, b1, act=1, out=hm),
    "ffn1+gelu+pre (training)": lambda: ops.linear(x, w1, b1, act=1, out=hm, pre_out=pre),
    "ffn1 no act": lambda: ops.linear(x, w1, b1, out=hm),
    "ffn2+res": lambda: ops.linear(hin, w2, b2, residual=x, out=s1),
}
for name, f in cases.items():
    clock()
    t = timeit(f, 20)
    mhz, cyc, parts = clock()
    print(f"{name:28s} {t:7.1f} us   wg0: {cyc/1e3:6.1f} kcyc at {mhz:4.0f} MHz  (prologue {parts[0]/1e3:5.1f}, k-loop {parts[1]/1e3:6.1f}, epilogue {parts[2]/1e3:5.1f})")
