"""GEMM launch time vs operand VALUES (DATA=randn|zeros|ones|small: the matrix pipe draws less power on
quiet data) and, with VY_GEMM_ROT=16, the shader clock and the cycles workgroup 0 was alive.
The k-loop's ingredients are taken apart in tools/probe/gemm_loop_probe.hip, not here."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops
from tools.bench_kernels import timeit
import ctypes as C
from vyomai_amd import _lib
def clock():
    out = (C.c_ulonglong * 6)()
    torch.cuda.synchronize()
    _lib.load().vy_debug_gemm_clock(out)
    n = max(out[2], 1)
    return (out[0] / (out[1] / 100.0) if out[1] else 0.0), (out[0] / n), [out[3] / n, out[4] / n, out[5] / n]
M = 16384
bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(bf).cuda()
line = [f"mask={os.environ.get('VY_GEMM_ROT', '0'):>2s} data={os.environ.get('DATA', 'randn'):6s}"]
for N, K in ((768, 768), (768, 3072), (2304, 768), (3072, 768)):
    x, w, b = r(M, K), (r(N, K) / K ** 0.5).contiguous(), r(N)
    mode = os.environ.get("DATA", "randn")      # operand values: the matrix pipe is throttled by toggling
    if mode == "zeros":
        x.zero_(); w.zero_()
    elif mode == "ones":
        x.fill_(1.0); w.fill_(1.0 / K)
    elif mode == "small":                       # few distinct values (sign only)
        x = x.sign(); w = w.sign() / K
    out = torch.empty(M, N, dtype=bf, device="cuda")
    clock()
    res = r(M, N) if os.environ.get("RES") else None
    t0 = timeit(lambda: ops.linear(x, w, b, residual=res, out=out), 20)
    mhz, cyc, parts = clock()
    line.append(f"N{N}K{K}: {t0:6.1f}us" + (f" {mhz:4.0f}MHz {cyc/1e3:5.1f}kcyc (pro {parts[0]/1e3:.1f} loop {parts[1]/1e3:.1f} epi {parts[2]/1e3:.1f})" if mhz else ""))
print("  ".join(line))
