"""Shader clock the GEMM kernels really run at, (a) in a warm micro-benchmark loop and (b) inside the
training step:  VY_GEMM_ROT=16 python tools/gemm_clock.py
(workgroup 0 of every GEMM launch accumulates s_memtime cycles and s_memrealtime ticks)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vyomai_amd import ops, _lib
from tools.bench_kernels import timeit

def read():
    out = (C.c_ulonglong * 6)()
    torch.cuda.synchronize()
    assert _lib.load().vy_debug_gemm_clock(out) == 0
    cyc, ticks, n = out[0], out[1], out[2]
    return (cyc / (ticks / 100.0) if ticks else 0.0), n, (ticks * 0.01 / n if n else 0.0)

M, bf = 16384, torch.bfloat16
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(bf).cuda()
read()
for N, K in ((768, 768), (768, 3072), (2304, 768), (3072, 768)):
    x, w, b = r(M, K), (r(N, K) / K ** 0.5).contiguous(), r(N)
    out = torch.empty(M, N, dtype=bf, device="cuda")
    for iters in (20, 400):
        t = timeit(lambda: ops.linear(x, w, b, out=out), iters)
        mhz, n, us = read()
        print(f"micro N={N} K={K} x{iters}: {t:7.1f} us/launch, workgroup 0 alive {us:6.1f} us, shader clock {mhz:6.0f} MHz ({n} launches)")

sys.argv = ["bench.py", "--no-decode", "--no-cpu-baseline", "--steps", "6", "--warmup", "2"]
read()
bench.main()
mhz, n, us = read()
print(f"training step + probe: shader clock {mhz:6.0f} MHz over {n} GEMM launches (workgroup 0 alive {us:.1f} us on average)")
