"""Fixed cost of a GEMM launch: time vs K at the training tile grid (M=16384)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops
from tools.bench_kernels import timeit
M = 16384
bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(bf).cuda()
for N in (768, 2304):
    for K in (64, 128, 256, 768, 1536, 3072):
        x, w, b = r(M, K), (r(N, K) / K ** 0.5).contiguous(), r(N)
        res = r(M, N)
        out = torch.empty(M, N, dtype=bf, device="cuda")
        t0 = timeit(lambda: ops.linear(x, w, b, out=out), 20)
        t1 = timeit(lambda: ops.linear(x, w, b, residual=res, out=out), 20)
        print(f"N={N:5d} K={K:5d}  plain {t0:7.1f} us   +res {t1:7.1f} us   ({2.0*M*N*K/t0*1e-6:7.1f} TF/s)")
