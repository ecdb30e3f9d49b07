"""wgrad of a stacked problem with as many 256 x 256 tiles as one / two layers have together (N = 9216 / 18432):
VY_WGRAD_VARIANT=8 VY_WGRAD_TARGET=216 python tools/bench_wgrad_stacked.py  vs  VY_WGRAD_VARIANT=0 -- the estimate
behind vy_linear_wgrad_grouped."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops
from tools.bench_kernels import timeit
M = 16384
bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(bf).cuda()
for N, K in ((9216, 768), (18432, 768)):
    dy, x = r(M, N), r(M, K)
    dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    t = timeit(lambda: ops.linear_wgrad(dy, x, dw, db, accumulate=True), 10)
    print(f"wgrad N={N} K={K} variant {os.environ.get('VY_WGRAD_VARIANT')} target {os.environ.get('VY_WGRAD_TARGET')}: {t:8.1f} us  {2.0*M*N*K/t*1e-6:8.1f} TFLOP/s")
