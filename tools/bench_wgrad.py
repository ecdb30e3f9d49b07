"""wgrad / layernorm-bwd micro-benchmark at config-2 shapes."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops
from tools.bench_kernels import timeit
M = 16384
bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(bf).cuda()
for name, N, K in (("qkv", 2304, 768), ("out", 768, 768), ("ffn1", 3072, 768), ("ffn2", 768, 3072)):
    dy, x = r(M, N), r(M, K)
    dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    t = timeit(lambda: ops.linear_wgrad(dy, x, dw, db, accumulate=True), 10)
    print(f"wgrad {name:5s} {t:8.1f} us  {2.0*M*N*K/t*1e-6:8.1f} TFLOP/s")
# LM head: dY = dlogits with a padded row stride
V, ld = 50265, 50272
buf = torch.empty(M, ld, dtype=bf, device="cuda").normal_()
buf[:, V:] = 0
xl = r(M, 768)
dw = torch.zeros(V, 768, device="cuda"); db = torch.zeros(V, device="cuda")
t = timeit(lambda: ops.linear_wgrad(buf[:, :V], xl, dw, db, accumulate=True), 5)
print(f"wgrad lmhead {t:8.1f} us  {2.0*M*V*768/t*1e-6:8.1f} TFLOP/s")
del buf, dw
x = r(M, 768); dy = r(M, 768); gam = r(768); bet = r(768)
y, mean, rstd = ops.layernorm(x, gam, bet, 1e-5, save_stats=True)
dg = torch.zeros(768, device="cuda"); dbt = torch.zeros(768, device="cuda")
t = timeit(lambda: ops.layernorm_bwd(dy, x, gam, mean, rstd, dg, dbt, accumulate=True), 10)
print(f"layernorm_bwd {t:8.1f} us  {4*M*768*2/t*1e-3:8.1f} GB/s")
