"""LayerNorm forward / backward at the training shape (16384 x 768 bf16): time and HBM rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops
from tools.bench_kernels import timeit
M, N = 16384, 768
bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
x = torch.randn(M, N, generator=g).to(bf).cuda(); dy = torch.randn(M, N, generator=g).to(bf).cuda()
gm = torch.ones(N, dtype=bf, device="cuda"); bt = torch.zeros(N, dtype=bf, device="cuda")
y, mean, rstd = ops.layernorm(x, gm, bt, 1e-5, save_stats=True)
dg = torch.zeros(N, dtype=torch.float32, device="cuda"); db = torch.zeros(N, dtype=torch.float32, device="cuda")
tf = timeit(lambda: ops.layernorm(x, gm, bt, 1e-5, save_stats=True), 50)
tb = timeit(lambda: ops.layernorm_bwd(dy, x, gm, mean, rstd, dg, db, accumulate=True), 50)
print(f"layernorm fwd {tf:.1f} us ({2 * M * N * 2 / tf * 1e-6:.2f} TB/s)   bwd (both launches) {tb:.1f} us ({3 * M * N * 2 / tb * 1e-6:.2f} TB/s)")
