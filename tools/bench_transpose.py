"""Batched W^T refresh: one vy_transpose_batched launch over the decoder's 2-D weights."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops
from tools.bench_kernels import timeit
bf = torch.bfloat16
shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)] * 12 + [(768, 768), (50265, 768)]
pairs = []
for R, C in shapes:
    src = torch.randn(R, C, dtype=torch.float32).to(bf).cuda()
    ld = (R + 7) // 8 * 8
    pairs.append((src, torch.zeros(C, ld, dtype=bf, device="cuda")[:, :R]))
tb = ops.TransposeBatch(pairs)
t = timeit(tb.run, 20)
nbytes = sum(2 * R * C * 2 for R, C in shapes)
print(f"batched transpose of {len(pairs)} matrices: {t:.1f} us, {nbytes / t * 1e-6:.2f} TB/s (read + write)")
