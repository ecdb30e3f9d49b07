"""Attention forward time vs sequence length (fixed B*h*L): separates per-workgroup fixed cost from per-tile cost."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops
from tools.bench_kernels import timeit
bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(bf).cuda()
h, dh = 12, 64
for L in (128, 256, 512, 1024, 2048, 4096):
    B = max(1, 16384 // L)
    q, k, v = r(B, h, L, dh), r(B, h, L, dh), r(B, h, L, dh)
    o = torch.empty(B, L, h * dh, dtype=bf, device="cuda")
    for causal in (False, True):
        t = timeit(lambda: ops.attention(q, k, v, causal=causal, out=o), 20)
        wgs = B * h * ((L + 127) // 128)
        tiles = (L // 64) if not causal else None
        fl = 4.0 * B * h * L * L * dh * (0.5 if causal else 1.0)
        print(f"L={L:5d} B={B:4d} causal={int(causal)}  {t:8.1f} us  WGs={wgs:6d}  {fl/t*1e-6:7.1f} TFLOP/s")
