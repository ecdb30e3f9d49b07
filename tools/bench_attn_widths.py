"""Attention forward / backward time per head width at the benchmark's token count (B*h*dh = 32 * 768 columns, L = 512,
causal): the tuned dh = 64 kernels beside the general ones (72 / 128 / 256)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops
from tools.bench_kernels import timeit
bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(bf).cuda()
B, L = 32, 512
for dh, h in ((64, 12), (72, 16), (96, 8), (128, 6), (256, 3)):
    for causal in (True, False):
        q, k, v = r(B, h, L, dh), r(B, h, L, dh), r(B, h, L, dh)
        do = r(B, L, h * dh)
        o = torch.empty(B, L, h * dh, dtype=bf, device="cuda")
        lse = torch.empty(B, h, L, dtype=torch.float32, device="cuda")
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        tf = timeit(lambda: ops.attention(q, k, v, causal=causal, out=o, lse=lse), 20)
        tb = timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, causal=causal), 20)
        fl = 4.0 * B * h * L * L * dh * (0.5 if causal else 1.0)
        print(f"dh={dh:3d} h={h:2d} causal={int(causal)}  fwd {tf:7.1f} us {fl / tf * 1e-6:6.1f} TFLOP/s   "
              f"bwd {tb:7.1f} us {2.5 * fl / tb * 1e-6:6.1f} TFLOP/s")
