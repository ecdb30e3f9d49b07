"""Event-timed mid-size-M GEMM shapes (PaliGemma-shape prefill, captioning decoder): python tools/bench_mid_gemm.py"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops
shapes = [(264, 32768, 2048), (264, 2048, 16384), (264, 2560, 2048), (264, 2048, 2048), (256, 4304, 1152), (256, 1152, 4304),
          (256, 1152, 1152), (256, 3456, 1152), (2112, 768, 768), (2112, 3072, 768), (2112, 768, 3072)]
for M, N, K in shapes:
    x = (torch.randn(M, K, device="cuda")).bfloat16()
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.linear(x, w, out=y)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.linear(x, w, out=y)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e3
    print(f"{M:5d} x {N:6d} x {K:6d}: {t:7.1f} us  {2.0 * M * N * K / t * 1e-6:7.1f} TFLOP/s  weights {N * K * 2 / t * 1e-6:5.2f} TB/s")
