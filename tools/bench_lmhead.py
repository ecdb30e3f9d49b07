"""The vocabulary projection of the training step (M = 16384, N = 50265, K = 768, bias, row stride 50272):
persistent pipelined kernel (VY_GEMM_PIPE_WIDE=1, default) against the one-shot 256 x 256 tiles (=0), equality and time."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops
M, N, K, ld = 16384, 50265, 768, 50272
bf, dev = torch.bfloat16, "cuda"
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).to(bf).to(dev)
w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(bf).to(dev)
b = torch.randn(N, generator=g).to(bf).to(dev)
buf = torch.zeros(M, ld, dtype=bf, device=dev)
out = buf[:, :N]
ops.linear(x, w, b, out=out)
torch.cuda.synchronize()
ref = (x[:64].float() @ w.float().T + b.float())
print("max |err| vs fp32 on 64 rows:", (out[:64].float() - ref).abs().max().item(), " pad columns zero:", float(buf[:, N:].abs().max()) == 0.0)
print("checksum", out.float().sum().item(), out[:, -200:].float().abs().sum().item())
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    ops.linear(x, w, b, out=out)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 100
print(f"{us:.0f} us per launch = {2.0 * M * N * K / us * 1e-6:.0f} TFLOP/s")
