"""Persistent pipelined GEMM (VY_GEMM_PIPE_WIDE=1 forces it for wide outputs) against the one-shot kernels as N grows
(M = 16384, K = 768, bias): is its per-tile time constant?"""
import os, sys, math, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops, _lib
lib = _lib.load(); lib.vy_debug_set_gemm_variant.argtypes = [C.c_int]
M, K = 16384, 768
bf, dev = torch.bfloat16, "cuda"
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).to(bf).to(dev)
for N in (2304, 4608, 9216, 18432, 36864, 49920):
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(bf).to(dev)
    b = torch.randn(N, generator=g).to(bf).to(dev)
    out = torch.empty(M, N, dtype=bf, device=dev)
    res = []
    for var in (30, 31):
        lib.vy_debug_set_gemm_variant(var)
        ops.linear(x, w, b, out=out); ops.linear(x, w, b, out=out); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            ops.linear(x, w, b, out=out)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 100
        res.append(us)
    tiles = (M // 256) * (N // 192) / 256
    print(f"N {N:6d}: persistent {res[0]:8.1f} us ({res[0] / tiles:5.1f} us per tile and CU, {2.0*M*N*K/res[0]*1e-6:4.0f} TF)   one-shot {res[1]:8.1f} us ({2.0*M*N*K/res[1]*1e-6:4.0f} TF)")
    del w, b, out
