"""The three dgrad launches of a layer + the FFN2 dgrad with the GELU derivative (B=32 x 512 rows, d=768), per
GEMM variant:  python tools/bench_dgrad.py [variants...]   (VY_GEMM_VARIANT values; -1 = default selection)"""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops, _lib
lib = _lib.load()
lib.vy_debug_set_gemm_variant.argtypes = [C.c_int]
M, d = 16384, 768
bf, dev = torch.bfloat16, "cuda"
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(bf).to(dev)
dy_d, dy_3d, dy_4d = r(M, d), r(M, 3 * d), r(M, 4 * d)
wo_t, wqkv_t, w1_t, w2_t = r(d, d) / 28, r(d, 3 * d) / 48, r(d, 4 * d) / 55, r(4 * d, d) / 28   # W^T layouts: [N_out = in_features][K = out_features]
pre = r(M, 4 * d)
res = r(M, d)
outs = {k: torch.empty(M, n, dtype=bf, device=dev) for k, n in (("d", d), ("4d", 4 * d))}


def t(fn, it=20):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


cases = {
    "out dgrad  (N 768, K 768)": (lambda: ops.linear_dgrad(dy_d, wo_t, out=outs["d"]), 2.0 * M * d * d),
    "qkv dgrad  (N 768, K 2304, + residual)": (lambda: ops.linear_dgrad(dy_3d, wqkv_t, add_to=res, out=outs["d"]), 2.0 * M * d * 3 * d),
    "ffn1 dgrad (N 768, K 3072, + residual)": (lambda: ops.linear_dgrad(dy_4d, w1_t, add_to=res, out=outs["d"]), 2.0 * M * d * 4 * d),
    "ffn2 dgrad (N 3072, K 768, * gelu'(pre))": (lambda: ops.linear_dgrad(dy_d, w2_t, pre=pre, act=1, out=outs["4d"]), 2.0 * M * d * 4 * d),
}
for v in [int(x) for x in sys.argv[1:]] or [-1]:
    lib.vy_debug_set_gemm_variant(v)
    line = [f"variant {v:3d}:"]
    for name, (fn, fl) in cases.items():
        us = t(fn)
        line.append(f"{name} {us:6.1f} us {fl / us * 1e-6:5.0f} TF")
    print("  |  ".join(line))
