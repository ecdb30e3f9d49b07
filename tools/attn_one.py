"""One attention-forward configuration, a few launches (for rocprofv3 --pmc passes): python tools/attn_one.py L B causal"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops
L, B, causal = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
h, dh = 12, 64
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).cuda()
q, k, v = r(B, h, L, dh), r(B, h, L, dh), r(B, h, L, dh)
o = torch.empty(B, L, h * dh, dtype=torch.bfloat16, device="cuda")
for _ in range(5):
    ops.attention(q, k, v, causal=causal, out=o)
torch.cuda.synchronize()
