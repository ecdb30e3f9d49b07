import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import ops
from tools.bench_kernels import timeit
g = torch.Generator().manual_seed(0)
V, d, M = 50265, 768, 16384
dout = torch.randn(M, d, generator=g).to(torch.bfloat16).cuda()
ids = torch.randint(0, V, (M,), generator=g).cuda()
dw = torch.zeros(V, d, device="cuda")
print("embedding bwd us", timeit(lambda: ops.embedding_bwd_(dout, ids, dw, None), 20))
table = torch.randn(V, d, generator=g).to(torch.bfloat16).cuda()
print("embedding fwd us", timeit(lambda: ops.embedding(table, ids), 20))
