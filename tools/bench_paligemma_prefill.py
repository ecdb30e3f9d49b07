"""configs[4] prefill only (vision tower + projector + 264-row language-model prefill + the first token), N times: wall time
per call, host time to enqueue it, and -- under rocprofv3 --kernel-trace -- the kernels of a prefill.
  python tools/bench_paligemma_prefill.py [n=5]"""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import shapes as cases
from vyomai_amd.models import paligemma as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
torch.manual_seed(0)
vis = P.SiglipVisionConfig(**cases.SIGLIP)
txt = types.SimpleNamespace(**cases.GEMMA)
with torch.device(dev):
    m = P.PaliGemmaForConditionalGeneration(P.PaliGemmaShape(vis, txt, txt.hidden_size))
m = m.to(torch.bfloat16).eval()
for p in m.parameters():
    if p.dim() > 1:
        torch.nn.init.normal_(p, std=0.02)
img = torch.rand(1, 3, 224, 224, device=dev)
ids = torch.randint(3, txt.vocab_size, (1, 8), device=dev)
for _ in range(2):
    m.generate(img, ids, max_new_tokens=1, max_cache_len=384)
torch.cuda.synchronize()
walls, hosts = [], []
for _ in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.generate(img, ids, max_new_tokens=1, max_cache_len=384)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    walls.append((t2 - t0) * 1e3); hosts.append((t1 - t0) * 1e3)
walls.sort(); hosts.sort()
print(f"prefill + first token: wall median {walls[n // 2]:.2f} ms (min {walls[0]:.2f}); host returns after median {hosts[n // 2]:.2f} ms")
