"""BASELINE.json configs[4]: PaliGemma-shape model (SigLIP-So400m 27L d=1152 + Gemma-2B 18L d=2048, MQA dh=256,
vocabulary 257216), random-init bf16, one 224x224 image + 8 text tokens, KV-cache greedy decode of 64 tokens
on one MI355X (Examples/paligemma.ipynb cell 30: max_cache_len 384).  Reports prefill and decode rates."""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden import cases
from vyomai_amd.models import paligemma as P

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
nparam = sum(p.numel() for p in m.parameters())
img = torch.rand(1, 3, 224, 224, device=dev)
ids = torch.randint(3, txt.vocab_size, (1, 8), device=dev)

def run(n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = m.generate(img, ids, max_new_tokens=n, max_cache_len=384)
    torch.cuda.synchronize()
    return time.perf_counter() - t, out

run(4)
t1, _ = run(1)
t64, out = run(64)
per = (t64 - t1) / 63
print(f"PaliGemma shape: {nparam/1e9:.2f} G parameters (bf16 {nparam*2/2**30:.1f} GiB); vision tower + 264-token prefill + first token "
      f"{t1*1e3:.1f} ms; decode {per*1e3:.3f} ms/token = {1/per:.1f} tokens/s (B=1); weight stream per token "
      f"{(nparam - txt.vocab_size*txt.hidden_size*0 - sum(p.numel() for p in m.vision_tower.parameters()))*2/per*1e-12:.2f} TB/s")
