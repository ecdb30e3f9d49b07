"""Decode-only timing: 512-token prompts, B=32, static cache, N new tokens."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vyomai_amd as V
from vyomai_amd import recipe

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--prompt", type=int, default=512)
ap.add_argument("--new", type=int, default=64)
a = ap.parse_args()
cfg = V.EncoderConfig(num_hidden_layers=12, max_position_embeddings=1024, hidden_dropout_prob=0.0)
m = V.DecoderModel(cfg, "rope", None)
recipe.load_recipe_(m)
m = m.to("cuda").to(torch.bfloat16).eval()
ids = torch.randint(3, cfg.vocab_size, (a.batch, a.prompt), device="cuda")
am = torch.ones_like(ids)
def gen(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    m.generate(ids, am, max_len=n, use_cache=True, use_static_cache=True)
    torch.cuda.synchronize(); return time.perf_counter() - t
gen(2); t1 = gen(1); tn = gen(a.new)
per = (tn - t1) / (a.new - 1)
print(f"prefill+1: {t1*1e3:.2f} ms; per token step: {per*1e3:.3f} ms -> {a.batch/per:.0f} tok/s")
