"""The native decode step (vy_decoder_step: 12 layers + LM head, B=32, context 512+) alone: eager launches and
hipGraph replay, microseconds per step from events -- no embedding lookup, token pick or Python in the timed
region.  Under rocprofv3 --kernel-trace --stats this gives the per-kernel durations of one step.
  python tools/bench_decode_step.py [--batch 32] [--ctx 576] [--iters 50]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vyomai_amd as V
from vyomai_amd import recipe
from vyomai_amd.decode_plan import DecodePlan
from vyomai_amd.layers.kv_cache import StaticCacheOne

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--ctx", type=int, default=576)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--attn", default="none")
ap.add_argument("--cap", type=int, default=0, help="static cache capacity (default ctx + 64)")
a = ap.parse_args()
cfg = V.EncoderConfig(num_hidden_layers=12, max_position_embeddings=1024, hidden_dropout_prob=0.0)
if a.attn == "gqa":
    cfg.num_key_value_heads = 4
m = V.DecoderModel(cfg, "rope", None if a.attn == "none" else "gqa")
recipe.load_recipe_(m)
m = m.to("cuda").to(torch.bfloat16).eval()
cache = StaticCacheOne(cfg, max_cache_len=a.cap or a.ctx + 64, batch_size=a.batch, dtype=torch.bfloat16)
plan = DecodePlan(m, cache, a.batch, torch.bfloat16, torch.device("cuda", 0))
for i in range(len(cache.key_cache)):
    cache.key_cache[i].normal_()
    cache.value_cache[i].normal_()
x = torch.randn(a.batch, cfg.hidden_size, device="cuda", dtype=torch.bfloat16)
logits = torch.empty(a.batch, plan.ldv, device="cuda", dtype=torch.bfloat16)
pos_dev = torch.full((1,), a.ctx, dtype=torch.int32, device="cuda")


def ev_time(fn, iters):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


eager = ev_time(lambda: plan._launch(x.data_ptr(), a.ctx, None, None, logits.data_ptr()), a.iters)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    plan._launch(x.data_ptr(), a.ctx, pos_dev.data_ptr(), None, logits.data_ptr())
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    plan._launch(x.data_ptr(), a.ctx, pos_dev.data_ptr(), None, logits.data_ptr())
graph = ev_time(g.replay, a.iters)
d, L = cfg.hidden_size, cfg.num_hidden_layers
hk = getattr(cfg, "num_key_value_heads", cfg.num_attention_heads) if a.attn == "gqa" else cfg.num_attention_heads
wbytes = (L * 12 * d * d + d * d + cfg.vocab_size * d) * 2
kvbytes = 2 * L * a.batch * (a.ctx + 1) * hk * (d // cfg.num_attention_heads) * 2
print(f"decode step B={a.batch} ctx={a.ctx}: eager {eager:.1f} us, graph replay {graph:.1f} us; "
      f"algorithmic bytes {1e-6 * (wbytes + kvbytes):.0f} MB -> {(wbytes + kvbytes) / graph * 1e-6:.2f} TB/s in the graph "
      f"({(wbytes + kvbytes) / graph * 1e-6 / 8 * 100:.1f}% of 8 TB/s)")
