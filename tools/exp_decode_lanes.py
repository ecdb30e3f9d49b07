"""Experiment: the native decode step (B=32) as N independent sub-batches on N streams inside ONE hipGraph (fork / join by
events) against the one-stream step -- a step is a chain of ~88 latency-bound launches; do independent chains overlap?
  python tools/exp_decode_lanes.py [--batch 32] [--ctx 576]"""
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
a = ap.parse_args()
cfg = V.EncoderConfig(num_hidden_layers=12, max_position_embeddings=1024, hidden_dropout_prob=0.0)
if a.attn == "gqa":
    cfg.num_key_value_heads = 4
m = V.DecoderModel(cfg, "rope", None if a.attn == "none" else "gqa")
recipe.load_recipe_(m)
m = m.to("cuda").to(torch.bfloat16).eval()
dev = torch.device("cuda", 0)


def ev_time(fn, iters):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


res = {}
for lanes in (1, 2, 4, 1, 2, 4):
    b = a.batch // lanes
    plans, xs, lgs = [], [], []
    for l in range(lanes):
        cache = StaticCacheOne(cfg, max_cache_len=a.ctx + 64, batch_size=b, dtype=torch.bfloat16)
        for i in range(len(cache.key_cache)):
            cache.key_cache[i].normal_(); cache.value_cache[i].normal_()
        p = DecodePlan(m, cache, b, torch.bfloat16, dev)
        plans.append(p)
        xs.append(torch.randn(b, cfg.hidden_size, device="cuda", dtype=torch.bfloat16))
        lgs.append(torch.empty(b, p.ldv, device="cuda", dtype=torch.bfloat16))
    pos_dev = torch.full((1,), a.ctx, dtype=torch.int32, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(lanes - 1)]

    def step():
        main = torch.cuda.current_stream()
        ev = main.record_event()
        for l in range(1, lanes):
            streams[l - 1].wait_event(ev)
            with torch.cuda.stream(streams[l - 1]):
                plans[l]._launch(xs[l].data_ptr(), a.ctx, pos_dev.data_ptr(), None, lgs[l].data_ptr())
        plans[0]._launch(xs[0].data_ptr(), a.ctx, pos_dev.data_ptr(), None, lgs[0].data_ptr())
        for l in range(1, lanes):
            main.wait_stream(streams[l - 1])

    step(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    t = ev_time(g.replay, a.iters)
    res.setdefault(lanes, []).append(t)
    print(f"{lanes} lane(s) of B={b}: graph replay {t:.1f} us per step")
    del g, plans
print({k: [round(x, 1) for x in v] for k, v in res.items()})
