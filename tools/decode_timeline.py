"""Where does a decode step's time go INSIDE each launch?  The decode-only kernels (vy_decode.hip) can stamp the
100 MHz wall clock at five points per workgroup (vy_debug_set_decode_stamps); this runs one eager step of the
12-layer model with stamps on and prints, per launch: the gap since the previous stamped launch's last
workgroup ended (launch boundary + whatever un-stamped kernels ran in between, e.g. attention), and the median
workgroup's timeline -- start -> loads issued -> products done -> after the barrier -> stores drained.
  python tools/decode_timeline.py [--batch 32] [--ctx 576] [--graph 0]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vyomai_amd as V  # noqa: E402
from vyomai_amd import _lib, recipe  # noqa: E402
from vyomai_amd.decode_plan import DecodePlan  # noqa: E402
from vyomai_amd.layers.kv_cache import StaticCacheOne  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--ctx", type=int, default=576)
ap.add_argument("--layers", type=int, default=12)
a = ap.parse_args()
cfg = V.EncoderConfig(num_hidden_layers=a.layers, max_position_embeddings=1024, hidden_dropout_prob=0.0)
m = V.DecoderModel(cfg, "rope", None)
recipe.load_recipe_(m)
m = m.to("cuda").to(torch.bfloat16).eval()
cache = StaticCacheOne(cfg, max_cache_len=a.ctx + 64, batch_size=a.batch, dtype=torch.bfloat16)
plan = DecodePlan(m, cache, a.batch, torch.bfloat16, torch.device("cuda", 0))
for i in range(len(cache.key_cache)):
    cache.key_cache[i].normal_()
    cache.value_cache[i].normal_()
x = torch.randn(a.batch, cfg.hidden_size, device="cuda", dtype=torch.bfloat16)
logits = torch.empty(a.batch, plan.ldv, device="cuda", dtype=torch.bfloat16)
lib = _lib.load()
lib.vy_debug_set_decode_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int]
for _ in range(5):
    plan._launch(x.data_ptr(), a.ctx, None, None, logits.data_ptr())
torch.cuda.synchronize()
NL, MW = 8 * a.layers + 8, 256
buf = torch.zeros(NL, MW, 8, dtype=torch.int64, device="cuda")
lib.vy_debug_set_decode_stamps(buf.data_ptr(), NL, MW)
plan._launch(x.data_ptr(), a.ctx, None, None, logits.data_ptr())
torch.cuda.synchronize()
lib.vy_debug_set_decode_stamps(None, 0, 0)
s = buf.cpu().numpy().astype(np.int64)
prev_end = None
t_first = None
print("launch  wgs   gap_us | median workgroup (us from its start): args  addr  issued  products  barrier  end | launch span  start spread")
tot = {}
for l in range(NL):
    rows = s[l][s[l][:, 0] > 0]
    if len(rows) == 0:
        continue
    st, en = rows[:, 0].min(), rows[:, 4].max()
    if t_first is None:
        t_first = st
    gap = (st - prev_end) / 100.0 if prev_end is not None else 0.0
    def rel(col):   # median over the workgroups of (stamp - start); a kernel that has no such stamp leaves 0 there
        return np.median(rows[:, col] - rows[:, 0]) / 100.0 if rows[:, col].max() > 0 else float("nan")
    med = [rel(c_) for c_ in (1, 2, 3, 4)]
    ma, mb = rel(5), rel(6)
    print(f"{l:4d}  {len(rows):4d}  {gap:7.2f} | {ma:5.2f} {mb:5.2f} {med[0]:6.2f} {med[1]:8.2f} {med[2]:8.2f} {med[3]:6.2f} | "
          f"{(en - st) / 100.0:7.2f}  {(rows[:, 0].max() - st) / 100.0:6.2f}")
    prev_end = en
print(f"first stamped start -> last stamped end: {(prev_end - t_first) / 100.0:.1f} us")
