"""Same-process A/B of the configs[1] training step under two settings of an environment knob that the host code
reads per call (e.g. VY_LANES), interleaved rounds (devices and clocks differ between boxes and over time):
  python tools/ab_env.py VY_LANES 0 1 [--rounds 6] [--steps 6] [--attn gqa] [--fwd-only]"""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vyomai_amd as V
from vyomai_amd import recipe
from vyomai_amd.training import FlatTrainer

ap = argparse.ArgumentParser()
ap.add_argument("knob"); ap.add_argument("a"); ap.add_argument("b")
ap.add_argument("--rounds", type=int, default=6); ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--attn", default="none"); ap.add_argument("--fwd-only", action="store_true")
a = ap.parse_args()
cfg = V.EncoderConfig(num_hidden_layers=12, max_position_embeddings=1024, hidden_dropout_prob=0.0)
if a.attn == "gqa":
    cfg.num_key_value_heads = 4
m = V.DecoderModel(cfg, "rope", None if a.attn == "none" else "gqa")
recipe.load_recipe_(m)
m = m.to("cuda").train()
tr = FlatTrainer(m, lr=5e-5, weight_decay=0.01)
torch.manual_seed(1234)
ids = torch.randint(3, cfg.vocab_size, (32, 512), device="cuda")


def run(val, n):
    os.environ[a.knob] = val
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if a.fwd_only:
            loss = m.clm_loss(ids, ids)
            del loss
        else:
            tr.train_step(lambda: m.clm_loss(ids, ids))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


run(a.a, 2); run(a.b, 2)
res = {a.a: [], a.b: []}
for r in range(a.rounds):
    for v in (a.a, a.b):
        res[v].append(run(v, a.steps))
for v, ts in res.items():
    ts = sorted(ts)
    print(f"{a.knob}={v}: median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f}  max {ts[-1]:.3f}   ({'forward only' if a.fwd_only else 'training step'})")
