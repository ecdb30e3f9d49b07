"""Which host-side torch ops (copies, fills, casts) still run inside a training step, and from where:
torch.profiler over one step of the bench model, grouped by op and Python call site."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import vyomai_amd as V
from vyomai_amd import recipe
from vyomai_amd.training import FlatTrainer
from torch.profiler import profile, ProfilerActivity

sys.argv = ["bench.py"]
a = bench.parse()
cfg = bench.make_cfg(a)
dev = torch.device("cuda", 0)
model = V.DecoderModel(cfg, "rope", None)
recipe.load_recipe_(model)
model = model.to(dev).train()
trainer = FlatTrainer(model, lr=5e-5, weight_decay=0.01)
ids = torch.randint(3, cfg.vocab_size, (a.batch, a.seq), device=dev)
step = lambda: trainer.train_step(lambda: model.clm_loss(ids, ids))
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.Counter()
dur = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name.split("::")[1] in ("copy_", "fill_", "zero_", "_to_copy", "clone", "contiguous", "cat", "add_", "mul", "div", "sum", "to"):
        site = next((s for s in (e.stack or []) if "vyomai_amd" in s or "bench" in s), "?")
        agg[(e.name, site)] += 1
        dur[(e.name, site)] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
for (name, site), n in agg.most_common(40):
    print(f"{n:4d}  {dur[(name, site)]:9.1f} us  {name:18s} {site}")
