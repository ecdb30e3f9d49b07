"""N training steps of configs[1] and nothing else (for rocprofv3 --kernel-trace --stats: is the step GPU-bound?):
  python tools/train_only.py [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vyomai_amd as V
from vyomai_amd import recipe
from vyomai_amd.training import FlatTrainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = V.EncoderConfig(num_hidden_layers=12, max_position_embeddings=1024, hidden_dropout_prob=0.0)
m = V.DecoderModel(cfg, "rope", None)
recipe.load_recipe_(m)
m = m.to("cuda").train()
tr = FlatTrainer(m, lr=5e-5, weight_decay=0.01)
torch.manual_seed(1234)
ids = torch.randint(3, cfg.vocab_size, (32, 512), device="cuda")
for _ in range(2):
    tr.train_step(lambda: m.clm_loss(ids, ids))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.train_step(lambda: m.clm_loss(ids, ids))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"train_only: {steps} steps, {dt / steps * 1e3:.3f} ms per step (+ 2 warm-up steps)")
