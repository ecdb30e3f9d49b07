"""configs[3] smoke: ViT-B/16 + vision-language decoder, caption training step through FlatTrainer."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vyomai_amd as V
from vyomai_amd import recipe
from vyomai_amd.training import FlatTrainer
from tests.golden import cases
DEV = "cuda"
vcfg = cases.vit_cfg()
cfg = cases.test_cfg()
cfg.num_hidden_layers, cfg.hidden_dropout_prob = 12, 0.0
if hasattr(vcfg, "hidden_dropout_prob"):
    vcfg.hidden_dropout_prob = 0.0
B = int(os.environ.get("B", "64"))
vlm = V.VisionLanguageModel(cfg, V.Vit(vcfg), "rope", None)
recipe.load_recipe_(vlm)
vlm = vlm.to(DEV).train()
img = torch.from_numpy(recipe.uniform("vit.img", (B, 3, 224, 224), 0.5, 0.5)).to(DEV).to(torch.bfloat16)
ids = torch.from_numpy(recipe.token_ids("cap.ids", (B, 32), 3, cfg.vocab_size)).to(DEV)
tr = FlatTrainer(vlm, lr=1e-4)
def loss_fn():
    logits = vlm(pixel_values=img, decoder_input_ids=ids).logits          # (B, 33, V): image token first
    lg = logits[:, 1:-1].float().reshape(-1, logits.shape[-1])
    return torch.nn.functional.cross_entropy(lg, ids[:, 1:].reshape(-1))
losses = []
for s in range(6):
    if s == 2:
        torch.cuda.synchronize(); t0 = time.time()
    losses.append(tr.train_step(loss_fn).item())
torch.cuda.synchronize()
dt = (time.time() - t0) / 4
print("losses", [round(x, 4) for x in losses])
print(f"caption training B={B}: {dt*1e3:.1f} ms/step, {B/dt:.0f} images/s, {B*33/dt:.0f} decoder tokens/s")
