import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import vyomai_amd as V
from vyomai_amd import recipe
from vyomai_amd.training import FlatTrainer
from tests.golden import cases
DEV="cuda"
def T(a): return torch.from_numpy(a)
for at in (None, "gqa"):
    cfg = cases.with_kv(cases.test_cfg(), at); cfg.num_hidden_layers, cfg.vocab_size, cfg.hidden_dropout_prob = 2, 1031, 0.0
    m = V.EncoderForMaskedLM(cfg, "absolute", at); recipe.load_recipe_(m); m = m.to(DEV).train()
    ids = T(recipe.token_ids("mlm.ids", (4, 50), 3, cfg.vocab_size)).to(DEV); am = T(cases.keypad(4, 50)).to(DEV)
    labels = ids.clone(); labels[:, ::3] = -100
    tr = FlatTrainer(m, lr=1e-3)
    def loss_fn():
        out = m(ids, am)
        return torch.nn.functional.cross_entropy(out.logits.float().reshape(-1, cfg.vocab_size), labels.reshape(-1), ignore_index=-100)
    ls = [tr.train_step(loss_fn).item() for _ in range(5)]
    print("MLM", at, [round(x,3) for x in ls]); assert ls[-1] < ls[0]
    s2s = V.EncoderDecoderModel.from_config(cfg, cfg, None, "rope", at, "rope", at); recipe.load_recipe_(s2s); s2s = s2s.to(DEV).train()
    tgt = T(recipe.token_ids("s2s.tgt", (4, 32), 3, cfg.vocab_size)).to(DEV)
    tr = FlatTrainer(s2s, lr=2e-3)
    ls = [tr.train_step(lambda: s2s.seq2seq_loss(ids, am, tgt, tgt)).item() for _ in range(5)]
    print("S2S", at, [round(x,3) for x in ls]); assert ls[-1] < ls[0]
print("ok")
