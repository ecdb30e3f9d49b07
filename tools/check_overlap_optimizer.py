import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import vyomai_amd as V
from vyomai_amd import recipe
from vyomai_amd.training import FlatTrainer
from tests.golden import cases
DEV = "cuda"
def run(**kw):
    cfg = cases.test_cfg()
    cfg.num_hidden_layers, cfg.vocab_size, cfg.hidden_dropout_prob = 2, 1031, 0.0
    m = V.DecoderModel(cfg, "rope", None)
    recipe.load_recipe_(m)
    m = m.to(DEV).train()
    ids = torch.from_numpy(recipe.token_ids("train.ids", (4, 48), 3, cfg.vocab_size)).to(DEV)
    tr = FlatTrainer(m, lr=1e-3, weight_decay=0.01, **kw)
    out = []
    for s in range(3):
        out.append(tr.train_step(lambda: m.clm_loss(ids, ids)).item())
    print(kw, "buckets", len(tr.reducer.buckets), tr.reducer.launch_order, out)
run(overlap_optimizer=False)
run(overlap_optimizer=True)
run(overlap_optimizer=True, bucket_bytes=1 << 30)
run(overlap_optimizer=True, bucket_bytes=8 << 20)
