"""Two data-parallel ranks (gloo; both on GPU 0 of the one-GPU box) against one process on the concatenated
batch: same parameters after three AdamW steps -- with the weight gradients of a layer deferred into grouped
launches and the buckets reduced / stepped while backward is still running.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/check_ddp_gpu.py"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vyomai_amd as V
from vyomai_amd import recipe
from vyomai_amd.training import FlatTrainer

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
cfg = V.EncoderConfig(num_hidden_layers=3, hidden_size=512, num_attention_heads=8, intermediate_size=2048,
                      hidden_dropout_prob=0.0, vocab_size=1000, max_position_embeddings=1024)
B, L, STEPS = 8, 512, 3
all_ids = torch.from_numpy(recipe.token_ids("ddp.ids", (world * B, L), 3, cfg.vocab_size)).to(dev)

def make():
    m = V.DecoderModel(cfg, "rope", None)
    recipe.load_recipe_(m)
    return m.to(dev).train()

mine = all_ids[rank * B:(rank + 1) * B]
# (a) the reduced gradients of one backward pass
m0 = make()
tr0 = FlatTrainer(m0, lr=1e-3, bucket_bytes=4 << 20, overlap_optimizer=False)
tr0.zero_grad()
tr0.backward(m0.clm_loss(mine, mine))
g_dp = (tr0.arena.grad * tr0.reducer.finish()).detach().cpu()
# (b) three full steps
m = make()
tr = FlatTrainer(m, lr=1e-3, bucket_bytes=4 << 20)
for _ in range(STEPS):
    loss = tr.train_step(lambda: m.clm_loss(mine, mine))
torch.cuda.synchronize()
flat = tr.arena.master.detach().cpu()
gathered = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(gathered, flat)
if rank == 0:
    assert all(torch.equal(gathered[0], g) for g in gathered[1:]), "ranks diverged"
    dist.destroy_process_group()
    ref = make()
    # one process, both ranks' batches: gradient accumulation over `world` micro-batches = the average DDP takes
    tr2 = FlatTrainer(ref, lr=1e-3, bucket_bytes=4 << 20, accumulate_steps=world)
    for _ in range(STEPS):
        for r in range(world):
            ids = all_ids[r * B:(r + 1) * B]
            tr2.train_step(lambda: ref.clm_loss(ids, ids))
    assert tr2.step_count == STEPS
    torch.cuda.synchronize()
    want = tr2.arena.master.detach().cpu()
    err = (flat - want).abs().max().item()
    # AdamW moves every element by ~lr per step whatever the size of its gradient, so elements whose gradient is
    # rounding noise may differ by 2 * lr per step; everything else must agree closely
    print(f"ranks identical; vs single process on both batches: max |dp - ref| = {err:.3e} (parameters up to {want.abs().max().item():.2f})")
    print("mean |dp - ref| =", (flat - want).abs().mean().item())
    assert (flat - want).abs().mean().item() < 1e-5
    ref0 = make()
    tr3 = FlatTrainer(ref0, lr=1e-3, bucket_bytes=4 << 20, overlap_optimizer=False, accumulate_steps=world)
    tr3.zero_grad()
    for r in range(world):
        ids = all_ids[r * B:(r + 1) * B]
        tr3.backward(ref0.clm_loss(ids, ids))
    torch.cuda.synchronize()
    g_ref = tr3.arena.grad.detach().cpu()
    gerr = (g_dp - g_ref).abs().max().item() / g_ref.abs().max().item()
    print(f"reduced gradient vs accumulated gradient: max relative error {gerr:.3e}")
    assert gerr < 2e-2, gerr
    assert err <= 2 * 1e-3 * STEPS + 1e-4, err
    print("ok")
else:
    dist.destroy_process_group()
