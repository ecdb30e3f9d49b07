"""RCCL on the one-GPU box: a ONE-rank nccl (= RCCL) process group with the bucket collectives really issued
(VY_DDP_FORCE_COLLECTIVES=1) -- communicator creation, async all-reduce of the fp32 and of the bf16 buckets on RCCL's
stream, the side stream's wait for them, per-bucket AdamW behind them, barrier -- must leave three training steps
equal to the same trainer without a process group (a one-rank sum is the identity) up to that trainer's own run-to-run noise
(fp32 buckets) / to bf16 rounding of the gradients (bf16 buckets).    python tools/check_rccl_one_rank.py"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vyomai_amd as V
from vyomai_amd import recipe
from vyomai_amd.training import FlatTrainer

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
cfg = V.EncoderConfig(num_hidden_layers=3, hidden_size=512, num_attention_heads=8, intermediate_size=2048,
                      hidden_dropout_prob=0.0, vocab_size=1000, max_position_embeddings=1024)
ids = torch.from_numpy(recipe.token_ids("rccl.ids", (8, 512), 3, cfg.vocab_size)).to(dev)


def train(**kw):
    m = V.DecoderModel(cfg, "rope", None)
    recipe.load_recipe_(m)
    m = m.to(dev).train()
    tr = FlatTrainer(m, lr=1e-3, bucket_bytes=4 << 20, **kw)
    for _ in range(3):
        loss = tr.train_step(lambda: m.clm_loss(ids, ids))
    torch.cuda.synchronize()
    return tr.arena.master.detach().clone(), float(loss), tr

# the weight gradients are summed with fp32 atomics and AdamW turns a sign flip of a noise-level gradient into a 2 lr
# difference: two runs of the SAME trainer set the noise floor the comparison is held to
ref, loss_ref, _ = train()
ref2, _, _ = train()
noise = (ref2 - ref).abs().mean().item()
os.environ["VY_DDP_FORCE_COLLECTIVES"] = "1"
import socket
_s = socket.socket(); _s.bind(("127.0.0.1", 0)); _port = _s.getsockname()[1]; _s.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_port}", rank=0, world_size=1, device_id=dev)
got, loss_got, tr = train()
assert tr.reducer.force and len(tr.reducer.launch_order) == len(tr.reducer.buckets) > 1
# per step: one all-reduce per bucket, in bucket order, + the one-element-per-parameter flag exchange
per_step = len(tr.reducer.buckets) + 1
assert tr.reducer.launch_order == list(range(len(tr.reducer.buckets)))
assert tr.reducer.collectives == 3 * per_step, (tr.reducer.collectives, per_step)
torch_path_collectives = tr.reducer.collectives
d = (got - ref).abs()
print(f"fp32 buckets over RCCL: loss {loss_got:.6f} vs {loss_ref:.6f}; max |param diff| {d.max().item():.3e} mean {d.mean().item():.3e}")
print(f"run-to-run noise of the trainer without a group: mean {noise:.3e}")
assert d.mean().item() <= 3 * noise + 1e-6 and abs(loss_got - loss_ref) < 1e-3
got16, loss16, tr16 = train(grad_comm_dtype=torch.bfloat16)
d16 = (got16 - ref).abs()
print(f"bf16 buckets over RCCL: loss {loss16:.6f}; max |param diff| {d16.max().item():.3e} mean {d16.mean().item():.3e}")
assert d16.mean().item() < 2e-4 and abs(loss16 - loss_ref) < 2e-2
dist.barrier()
dist.destroy_process_group()
# the library's own communicator (vy_ddp_*: ncclGetUniqueId / ncclCommInitRank / ncclAllReduce on a stream it is handed),
# no torch.distributed group at all
gotn, lossn, trn = train(native_rccl=True)
assert trn.native is not None and trn.reducer.native is trn.native and trn.native.world == 1
# the native transport issued exactly the collectives the torch.distributed transport issued, every one through vy_ddp_*
assert trn.reducer.collectives == torch_path_collectives == trn.native.issued, \
    (trn.reducer.collectives, torch_path_collectives, trn.native.issued)
print(f"collectives over 3 steps: torch.distributed {torch_path_collectives}, native {trn.native.issued}")
dn = (gotn - ref).abs()
print(f"fp32 buckets over the native communicator: loss {lossn:.6f}; max |param diff| {dn.max().item():.3e} mean {dn.mean().item():.3e}")
assert dn.mean().item() <= 3 * noise + 1e-6 and abs(lossn - loss_ref) < 1e-3
trn.native.close()
print("ok")
