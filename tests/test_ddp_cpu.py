"""N>1 path on CPU: the flat-arena bucket reducer over gloo, world_size 2 (no GPU needed)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.unused = nn.Linear(4, 4)  # never receives a gradient (find_unused_parameters case)
        self.a = nn.Linear(16, 32)
        self.b = nn.Linear(32, 8)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vyomai_amd.training import BucketReducer, FlatArena
    torch.manual_seed(0)
    model = Tiny()
    ref = Tiny()
    ref.load_state_dict(model.state_dict())
    arena = FlatArena(model, shadow_dtype=None)
    red = BucketReducer(arena, bucket_bytes=1024)  # several small buckets
    assert len(red.buckets) >= 2
    # parameters are views of the arena and still hold their values
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert torch.equal(p, q), n
        assert p.data_ptr() >= arena.master.data_ptr()
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(5, 16, generator=g)
    arena.zero_grad()
    red.reset()
    model(x).pow(2).sum().backward()
    scale = red.finish()
    assert scale == 1.0 / world
    # every bucket went out exactly once, the last-registered parameters first
    assert sorted(red.launch_order) == list(range(len(red.buckets)))
    pos = {b: i for i, b in enumerate(red.launch_order)}
    # backward order: the head (b) is reduced no later than the first layer (a); the bucket of the
    # parameter that never got a gradient is flushed by finish()
    assert pos[red.bucket_of[id(model.b.weight)]] <= pos[red.bucket_of[id(model.a.weight)]]
    # a bucket holding a never-used parameter can only go out at finish(): last
    assert pos[red.bucket_of[id(model.unused.weight)]] == len(red.buckets) - 1
    # reference: average of both ranks' gradients computed without the reducer
    grads = []
    for r in range(world):
        gg = torch.Generator().manual_seed(100 + r)
        xr = torch.randn(5, 16, generator=gg)
        ref.zero_grad()
        ref(xr).pow(2).sum().backward()
        grads.append([p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in ref.parameters()])
    for i, p in enumerate(model.parameters()):
        want = sum(gr[i] for gr in grads) / world
        assert torch.allclose(p.grad * scale, want, atol=1e-6), i
    if rank == 0:
        out.put("ok")
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bucket_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"


def test_bucket_waits_for_every_parameter_once():
    """A parameter may report readiness twice in one backward (the kernel-side notification AND the
    autograd post-accumulate hook): it must count once, or a bucket is reduced -- and, with the
    overlapped optimizer, stepped -- before its last gradients exist."""
    from vyomai_amd.training import BucketReducer, FlatArena
    torch.manual_seed(0)
    model = Tiny()
    arena = FlatArena(model, shadow_dtype=None)
    red = BucketReducer(arena, bucket_bytes=1 << 30)   # one bucket
    assert len(red.buckets) == 1
    fired = []
    red.on_bucket = lambda b, work: fired.append(b)
    red.reset()
    params = list(model.parameters())
    for p in params[:-1]:
        red.mark_ready(p)
        red.mark_ready(p)          # duplicate report
    assert fired == [], "bucket went out before its last parameter was ready"
    red.mark_ready(params[-1])
    assert fired == [0]
    red.mark_ready(params[-1])     # late duplicate: ignored
    assert fired == [0]
