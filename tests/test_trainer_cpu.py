"""Host logic of the data-parallel trainer and of the compute-dtype weight copies, on CPU (no kernels):
gradient accumulation with world_size 2 over gloo, frozen / unused parameters, the stale-state guards,
and the two ways a cached bf16 copy of a parameter used to go stale."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from tests.golden import cases


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.unused = nn.Linear(8, 8)   # never receives a gradient
        self.a = nn.Linear(16, 32)
        self.frozen = nn.Linear(32, 32)
        self.b = nn.Linear(32, 8)
        for p in self.frozen.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        return self.b(self.frozen(torch.tanh(self.a(x))))


def _torch_adamw(tr):
    """Stand-in for the vy_adamw_step launch (GPU only): torch.optim.AdamW's arithmetic on an arena range."""
    def step(lo, hi, step_no, scale_dev=None, gate=None):
        if gate is not None and float(gate) == 0.0:
            return
        a = tr.arena
        g = a.grad[lo:hi] * tr._scale
        if scale_dev is not None:
            g = g * scale_dev
        b1, b2 = tr.betas
        p = a.master[lo:hi]
        p.mul_(1 - tr.lr * tr.weight_decay)
        tr.m[lo:hi].mul_(b1).add_(g, alpha=1 - b1)
        tr.v[lo:hi].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (tr.v[lo:hi].sqrt() / (1 - b2 ** step_no) ** 0.5).add_(tr.eps)
        p.addcdiv_(tr.m[lo:hi], denom, value=-tr.lr / (1 - b1 ** step_no))
        if a.shadow is not None:
            a.shadow[lo:hi].copy_(p)
    return step


def _make_trainer(model, **kw):
    from vyomai_amd.training import FlatTrainer
    tr = FlatTrainer(model, lr=1e-2, weight_decay=0.1, **kw)
    tr._adamw = _torch_adamw(tr)
    tr._sumsq = lambda: tr.arena.grad.double().pow(2).sum().float()
    return tr


def test_frozen_and_unused_parameters_are_not_touched():
    torch.manual_seed(0)
    model = Tiny()
    ref = Tiny()
    ref.load_state_dict(model.state_dict())
    tr = _make_trainer(model)
    opt = torch.optim.AdamW([p for p in ref.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.1)
    x = torch.randn(6, 16)
    for _ in range(3):
        tr.train_step(lambda: model(x).pow(2).mean())
        opt.zero_grad()
        ref(x).pow(2).mean().backward()
        opt.step()
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if n.startswith(("frozen", "unused")):
            assert torch.equal(p, q), f"{n} changed although it never had a gradient"   # bit-identical
        else:
            assert torch.allclose(p, q, atol=1e-6), n
    # the frozen weights were not decayed in the kernels' bf16 copy either
    from vyomai_amd.layers.attention import _shadow
    assert torch.equal(_shadow(model.frozen.weight, torch.bfloat16).float(), ref.frozen.weight.bfloat16().float())


def test_backward_without_zero_grad_raises():
    torch.manual_seed(0)
    model = Tiny()
    tr = _make_trainer(model)
    x = torch.randn(4, 16)
    tr.zero_grad()
    tr.backward(model(x).pow(2).mean())
    assert tr.optimizer_step() is True
    with pytest.raises(RuntimeError, match="zero_grad"):
        tr.backward(model(x).pow(2).mean())
    tr.zero_grad()
    tr.backward(model(x).pow(2).mean())   # fine again


def test_gradient_clipping_matches_torch():
    torch.manual_seed(0)
    model = Tiny()
    ref = Tiny()
    ref.load_state_dict(model.state_dict())
    tr = _make_trainer(model, max_grad_norm=0.05)
    opt = torch.optim.AdamW([p for p in ref.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.1)
    x = torch.randn(6, 16) * 3
    for _ in range(2):
        tr.train_step(lambda: model(x).pow(2).mean())
        opt.zero_grad()
        ref(x).pow(2).mean().backward()
        norm = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.05)
        opt.step()
        assert torch.allclose(tr.last_grad_norm, norm, rtol=1e-5)
        assert norm > 0.05   # the clip is active in this test
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p, q, atol=1e-6), n


def _accum_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = Tiny()
    ref = Tiny()
    ref.load_state_dict(model.state_dict())
    tr = _make_trainer(model, accumulate_steps=2, bucket_bytes=1024,
                       grad_comm_dtype=None if os.environ.get("VY_TEST_COMM") != "bf16" else torch.bfloat16)
    assert len(tr.reducer.buckets) >= 2
    g = torch.Generator().manual_seed(7)
    data = torch.randn(3, 2, world, 5, 16, generator=g)   # [step][micro][rank]
    opt = torch.optim.AdamW([p for p in ref.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.1)
    for step in range(3):
        for micro in range(2):
            applied_before = tr.step_count
            tr.train_step(lambda: model(data[step, micro, rank]).pow(2).mean())
            # the update happens on the second micro-step only, and only then is anything reduced
            assert tr.step_count == applied_before + (1 if micro == 1 else 0)
        # one rank on the concatenated batch: mean over 2 micro x 2 ranks of equally sized batches
        opt.zero_grad()
        xs = data[step].reshape(-1, 16)
        ref(xs).pow(2).mean().backward()
        if tr.reducer.comm_dtype is not None:
            # bf16 exchange: the averaged gradients agree to bf16 rounding of each rank's contribution (the
            # parameters do not have to: Adam turns a sum that cancels to 0 in bf16 into a missing +-lr step)
            for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
                if q.grad is not None:
                    scale = max(float(q.grad.abs().max()), 1e-6)
                    assert torch.allclose(p.grad * tr._scale, q.grad, atol=1.6e-2 * scale), (n, step)
            ref.load_state_dict(model.state_dict())   # keep the two in step for the next round
            continue
        opt.step()
    tol = 1e-6
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p, q, atol=tol), (n, (p - q).abs().max())
    if rank == 0:
        out.put("ok")
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("comm", ["fp32", "bf16"])
def test_accumulate_2_world_2_equals_one_rank_on_the_concatenated_batch(comm):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    os.environ["VY_TEST_COMM"] = comm
    try:
        procs = [ctx.Process(target=_accum_worker, args=(r, 2, port, q)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(150)
            assert p.exitcode == 0
    finally:
        os.environ.pop("VY_TEST_COMM", None)
    assert q.get(timeout=5) == "ok"


class Branchy(nn.Module):
    """`side` is used only when the batch asks for it -- e.g. a vision tower that a text-only batch skips."""
    def __init__(self):
        super().__init__()
        self.nobody = nn.Linear(8, 8)   # used by no rank at all
        self.a = nn.Linear(16, 32)
        self.side = nn.Linear(16, 32)
        self.b = nn.Linear(32, 8)

    def forward(self, x, use_side):
        h = torch.tanh(self.a(x))
        if use_side:
            h = h + self.side(x)
        return self.b(h)


def _unused_on_one_rank_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = Branchy()
    ref = Branchy()
    ref.load_state_dict(model.state_dict())
    nobody0 = model.nobody.weight.detach().clone()
    tr = _make_trainer(model, bucket_bytes=1024)
    assert len(tr.reducer.buckets) >= 3
    g = torch.Generator().manual_seed(11)
    data = torch.randn(3, world, 5, 16, generator=g)
    opt = torch.optim.AdamW([p for n, p in ref.named_parameters() if not n.startswith("nobody")], lr=1e-2, weight_decay=0.1)
    for step in range(3):
        # rank 1's batch never reaches `side`; rank 0's does
        tr.train_step(lambda: model(data[step, rank], use_side=(rank == 0)).pow(2).mean())
        # every rank issued its collectives in bucket order although they became ready in different orders
        assert tr.reducer.launch_order == list(range(len(tr.reducer.buckets))), tr.reducer.launch_order
        # what torch DDP (find_unused_parameters=True) computes: the mean over ranks, a rank without the branch counting 0
        opt.zero_grad()
        loss = sum(ref(data[step, r], use_side=(r == 0)).pow(2).mean() for r in range(world)) / world
        loss.backward()
        opt.step()
    # the replicas stay bit-identical: gather rank 1's masters on rank 0
    mine = tr.arena.master.clone()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    assert torch.equal(both[0], both[1]), "the ranks' parameters diverged"
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p, q, atol=1e-6), (n, (p - q).abs().max())
    assert torch.equal(model.nobody.weight, nobody0), "a parameter no rank touched was updated (weight decay)"
    # a second backward before optimizer_step() would add local gradients on top of a reduced sum: refused
    tr.zero_grad()
    tr.backward(model(data[0, rank], True).pow(2).mean())
    with pytest.raises(RuntimeError, match="accumulate_steps"):
        tr.backward(model(data[0, rank], True).pow(2).mean())
    if rank == 0:
        out.put("ok")
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_parameter_unused_on_one_rank_keeps_the_replicas_identical():
    """the touched set was rank-local, so a parameter one rank's batch skipped was updated on the other
    ranks only; and buckets went out in readiness order, which differs between such ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_unused_on_one_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
        assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"


def test_second_backward_without_accumulate_steps_is_allowed_only_where_it_is_harmless():
    """One rank, no overlapped optimizer, no collectives: the gradients simply add up (plain torch semantics)."""
    torch.manual_seed(0)
    model = Tiny()
    tr = _make_trainer(model)
    x = torch.randn(4, 16)
    tr.zero_grad()
    tr.backward(model(x).pow(2).mean())
    g1 = tr.arena.grad.clone()
    tr.backward(model(x).pow(2).mean())
    assert torch.allclose(tr.arena.grad, 2 * g1, atol=1e-6)


# ---- compute-dtype copies of the weights (layers.attention._shadow / _packed_shadow) -------------

def test_packed_qkv_copy_follows_in_place_updates_of_the_members():
    """fp32 parameters, bf16 kernels, a plain torch optimizer: optimizer.step() writes query/key/value.weight
    in place; those writes never bump the version counter of the packed [Wq;Wk;Wv] buffer, on which the bf16
    copy used to be keyed -- the forward projection kept reading the old weights."""
    import vyomai_amd as V
    cfg = cases.micro_cfg()
    att = V.layers.attention.EncoderAttention(cfg, 0) if hasattr(V, "layers") else None
    from vyomai_amd.layers.attention import EncoderAttention
    att = EncoderAttention(cfg, 0)
    w0, _ = att._packed_shadow(torch.bfloat16)
    w0 = w0.clone()
    with torch.no_grad():
        att.query.weight.add_(1.0)            # what optimizer.step() / load_state_dict do
        att.value.bias.mul_(0.0)
    w1, b1 = att._packed_shadow(torch.bfloat16)
    d = cfg.hidden_size
    assert torch.allclose(w1[:d].float(), w0[:d].float() + 1.0, atol=2e-2)
    assert torch.equal(w1[d:], w0[d:])
    assert torch.count_nonzero(b1[2 * d:]) == 0
    # and the cache is reused while nothing changes
    assert att._packed_shadow(torch.bfloat16)[0] is w1


def test_arena_copy_is_refreshed_in_place_after_load_state_dict():
    """A model owned by FlatTrainer: load_state_dict bumps every parameter's version.  The bf16 copy must stay
    the trainer's arena view (refreshed in place) -- a fresh detached copy would never see the fused AdamW
    kernel's updates again."""
    import vyomai_amd as V
    from vyomai_amd.layers.attention import _shadow
    torch.manual_seed(0)
    cfg = cases.micro_cfg()
    cfg.num_hidden_layers = 1
    model = V.DecoderModel(cfg, "rope", None)
    tr = _make_trainer(model)
    lay = model.all_layer[0]
    w = lay.feed_forward.out.weight
    view_before = _shadow(w, torch.bfloat16)
    pw_before, _ = lay.attention._packed_shadow(torch.bfloat16)
    sd = {k: v.clone() + 0.5 for k, v in model.state_dict().items()}
    tr.load_state_dict(sd)
    view_after = _shadow(w, torch.bfloat16)
    assert view_after.data_ptr() == view_before.data_ptr(), "the arena view was replaced by a detached copy"
    lo = tr.arena.shadow.data_ptr()
    assert lo <= view_after.data_ptr() < lo + tr.arena.shadow.numel() * 2
    assert torch.allclose(view_after.float(), w.detach().bfloat16().float())
    assert torch.allclose(w, sd["all_layer.0.feed_forward.out.weight"])
    pw_after, _ = lay.attention._packed_shadow(torch.bfloat16)
    assert pw_after.data_ptr() == pw_before.data_ptr()
    assert torch.allclose(pw_after[: cfg.hidden_size].float(), lay.attention.query.weight.detach().bfloat16().float())
    # a direct in-place edit without the trainer's helper is picked up too (version bump -> in-place refresh)
    with torch.no_grad():
        w.mul_(2.0)
    again = _shadow(w, torch.bfloat16)
    assert again.data_ptr() == view_before.data_ptr()
    assert torch.allclose(again.float(), w.detach().bfloat16().float())


def test_zero_grad_after_an_unstepped_backward_starts_a_fresh_window():
    """backward() without optimizer_step(), then train_step(): zero_grad() drops the finished-but-unstepped window (it
    used to count as 'mid-accumulation' and return without clearing anything)."""
    torch.manual_seed(0)
    model = Tiny()
    tr = _make_trainer(model)
    x = torch.randn(4, 16)
    tr.zero_grad()
    tr.backward(model(x).pow(2).mean())
    g1 = tr.arena.grad.clone()
    assert g1.abs().sum() > 0
    tr.zero_grad()
    assert tr.arena.grad.abs().sum() == 0 and tr._micro == 0
    tr.train_step(lambda: model(x).pow(2).mean())          # a complete step, no complaint about a second backward
    assert tr.step_count == 1


def test_arena_parameters_start_on_128_byte_lines_of_the_bf16_copy():
    """Every parameter (except the followers inside a packed q/k/v group, which must stay adjacent) starts at a multiple of 64
    elements: the odd-sized vocabulary bias used to push the LM head's matrices half a cache line off."""
    import vyomai_amd as V
    from vyomai_amd.training import FlatArena, LINE, _ordered_params
    cfg = cases.micro_cfg()
    cfg.vocab_size = 1031                     # odd, like 50265
    m = V.DecoderModel(cfg, "rope", "gqa")
    arena = FlatArena(m, shadow_dtype=torch.bfloat16)
    followers = _ordered_params.followers
    names = {id(p): n for n, p in arena.items}
    assert any("lm_head.decoder.weight" in n for n in names.values())
    for p, o in zip(arena.params, arena.offsets):
        if id(p) not in followers:
            assert o % LINE == 0, (names[id(p)], o)
    att = m.all_layer[0].attention
    w, b = att._packed()                       # the packed views still cover the three members without gaps
    assert w.shape[0] == att.query.weight.shape[0] + att.key.weight.shape[0] + att.value.weight.shape[0]
    assert att.key.weight.data_ptr() == att.query.weight.data_ptr() + att.query.weight.numel() * 4
