"""Dropout of the two hidden-state sites (reference layers/attention.py:70, layers/ffn.py:38; p = 0.1 in
EncoderConfig()), gradient clipping, the out-of-range label guard and default-config training, on the MI355X.

The dropout mask is a pure function of (seed, offset, row, column) that the forward epilogue, the backward
pass and vy_dropout all regenerate: the tests export it with vy_dropout(ones) and hand it to the CPU oracle /
torch autograd, so fused-dropout outputs AND gradients are compared value for value, not just statistically."""
import math

import numpy as np
import pytest
import torch

from oracle import vyom_oracle as O
from tests.golden import cases
from vyomai_amd import recipe

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + 131 * len(shape) + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def mask_of(M, N, p, seed, off):
    from vyomai_amd import ops
    return ops.dropout(torch.ones(M, N, dtype=torch.float32, device=DEV), p, seed, off).cpu()


def test_mask_statistics_and_limits():
    from vyomai_amd import ops
    M, N, p = 512, 768, 0.1
    m = mask_of(M, N, p, 1234, 1)
    vals = torch.unique(m)
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1.0 / (1 - p)) < 1e-6
    keep = float((m != 0).double().mean())
    sigma = math.sqrt(p * (1 - p) / (M * N))
    assert abs(keep - (1 - p)) < 5 * sigma + 1e-5, keep          # threshold is round(p * 65536) / 65536
    # rows and columns are not correlated with the counter layout (8-column chunks, one Philox call each)
    assert abs(float((m[:, ::8] != 0).double().mean()) - (1 - p)) < 5 * sigma * math.sqrt(8)
    assert abs(float((m[::2] != 0).double().mean()) - (1 - p)) < 5 * sigma * math.sqrt(2)
    # another offset / seed: another mask; the same: the same mask
    assert not torch.equal(m, mask_of(M, N, p, 1234, 2))
    assert not torch.equal(m, mask_of(M, N, p, 1235, 1))
    assert torch.equal(m, mask_of(M, N, p, 1234, 1))
    # limits: p = 0 copies, p = 1 drops everything
    x = rnd(37, 100).to(DEV)
    assert torch.equal(ops.dropout(x, 0.0, 1, 1), x)
    assert torch.count_nonzero(ops.dropout(x, 1.0, 1, 1)) == 0
    # bf16 and odd widths (tail of the last 8-column chunk)
    xb = rnd(5, 13).to(BF).to(DEV)
    yb = ops.dropout(xb, 0.5, 9, 3).float().cpu()
    mb = mask_of(5, 13, 0.5, 9, 3)
    assert torch.equal(yb, (xb.float().cpu() * mb).to(BF).float())


@pytest.mark.parametrize("dtype", [BF, torch.float32])
@pytest.mark.parametrize("M,N,K", [(3, 768, 768), (32, 768, 3072), (300, 768, 768), (2048, 768, 768),
                                   (4096, 768, 3072), (51, 1003, 72)])
def test_epilogue_mask_equals_the_standalone_pass(dtype, M, N, K):
    """vy_linear_dropout_fwd (every kernel family: matrix-vector, skinny, 128-tile, 256 x 192 tile, fp32) drops
    exactly the elements vy_dropout drops for the same (seed, offset)."""
    from vyomai_amd import ops
    p, seed, off = 0.25, 77, 5
    x = rnd(M, K, seed=1).to(dtype).to(DEV)
    w = rnd(N, K, seed=2, scale=1 / math.sqrt(K)).to(dtype).to(DEV)
    b = rnd(N, seed=3, scale=0.1).to(dtype).to(DEV)
    r = rnd(M, N, seed=4).to(dtype).to(DEV)
    y = ops.linear(x, w, b, residual=r, dropout=(p, seed, off))
    pre = ops.linear(x, w, b)                                   # the projection in the storage dtype
    want = (ops.dropout(pre, p, seed, off).float() + r.float()).to(dtype)
    tol = 2e-2 if dtype == BF else 1e-5
    assert torch.allclose(y.float(), want.float(), atol=tol, rtol=tol)
    # dropped elements are exactly the residual
    m = mask_of(M, N, p, seed, off).to(DEV)
    assert torch.equal(y[m == 0], r[m == 0])
    assert 0.70 < float((m != 0).float().mean()) < 0.80 or M * N < 4000


def test_modules_with_dropout_match_the_oracle_given_the_mask():
    """AttentionSelfOutput and FeedForward in train() at p = 0.1 (fp32, 1e-5) against the oracle evaluated with
    the exported masks; eval() is unchanged."""
    from vyomai_amd import rng
    from vyomai_amd.layers.attention import AttentionSelfOutput
    from vyomai_amd.layers.ffn import FeedForward
    cfg = cases.wide_cfg()
    cfg.hidden_dropout_prob = 0.1
    B, L, d = 2, 17, cfg.hidden_size
    c = O.Cfg.of(cfg)
    x, res = rnd(B, L, d, seed=5), rnd(B, L, d, seed=6)
    for cls, name in ((AttentionSelfOutput, "aso"), (FeedForward, "ffn")):
        mod = cls(cfg)
        for n, t in mod.state_dict().items():
            t.copy_(T(recipe.param_value(f"drop.{name}." + n, tuple(t.shape))))
        sd = {k: v.clone() for k, v in mod.state_dict().items()}
        mod = mod.to(DEV).train()
        rng.manual_seed(4242)
        with torch.no_grad():
            y = mod(x.to(DEV), res.to(DEV))
        mask = mask_of(B * L, d, 0.1, 4242, 1).view(B, L, d)     # first draw after manual_seed: offset 1
        if name == "aso":
            want = O.attention_self_output(sd, "", x, res, c.layer_norm_eps, drop=mask)
            plain = O.attention_self_output(sd, "", x, res, c.layer_norm_eps)
        else:
            want = O.feed_forward(sd, "", c, x, res, drop=mask)
            plain = O.feed_forward(sd, "", c, x, res)
        assert torch.allclose(y.cpu(), want, atol=2e-5, rtol=1e-5), (name, (y.cpu() - want).abs().max())
        assert not torch.allclose(y.cpu(), plain, atol=1e-3)
        with torch.no_grad():
            y2 = mod(x.to(DEV), res.to(DEV))                      # next draw: another mask
            assert not torch.allclose(y2, y, atol=1e-3)
            ye = mod.eval()(x.to(DEV), res.to(DEV))
        assert torch.allclose(ye.cpu(), plain, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("at", [None, "gqa"])
def test_layer_gradients_with_dropout_vs_autograd(at):
    """One DecoderLayer in train() at p = 0.1, bf16 kernels: output and every gradient against torch autograd
    through the fp32 CPU oracle with the SAME masks (offsets 1 and 2 after manual_seed)."""
    from vyomai_amd import rng
    from vyomai_amd.layers.mask import AttnMask
    from vyomai_amd.layers.positional_embeddings import RopeSlice, RopeTable
    from vyomai_amd.models.decoder import DecoderLayer
    cfg = cases.wide_cfg()
    cfg.hidden_dropout_prob = 0.1
    B, L, d = 2, 40, cfg.hidden_size
    dh = d // cfg.num_attention_heads
    layer = DecoderLayer(cfg, 0, at)
    for n, t in layer.state_dict().items():
        t.copy_(T(recipe.param_value(f"dropl.{at}." + n, tuple(t.shape))))
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in layer.state_dict().items()}
    layer = layer.to(DEV).train()
    x0 = T(recipe.uniform("dropl.x", (B, L, d)))
    gout = T(recipe.uniform("dropl.gout", (B, L, d)))
    x = x0.to(DEV).to(BF).requires_grad_(True)
    kp = T(cases.keypad(B, L))
    mask = AttnMask.from_padding(kp.to(DEV), causal=True, start_pos=0, query_len=L)
    freqs_tab = O.rotary_angles(dh, cfg.max_position_embeddings)
    rng.manual_seed(99)
    y, _ = layer(x, mask, RopeSlice(RopeTable(freqs_tab), 0, L))
    (y.float() * gout.to(DEV).to(BF).float()).sum().backward()
    m1 = mask_of(B * L, d, 0.1, 99, 1).view(B, L, d)
    m2 = mask_of(B * L, d, 0.1, 99, 2).view(B, L, d)
    # reference: fp32 autograd through the oracle on the bf16-rounded input
    xr = x0.to(BF).float().requires_grad_(True)
    c = O.Cfg.of(cfg)
    add = O.decoder_additive_mask(B, L, kp.float(), 0, torch.float32)
    yr = O.block(sd, "", c, xr, add, freqs_tab[:, :L], at == "gqa", drops=(m1, m2))
    (yr * gout.to(BF).float()).sum().backward()

    def rel(a, b):
        return float((a.detach().float().cpu() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-12))
    assert rel(y, yr) < 3e-2, rel(y, yr)
    assert rel(x.grad, xr.grad) < 5e-2, rel(x.grad, xr.grad)
    for n, p in layer.named_parameters():
        assert rel(p.grad, sd[n].grad) < 6e-2, (n, rel(p.grad, sd[n].grad))


def test_default_config_training_step_runs():
    """EncoderConfig() (hidden_dropout_prob = 0.1) + train(): the reference Examples' loop shape -- accumulate 2
    micro-steps, clip, step -- on a 2-layer decoder; the loss is finite and goes down."""
    import vyomai_amd as V
    from vyomai_amd.training import FlatTrainer
    cfg = V.EncoderConfig(num_hidden_layers=2, vocab_size=1031)
    assert cfg.hidden_dropout_prob == 0.1
    m = V.DecoderModel(cfg, "rope", None)
    recipe.load_recipe_(m)
    m = m.to(DEV).train()
    tr = FlatTrainer(m, lr=2e-3, accumulate_steps=2, max_grad_norm=1.0)
    ids = T(recipe.token_ids("dflt.ids", (4, 64), 3, cfg.vocab_size)).to(DEV)
    losses = []
    for step in range(12):
        losses.append(float(tr.train_step(lambda: m.clm_loss(ids, ids))))
    assert all(math.isfinite(l) for l in losses)
    assert tr.step_count == 6
    assert losses[-1] < losses[0] - 0.05, losses
    assert float(tr.last_grad_norm) > 0


def test_sumsq_and_device_scaled_adamw():
    from vyomai_amd import ops
    for n in (1, 5, 1024, 100_003, 3_000_001):
        g = rnd(n, seed=n)
        got = float(ops.sumsq(g.to(DEV)))
        want = float(g.double().pow(2).sum())
        assert abs(got - want) <= 2e-6 * want + 1e-12, (n, got, want)
    n = 10_007
    p0, g = rnd(n, seed=1), rnd(n, seed=2)
    outs = []
    for host_scale, dev_scale in ((0.37, None), (1.0, 0.37)):
        p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        sd = None if dev_scale is None else torch.tensor([dev_scale], dtype=torch.float32, device=DEV)
        ops.adamw_step(p, g.to(DEV), m, v, None, 1e-2, 0.9, 0.999, 1e-8, 0.01, 1, host_scale, scale_dev=sd)
        outs.append((p.cpu(), m.cpu(), v.cpu()))
    for a, b in zip(*outs):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-9)


def test_trainer_clip_matches_torch_clip_grad_norm():
    import vyomai_amd as V
    from vyomai_amd.training import FlatTrainer
    cfg = cases.test_cfg()
    cfg.num_hidden_layers, cfg.vocab_size, cfg.hidden_dropout_prob = 1, 1031, 0.0
    ids = T(recipe.token_ids("clip.ids", (2, 32), 3, cfg.vocab_size)).to(DEV)
    res = {}
    for name, mx in (("clip", 0.5), ("free", None)):
        m = V.DecoderModel(cfg, "rope", None)
        recipe.load_recipe_(m)
        m = m.to(DEV).train()
        tr = FlatTrainer(m, lr=1e-3, max_grad_norm=mx, overlap_optimizer=False)
        tr.zero_grad()
        tr.backward(m.clm_loss(ids, ids))
        g = tr.arena.grad.clone()
        before = tr.arena.master.clone()
        tr.optimizer_step()
        res[name] = (g, before, tr.arena.master.clone(), tr)
    g, before, after, tr = res["clip"]
    norm = g.double().norm()
    assert norm > 0.5                                              # the clip is active
    assert abs(float(tr.last_grad_norm) - float(norm)) < 1e-4 * float(norm)
    # reference: torch AdamW on the clipped gradient
    coef = min(1.0, 0.5 / (float(norm) + 1e-6))
    p = before.clone().requires_grad_(True)
    opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=0.01)
    p.grad = g * coef
    opt.step()
    touched = g != 0
    assert torch.allclose(after[touched], p.detach()[touched], atol=2e-6, rtol=1e-5)
    assert torch.equal(res["free"][0], g)                           # same gradients, different update
    assert not torch.allclose(res["free"][2], after)


def test_out_of_range_labels_are_ignored_and_flagged():
    from vyomai_amd import ops
    M, V = 6, 1003
    ld = (V + 7) // 8 * 8
    logits = rnd(M, V, seed=3)
    labels = torch.tensor([5, -100, V, -7, 1002, 0])
    good = torch.tensor([True, False, False, False, True, True])
    for fused in (True, False):
        buf = torch.zeros(M, ld, dtype=BF, device=DEV)
        buf[:, :V] = logits.to(BF).to(DEV)
        lse = torch.empty(M, dtype=torch.float32, device=DEV)
        acc = torch.zeros(2, dtype=torch.float32, device=DEV)
        err = torch.zeros(1, dtype=torch.int32, device=DEV)
        gs = torch.ones(1, dtype=torch.float32, device=DEV)
        if fused:
            acc[1] = float(good.sum())
            ops.xent_fused_(buf[:, :V], labels.to(DEV), -100, lse, acc[0:1], acc[1:2], gs, err)
        else:
            ops.xent_fwd(buf[:, :V], labels.to(DEV), -100, lse, acc[0:1], acc[1:2], err)
            ops.xent_bwd_(buf[:, :V], labels.to(DEV), -100, lse, gs, acc[1:2])
        assert int(err.item()) == 1
        lg = logits.to(BF).double()
        want = torch.nn.functional.cross_entropy(lg[good], labels[good], reduction="sum")
        assert abs(float(acc[0]) - float(want)) < 2e-2
        assert float(acc[1]) == 3.0
        assert torch.count_nonzero(buf[~good]) == 0                  # ignored rows: zero gradient
        assert torch.count_nonzero(buf[good]) > 0
    # through the model API
    import vyomai_amd as Vm
    cfg = cases.micro_cfg()
    m = Vm.DecoderModel(cfg, "rope", None).to(DEV).to(BF).train()
    cfg.hidden_dropout_prob = 0.0
    ids = torch.randint(3, cfg.vocab_size, (2, 8), device=DEV)
    bad = ids.clone()
    bad[0, 3] = cfg.vocab_size + 5
    loss = m.clm_loss(ids, bad)
    assert torch.isfinite(loss)
    with pytest.raises(ValueError, match="outside"):
        m.lm_head.raise_on_label_error()
    # the mean must not count the bad row on either path -- the same loss as with that label set to ignore_index
    asign = bad.clone()
    asign[0, 3] = -100
    m.eval()                                   # (the model was built with the default dropout: no random masks here)
    loss = m.clm_loss(ids, bad)
    m.lm_head.label_error.zero_()
    want = m.clm_loss(ids, asign)
    assert abs(float(loss.detach()) - float(want.detach())) < 2e-5, (float(loss.detach()), float(want.detach()))      # fused (bf16, V <= 65536) path
    m32 = Vm.DecoderModel(cfg, "rope", None).to(DEV).eval()
    m32.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
    l32, w32 = m32.clm_loss(ids, bad), m32.clm_loss(ids, asign)                  # two-pass path (fp32)
    assert abs(float(l32.detach()) - float(w32.detach())) < 2e-5
    assert abs(float(l32.detach()) - float(loss.detach())) < 5e-2                                  # the two paths agree on the scale
    m.train()
    m.lm_head.label_error.zero_()
    m.clm_loss(ids, ids)
    m.lm_head.raise_on_label_error()                                # clean labels: no error


@pytest.mark.parametrize("variant", [8, 9])
def test_gemm_tile_variants_agree_with_the_default_selection(variant):
    """The A/B knob that is left (VY_GEMM_VARIANT / vy_debug_set_gemm_variant: 8 / 9 force the 256 x 256 / 256 x 192 tiles of
    the two-stage 32 x 32 x 16 kernel) against the default 16 x 16 x 32 kernels and fp64: same k order per output element,
    so equal up to the rounding of the fp32 chain."""
    import ctypes as C
    from vyomai_amd import ops, _lib
    lib = _lib.load()
    lib.vy_debug_set_gemm_variant.argtypes = [C.c_int]
    try:
        for M, N, K, act in ((2048, 2304, 768, 0), (1280, 768, 3072, 1), (1536, 3072, 768, 1)):
            x = rnd(M, K, seed=1).to(BF).to(DEV)
            w = rnd(N, K, seed=2, scale=1 / math.sqrt(K)).to(BF).to(DEV)
            b = rnd(N, seed=3, scale=0.1).to(BF).to(DEV)
            r = rnd(M, N, seed=4).to(BF).to(DEV)
            lib.vy_debug_set_gemm_variant(-1)
            y0 = ops.linear(x, w, b, act=act, residual=r)
            lib.vy_debug_set_gemm_variant(variant)
            y1 = ops.linear(x, w, b, act=act, residual=r)
            assert (y0.float() - y1.float()).abs().max() <= 2 ** -6 * max(1.0, float(y0.float().abs().max()))
            pre = x.double() @ w.double().t() + b.double()
            want = (O.gelu_erf(pre) if act else pre) + r.double()
            assert torch.allclose(y1.double(), want, atol=3e-2, rtol=1e-2)
    finally:
        lib.vy_debug_set_gemm_variant(-1)
