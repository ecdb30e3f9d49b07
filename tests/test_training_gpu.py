"""Training path on the MI355X: gradients of one decoder layer vs the REAL reference's autograd
(tests/golden/grads.npz), and a few optimizer steps vs the CPU oracle trained with torch AdamW."""
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


def rel_err(got, want):
    got = got.detach().float().cpu().numpy()
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-12))


@pytest.mark.parametrize("at", [None, "gqa"])
def test_layer_gradients_vs_reference(golden, at):
    from vyomai_amd.layers.mask import AttnMask
    from vyomai_amd.layers.positional_embeddings import RopeSlice, RopeTable
    from vyomai_amd.models.decoder import DecoderLayer
    g = golden("grads")
    tag = "wide"
    cfg = cases.wide_cfg()
    cfg.hidden_dropout_prob = 0.0
    B, L = cases.MODULE_BL[tag]
    d = cfg.hidden_size
    dh = d // cfg.num_attention_heads
    layer = DecoderLayer(cfg, 0, at)
    for n, t in layer.state_dict().items():
        t.copy_(T(recipe.param_value(f"{tag}.layer.{at}." + n, tuple(t.shape))))
    layer = layer.to(DEV).train()
    x = T(recipe.uniform(f"{tag}.x", (B, L, d))).to(DEV).to(BF).requires_grad_(True)
    gout = T(recipe.uniform(f"{tag}.gout", (B, L, d))).to(DEV).to(BF)
    mask = AttnMask.from_padding(T(cases.keypad(B, L)).to(DEV), causal=True, start_pos=0, query_len=L)
    freqs = RopeSlice(RopeTable(O.rotary_angles(dh, cfg.max_position_embeddings)), 0, L)
    y, _ = layer(x, mask, freqs)
    (y.float() * gout.float()).sum().backward()
    assert rel_err(y, g[f"{tag}.{at}.y"]) < 3e-2
    assert rel_err(x.grad, g[f"{tag}.{at}.dx"]) < 5e-2, rel_err(x.grad, g[f"{tag}.{at}.dx"])
    for n, p in layer.named_parameters():
        want = g[f"{tag}.{at}.d.{n}"]
        got = p.grad if p.grad.numel() <= 4096 else cases.sub2(p.grad)
        e = rel_err(got, want)
        assert e < 6e-2, (n, e)


def test_trainer_follows_oracle_training():
    """3 AdamW steps of a 2-layer decoder: FlatTrainer (bf16 kernels, fp32 masters, fused AdamW) vs
    the fp32 CPU oracle optimised by torch.optim.AdamW on the same recipe weights and batch."""
    import vyomai_amd as V
    from vyomai_amd.training import FlatTrainer
    cfg = cases.test_cfg()
    cfg.num_hidden_layers, cfg.vocab_size, cfg.hidden_dropout_prob = 2, 1031, 0.0
    m = V.DecoderModel(cfg, "rope", None)
    recipe.load_recipe_(m)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()
          if k != "lm_head.decoder.bias"}
    m = m.to(DEV).train()
    ids = T(recipe.token_ids("train.ids", (4, 48), 3, cfg.vocab_size))
    labels = ids.clone()
    labels[0, 40:] = -100
    tr = FlatTrainer(m, lr=1e-3, weight_decay=0.01)
    opt = torch.optim.AdamW(list(sd.values()), lr=1e-3, weight_decay=0.01)
    c = O.Cfg.of(cfg)
    for step in range(3):
        loss = tr.train_step(lambda: m.clm_loss(ids.to(DEV), labels.to(DEV)))
        full = dict(sd)
        full["lm_head.decoder.bias"] = sd["lm_head.bias"]
        opt.zero_grad()
        ref = O.clm_loss(O.decoder_forward(full, c, ids, None, "rope", None).logits, labels)
        ref.backward()
        opt.step()
        print(f"step {step}: HIP loss {loss.item():.5f}  oracle loss {ref.item():.5f}")
        assert abs(loss.item() - ref.item()) < 3e-2 * max(1.0, abs(ref.item()))
    w = m.all_layer[1].feed_forward.out.weight.detach().float().cpu()
    wr = sd["all_layer.1.feed_forward.out.weight"].detach()
    # Adam moves every element by ~lr per step whatever the gradient magnitude, so an element whose
    # tiny gradient changes sign under bf16 noise differs by up to 2*lr*steps; the bulk must agree
    assert (w - wr).abs().mean() < 2e-4 and (w - wr).abs().max() <= 2 * 3 * 1e-3 + 1e-4
    assert m.all_layer[0].attention.query.weight.grad.data_ptr() >= tr.arena.grad.data_ptr()


def test_xent_kernel():
    from vyomai_amd import ops
    M, V = 300, 50265
    ld = (V + 7) // 8 * 8
    g = torch.Generator().manual_seed(0)
    lg = (torch.randn(M, V, generator=g) * 2).to(BF)
    labels = torch.randint(0, V, (M,), generator=g)
    labels[::7] = -100
    buf = torch.zeros(M, ld, dtype=BF, device=DEV)
    buf[:, :V] = lg.to(DEV)
    lse = torch.empty(M, device=DEV)
    acc = torch.zeros(2, device=DEV)
    ops.xent_fwd(buf[:, :V], labels.to(DEV), -100, lse, acc[0:1], acc[1:2])
    x = lg.double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(x, labels, ignore_index=-100)
    ref.backward()
    assert abs((acc[0] / acc[1]).item() - ref.item()) < 1e-3
    ops.xent_bwd_(buf[:, :V], labels.to(DEV), -100, lse, torch.ones(1, device=DEV), acc[1:2])
    got = buf[:, :V].float().cpu()
    assert (got - x.grad.float()).abs().max() < 2e-4
    assert float(buf[:, V:].abs().max()) == 0.0


def test_xent_fused_kernel():
    """One-pass loss + in-place gradient == vy_xent_fwd followed by vy_xent_bwd (and torch fp64)."""
    from vyomai_amd import ops
    M, V = 300, 50265
    ld = (V + 7) // 8 * 8
    g = torch.Generator().manual_seed(1)
    lg = (torch.randn(M, V, generator=g) * 2).to(BF)
    labels = torch.randint(0, V, (M,), generator=g)
    labels[::5] = -100
    buf = torch.full((M, ld), 3.0, dtype=BF, device=DEV)   # garbage in the pad columns: must come out 0
    buf[:, :V] = lg.to(DEV)
    lse = torch.empty(M, device=DEV)
    acc = torch.zeros(2, device=DEV)
    acc[1] = float((labels != -100).sum())
    gs = torch.full((1,), 0.5, device=DEV)
    ops.xent_fused_(buf[:, :V], labels.to(DEV), -100, lse, acc[0:1], acc[1:2], gs)
    x = lg.double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(x, labels, ignore_index=-100)
    (0.5 * ref).backward()
    assert abs((acc[0] / acc[1]).item() - ref.item()) < 1e-3
    got = buf[:, :V].float().cpu()
    assert (got - x.grad.float()).abs().max() < 2e-4
    assert float(buf[:, V:].abs().max()) == 0.0
    ref_lse = torch.logsumexp(lg.double(), dim=1)
    keep = labels != -100
    assert (lse.cpu().double()[keep] - ref_lse[keep]).abs().max() < 1e-3


def test_lm_head_loss_upstream_gradient_scale():
    """(c * loss).backward() == c * grads of loss.backward(): the unit gradient stored by the fused
    kernel is scaled through the GEMMs by the device scalar, with no host sync."""
    import vyomai_amd as V
    from vyomai_amd.training import FlatTrainer
    grads = []
    for c in (1.0, 0.25):
        cfg = cases.test_cfg()
        cfg.num_hidden_layers, cfg.vocab_size, cfg.hidden_dropout_prob = 2, 1031, 0.0
        m = V.DecoderModel(cfg, "rope", None)
        recipe.load_recipe_(m)
        m = m.to(DEV).train()
        ids = T(recipe.token_ids("train.ids", (4, 48), 3, cfg.vocab_size)).to(DEV)
        tr = FlatTrainer(m, lr=1e-3)
        tr.zero_grad()
        loss = m.clm_loss(ids, ids)
        (loss * c).backward()
        torch.cuda.synchronize()
        grads.append(tr.arena.grad.clone())
    err = (grads[1] - 0.25 * grads[0]).abs().max() / grads[0].abs().max()
    assert err < 2e-2, err


def test_embedding_kernels():
    """vy_embedding_fwd/bwd == F.embedding and its autograd (padding row gets no gradient)."""
    from vyomai_amd import ops
    g = torch.Generator().manual_seed(3)
    V, d, M = 1031, 768, 500
    for dt in (torch.float32, BF):
        table = torch.randn(V, d, generator=g).to(dt).to(DEV)
        ids = torch.randint(0, V, (5, 100), generator=g).to(DEV)
        ids[0, :7] = 3
        out = ops.embedding(table, ids)
        assert torch.equal(out, torch.nn.functional.embedding(ids, table))
        dout = torch.randn(5, 100, d, generator=g).to(dt).to(DEV)
        dw = torch.zeros(V, d, device=DEV)
        ops.embedding_bwd_(dout, ids, dw, padding_idx=3)
        ref = torch.zeros(V, d, device=DEV)
        keep = (ids != 3).reshape(-1)
        ref.index_add_(0, ids.reshape(-1)[keep], dout.reshape(-1, d).float()[keep])
        assert (dw - ref).abs().max() < 1e-4
        assert float(dw[3].abs().max()) == 0.0
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    bad = ids.clone(); bad[1, 1] = V + 5
    out = ops.embedding(table, bad, err_flag=err)
    assert int(err.item()) == 1 and float(out[1, 1].abs().max()) == 0.0


def test_transpose_batched():
    """vy_transpose_batched: several matrices (odd sizes, padded row strides) in one launch."""
    from vyomai_amd import ops
    g = torch.Generator().manual_seed(5)
    for dt in (BF, torch.float32):
        pairs = []
        for R, C in ((768, 768), (2304, 768), (1031, 768), (70, 200), (3, 5)):
            src = torch.randn(R, C, generator=g).to(dt).to(DEV)
            ld = (R + 7) // 8 * 8
            dst = torch.full((C, ld), 7.0, dtype=dt, device=DEV)[:, :R]
            pairs.append((src, dst))
        ops.TransposeBatch(pairs).run()
        for src, dst in pairs:
            assert torch.equal(dst, src.t()), (src.shape, dt)


def test_unfused_lm_head_backward_matches_fused_loss():
    """Training through model(...).logits + torch cross-entropy (LMHeadFn: the gradient arrives
    unpadded, vocabulary 1031 is odd) gives the gradients of the fused clm_loss path."""
    import vyomai_amd as V
    from vyomai_amd.training import FlatTrainer
    grads = []
    for fused in (True, False):
        cfg = cases.test_cfg()
        cfg.num_hidden_layers, cfg.vocab_size, cfg.hidden_dropout_prob = 2, 1031, 0.0
        m = V.DecoderModel(cfg, "rope", None)
        recipe.load_recipe_(m)
        m = m.to(DEV).train()
        ids = T(recipe.token_ids("train.ids", (4, 48), 3, cfg.vocab_size)).to(DEV)
        tr = FlatTrainer(m, lr=1e-3)
        tr.zero_grad()
        if fused:
            loss = m.clm_loss(ids, ids)
        else:
            logits = m(ids).logits
            loss = torch.nn.functional.cross_entropy(logits[:, :-1].float().reshape(-1, cfg.vocab_size),
                                                     ids[:, 1:].reshape(-1))
        loss.backward()
        torch.cuda.synchronize()
        grads.append((loss.item(), tr.arena.grad.clone()))
    assert abs(grads[0][0] - grads[1][0]) < 2e-2
    err = (grads[0][1] - grads[1][1]).abs().max() / grads[0][1].abs().max()
    assert err < 3e-2, err


def test_grouped_weight_gradients_match_ungrouped():
    """At training sizes (>= 4096 rows) the weight gradients of a layer are deferred and launched together
    (autograd_train._WgradGroup): same gradients and the same training trajectory as one launch per projection;
    no parameter is reported ready (DDP bucket / per-bucket AdamW) while its gradient launch is still pending --
    autograd's own post-accumulate hook fires earlier than that and must be ignored for deferred weights."""
    import vyomai_amd as V
    from vyomai_amd import autograd_train as AT
    from vyomai_amd.training import FlatTrainer
    cfg = cases.with_kv(cases.test_cfg(), None)
    cfg.num_hidden_layers, cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size = 3, 512, 8, 2048
    cfg.hidden_dropout_prob, cfg.vocab_size = 0.0, 1000
    ids = T(recipe.token_ids("grp.ids", (8, 512), 3, cfg.vocab_size)).to(DEV)
    grads, losses, finals = {}, {}, {}
    for grouped in (False, True):
        m = V.DecoderModel(cfg, "rope", None)
        recipe.load_recipe_(m)
        m = m.to(DEV).train()
        tr = FlatTrainer(m, lr=1e-3, bucket_bytes=4 << 20)     # several buckets: some are final mid-backward
        early, launches = [], []
        real_ready = tr.reducer.mark_ready

        def spy(p):
            if any(p is q for *_, members in AT._wgrad_group.items for q in members):
                early.append(p)
            real_ready(p)

        tr.reducer.mark_ready = spy
        for p in tr.arena.params:
            p._vy_ready = spy
        old = AT._GROUP_WGRADS
        AT._GROUP_WGRADS = grouped
        real = AT.ops.linear_wgrad_grouped
        AT.ops.linear_wgrad_grouped = lambda items: (launches.append(len(items)), real(items))[1]
        try:
            tr.zero_grad()
            loss = m.clm_loss(ids, ids)
            tr.backward(loss)
            assert not AT._wgrad_group.items and not AT._wgrad_group.armed
            torch.cuda.synchronize()
            grads[grouped] = tr.arena.grad.clone()
            losses[grouped] = [float(tr.train_step(lambda: m.clm_loss(ids, ids))) for _ in range(4)]
            torch.cuda.synchronize()
            finals[grouped] = tr.arena.master.clone()
        finally:
            AT._GROUP_WGRADS = old
            AT.ops.linear_wgrad_grouped = real
        assert (len(launches) > 0) == grouped, launches
        assert not early, f"{len(early)} parameters were reported ready before their gradient launch"
    a, b = grads[False], grads[True]
    assert torch.isfinite(b).all()
    scale = a.abs().max().item()
    assert (a - b).abs().max().item() <= 2e-3 * scale, ((a - b).abs().max().item(), scale)
    assert losses[True][-1] < losses[True][0]
    assert max(abs(x - y) for x, y in zip(losses[False], losses[True])) < 2e-2, (losses[False], losses[True])
    assert (finals[False] - finals[True]).abs().max().item() < 5e-3


def test_vlm_caption_loss_matches_cross_entropy_of_the_logits():
    """VisionLanguageModel.caption_loss (fused head + cross-entropy) against the captioning notebooks' loss on
    the materialised logits: cross_entropy(logits[:, 1:-1], ids[:, 1:]); with a padding mask the padded targets
    drop out."""
    import vyomai_amd as V
    from vyomai_amd.training import FlatTrainer
    vcfg = cases.vit_cfg()
    vcfg.num_hidden_layers = 1
    if hasattr(vcfg, "hidden_dropout_prob"):
        vcfg.hidden_dropout_prob = 0.0
    cfg = cases.with_kv(cases.test_cfg(), None)
    cfg.num_hidden_layers, cfg.hidden_dropout_prob, cfg.vocab_size = 2, 0.0, 3000
    vlm = V.VisionLanguageModel(cfg, V.Vit(vcfg), "rope", None)
    recipe.load_recipe_(vlm)
    vlm = vlm.to(DEV).train()
    FlatTrainer(vlm, lr=1e-4)      # bf16 compute dtype + shadows
    img = T(recipe.uniform("cap.img", (3, 3, 224, 224), 0.5, 0.5)).to(DEV).to(BF)
    ids = T(recipe.token_ids("cap.ids2", (3, 12), 3, cfg.vocab_size)).to(DEV)
    with torch.no_grad():
        logits = vlm(pixel_values=img, decoder_input_ids=ids).logits.float()
    want = torch.nn.functional.cross_entropy(logits[:, 1:-1].reshape(-1, logits.shape[-1]), ids[:, 1:].reshape(-1))
    got = vlm.caption_loss(img, ids)
    assert abs(float(got) - float(want)) < 2e-2 * max(1.0, float(want)), (float(got), float(want))
    am = torch.ones_like(ids)
    am[1, 8:] = 0
    with torch.no_grad():
        lg = vlm(pixel_values=img, decoder_input_ids=ids, decoder_attention_mask=am).logits.float()
    tgt = ids[:, 1:].clone()
    tgt[am[:, 1:] == 0] = -100
    want_m = torch.nn.functional.cross_entropy(lg[:, 1:-1].reshape(-1, lg.shape[-1]), tgt.reshape(-1), ignore_index=-100)
    got_m = vlm.caption_loss(img, ids, am)
    assert abs(float(got_m) - float(want_m)) < 2e-2 * max(1.0, float(want_m)), (float(got_m), float(want_m))
    got_m.backward()


# ---- configs[3] training: ViT and vision-language gradients vs the REAL reference's autograd -------------

def _grad_sample(g):
    g = g.detach().float().cpu()
    if g.numel() <= 4096:
        return g.numpy()
    g2 = g.reshape(g.shape[0], -1) if g.dim() != 3 else g.reshape(-1, g.shape[-1])
    return cases.sub2(g2).numpy()


def _rel(got, want, floor=1e-3):
    """max |got - want| over the gradient's scale.  `floor`: a gradient that is analytically zero (the key bias
    without RoPE: softmax is invariant to a constant added to every key) is rounding noise in both
    implementations -- 1e-9 in the fp32 reference, 1e-5 under bf16 kernels -- and is checked absolutely."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), floor))


@pytest.mark.parametrize("dt", [BF, torch.float32])
def test_vit_gradients_vs_reference(golden, dt):
    """(bf16 at bf16 tolerance; fp32 -- forward and backward on the plain-FMA kernels -- at 2e-4.)  One-layer Vit, B = 2: patchify weight / bias, cls_token, position table (the in-place double add gives
    both a factor 2), the fused qkv projection at L = 197 non-causal, FeedForward -- against the reference's
    autograd (tests/golden/grads_vision.npz)."""
    import vyomai_amd as V
    g = golden("grads_vision")
    vcfg = cases.vit_cfg()
    vcfg.num_hidden_layers, vcfg.hidden_dropout_prob = 1, 0.0
    vit = V.Vit(vcfg)
    for n, t in vit.state_dict().items():
        t.copy_(T(recipe.param_value("vgrad.vit." + n, tuple(t.shape))))
    vit = vit.to(DEV).train()
    img = T(recipe.uniform("vgrad.img", (2, 3, 224, 224), 0.5, 0.5)).to(DEV).to(dt)
    y = vit(img).logits
    assert y.dtype == dt
    gout = T(recipe.uniform("vgrad.gout", tuple(y.shape))).to(DEV)
    (y.float() * gout).sum().backward()
    ty, tg = (3e-2, 6e-2) if dt == BF else (3e-5, 2e-4)
    assert _rel(cases.sub(y.detach().float().cpu()).numpy(), g["vit.y"]) < ty
    for n, p in vit.named_parameters():
        assert p.grad is not None, f"{n} received no gradient"
        e = _rel(_grad_sample(p.grad), g["vit.d." + n])
        assert e < tg, (n, e)


@pytest.mark.parametrize("dt", [BF, torch.float32])
@pytest.mark.parametrize("pos,at", [("rope", None), ("absolute", "gqa")])
def test_vlm_caption_gradients_vs_reference(golden, pos, at, dt):
    """VisionLanguageModel (1-layer Vit + 1-layer decoder), caption loss with a padded row: the loss and every
    gradient -- the encoder's included, which reach it only through the prepended image token -- against the
    reference's autograd under cross_entropy(logits[:, 1:-1], ids[:, 1:])."""
    import vyomai_amd as V
    g = golden("grads_vision")
    vcfg = cases.vit_cfg()
    vcfg.num_hidden_layers, vcfg.hidden_dropout_prob = 1, 0.0
    c = cases.with_kv(cases.test_cfg(), at)
    c.num_hidden_layers, c.vocab_size, c.hidden_dropout_prob = 1, 1031, 0.0
    vlm = V.VisionLanguageModel(c, V.Vit(vcfg), pos, at)
    for n, t in vlm.state_dict().items():
        if t.is_floating_point():
            t.copy_(T(recipe.param_value(f"vgrad.vlm.{pos}.{at}." + n, tuple(t.shape))))
    vlm = vlm.to(DEV).train()
    for m in vlm.modules():
        if hasattr(m, "compute_dtype"):
            m.compute_dtype = dt
    img = T(recipe.uniform("vgrad.img", (2, 3, 224, 224), 0.5, 0.5)).to(DEV).to(dt)
    ids = T(recipe.token_ids("vgrad.ids", (2, 12), 3, c.vocab_size)).to(DEV)
    am = torch.ones(2, 12, dtype=torch.long, device=DEV)
    am[1, 9:] = 0
    loss = vlm.caption_loss(img, ids, am)
    loss.backward()
    want = float(g[f"vlm.{pos}.{at}.loss"][0])
    tl, tg = (2e-2, 8e-2) if dt == BF else (1e-5, 3e-4)
    assert abs(float(loss) - want) < tl * max(1.0, want), (float(loss), want)
    seen = 0
    for n, p in vlm.named_parameters():
        key = f"vlm.{pos}.{at}.d." + n
        if key not in g:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        assert p.grad is not None, f"{n} received no gradient"
        e = _rel(_grad_sample(p.grad), g[key])
        assert e < tg, (n, e)
        seen += 1
    assert seen > 20 and any(n.startswith("encoder.") for n, _ in vlm.named_parameters())


def test_decoder_layer_at_the_benchmark_size_vs_oracle():
    """One DecoderLayer at the size bench.py runs (B = 32, L = 512, d = 768, 12 heads, RoPE, causal), bf16 forward
    AND backward, against fp32 autograd through the CPU oracle on the same recipe weights: the only kernels-vs-
    oracle evidence at M = 16384 rows (the GEMM tiles, the flash attention grids and the grouped weight
    gradients all take their large-shape paths here)."""
    from vyomai_amd.layers.mask import AttnMask
    from vyomai_amd.layers.positional_embeddings import RopeSlice, RopeTable
    from vyomai_amd.models.decoder import DecoderLayer
    import vyomai_amd as V
    cfg = V.EncoderConfig(num_hidden_layers=1, max_position_embeddings=1024, hidden_dropout_prob=0.0)
    B, L, d = 32, 512, cfg.hidden_size
    dh = d // cfg.num_attention_heads
    layer = DecoderLayer(cfg, 0, None)
    for n, t in layer.state_dict().items():
        t.copy_(T(recipe.param_value("big.layer." + n, tuple(t.shape))))
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in layer.state_dict().items()}
    layer = layer.to(DEV).train()
    x0 = T(recipe.uniform("big.x", (B, L, d))).to(BF)
    g0 = T(recipe.uniform("big.gout", (B, L, d))).to(BF)
    x = x0.to(DEV).requires_grad_(True)
    tab = O.rotary_angles(dh, cfg.max_position_embeddings)
    mask = AttnMask.from_padding(None, causal=True, start_pos=0, query_len=L)
    y, _ = layer(x, mask, RopeSlice(RopeTable(tab), 0, L))
    (y.float() * g0.to(DEV).float()).sum().backward()
    torch.cuda.synchronize()
    # oracle: fp32, all host cores
    xr = x0.float().requires_grad_(True)
    c = O.Cfg.of(cfg)
    add = O.decoder_additive_mask(B, L, None, 0, torch.float32)
    yr = O.block(sd, "", c, xr, add, tab[:, :L], False)
    (yr * g0.float()).sum().backward()

    def rel(a, b):
        return float((a.detach().float().cpu() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-12))

    def rel_rms(a, b):
        a, b = a.detach().float().cpu(), b.detach()
        return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12))
    assert rel(y, yr) < 3e-2, rel(y, yr)
    assert rel_rms(y, yr) < 6e-3, rel_rms(y, yr)
    assert rel(x.grad, xr.grad) < 6e-2, rel(x.grad, xr.grad)
    assert rel_rms(x.grad, xr.grad) < 1.5e-2, rel_rms(x.grad, xr.grad)
    for n, p in layer.named_parameters():
        e = rel(p.grad, sd[n].grad)
        assert e < 6e-2, (n, e)


@pytest.mark.parametrize("d,heads,kv,at", [(512, 4, 2, "gqa"), (288, 4, 4, None), (512, 2, 1, "gqa"), (768, 6, 6, None)])
def test_decoder_layer_other_head_widths_vs_oracle(d, heads, kv, at):
    """DecoderLayer forward AND backward with heads of 128 / 72 / 256 / 128 (the general attention kernels of
    vy_attn.hip / vy_bwd.hip, rotary inverse of dq / dk after them) under a causal + key-padding mask, against fp32
    autograd through the CPU oracle on the same recipe weights (reference layers/attention.py:75-147, 150-250)."""
    from vyomai_amd.layers.mask import AttnMask
    from vyomai_amd.layers.positional_embeddings import RopeSlice, RopeTable
    from vyomai_amd.models.decoder import DecoderLayer
    import vyomai_amd as V
    cfg = V.EncoderConfig(hidden_size=d, num_attention_heads=heads, num_hidden_layers=1, max_position_embeddings=256,
                          hidden_dropout_prob=0.0)
    cfg.num_key_value_heads = kv
    B, L = 3, 150
    dh = d // heads
    layer = DecoderLayer(cfg, 0, at)
    for n, t in layer.state_dict().items():
        t.copy_(T(recipe.param_value(f"wideheads.{d}.{heads}." + n, tuple(t.shape))))
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in layer.state_dict().items()}
    layer = layer.to(DEV).train()
    x0 = T(recipe.uniform("wideheads.x", (B, L, d))).to(BF)
    g0 = T(recipe.uniform("wideheads.gout", (B, L, d))).to(BF)
    x = x0.to(DEV).requires_grad_(True)
    tab = O.rotary_angles(dh, cfg.max_position_embeddings)
    kp = torch.ones(B, L, dtype=torch.int64)
    kp[0, L - 31:] = 0
    kp[2, L - 5:] = 0
    mask = AttnMask.from_padding(kp.to(DEV), causal=True, start_pos=0, query_len=L)
    y, _ = layer(x, mask, RopeSlice(RopeTable(tab), 0, L))
    (y.float() * g0.to(DEV).float()).sum().backward()
    torch.cuda.synchronize()
    xr = x0.float().requires_grad_(True)
    c = O.Cfg.of(cfg)
    add = O.decoder_additive_mask(B, L, kp, 0, torch.float32)
    yr = O.block(sd, "", c, xr, add, tab[:, :L], at == "gqa")
    (yr * g0.float()).sum().backward()

    def rel(a, b):
        return float((a.detach().float().cpu() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-12))
    assert rel(y, yr) < 3e-2, rel(y, yr)
    assert rel(x.grad, xr.grad) < 6e-2, rel(x.grad, xr.grad)
    for n, p in layer.named_parameters():
        e = rel(p.grad, sd[n].grad)
        assert e < 6e-2, (n, e)


@pytest.mark.parametrize("at", [None, "gqa"])
def test_layer_gradients_fp32_vs_reference(golden, at):
    """The same decoder layer as test_layer_gradients_vs_reference, run in fp32 end to end (forward AND backward on the
    plain-FMA kernels): output and every gradient against the REAL reference's autograd (tests/golden/grads.npz) at
    fp32 tolerance -- the backward arithmetic itself is pinned, not only its bf16 rendering."""
    from vyomai_amd.layers.mask import AttnMask
    from vyomai_amd.layers.positional_embeddings import RopeSlice, RopeTable
    from vyomai_amd.models.decoder import DecoderLayer
    g = golden("grads")
    tag = "wide"
    cfg = cases.wide_cfg()
    cfg.hidden_dropout_prob = 0.0
    B, L = cases.MODULE_BL[tag]
    d = cfg.hidden_size
    dh = d // cfg.num_attention_heads
    layer = DecoderLayer(cfg, 0, at)
    for n, t in layer.state_dict().items():
        t.copy_(T(recipe.param_value(f"{tag}.layer.{at}." + n, tuple(t.shape))))
    layer = layer.to(DEV).train()
    x = T(recipe.uniform(f"{tag}.x", (B, L, d))).to(DEV).requires_grad_(True)
    gout = T(recipe.uniform(f"{tag}.gout", (B, L, d))).to(DEV)
    mask = AttnMask.from_padding(T(cases.keypad(B, L)).to(DEV), causal=True, start_pos=0, query_len=L)
    freqs = RopeSlice(RopeTable(O.rotary_angles(dh, cfg.max_position_embeddings)), 0, L)
    y, _ = layer(x, mask, freqs)
    assert y.dtype == torch.float32
    (y * gout).sum().backward()
    assert rel_err(y, g[f"{tag}.{at}.y"]) < 2e-5, rel_err(y, g[f"{tag}.{at}.y"])
    assert rel_err(x.grad, g[f"{tag}.{at}.dx"]) < 1e-4, rel_err(x.grad, g[f"{tag}.{at}.dx"])
    for n, p in layer.named_parameters():
        want = g[f"{tag}.{at}.d.{n}"]
        got = p.grad if p.grad.numel() <= 4096 else cases.sub2(p.grad)
        e = rel_err(got, want)
        assert e < 1e-4, (n, e)


def test_trainer_fp32_follows_oracle_training():
    """FlatTrainer(compute_dtype=torch.float32): 3 AdamW steps of a 2-layer decoder on the fp32 kernels (forward,
    backward, fused AdamW on the masters, no shadow arena) against the fp32 CPU oracle under torch.optim.AdamW -- loss
    and updated weights at fp32 tolerance (Examples/vyom-ai-decoder_clm.ipynb cells 29-31 in full precision)."""
    import vyomai_amd as V
    from vyomai_amd.training import FlatTrainer
    cfg = cases.test_cfg()
    cfg.num_hidden_layers, cfg.vocab_size, cfg.hidden_dropout_prob = 2, 1031, 0.0
    m = V.DecoderModel(cfg, "rope", None)
    recipe.load_recipe_(m)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()
          if k != "lm_head.decoder.bias"}
    m = m.to(DEV).train()
    ids = T(recipe.token_ids("train.ids", (4, 48), 3, cfg.vocab_size))
    labels = ids.clone()
    labels[0, 40:] = -100
    # a bf16 trainer alive in the same process (the W^T registry of the dgrad GEMMs is shared: one refresh launch per dtype)
    m16 = V.DecoderModel(cfg, "rope", None)
    recipe.load_recipe_(m16)
    m16 = m16.to(DEV).train()
    tr16 = FlatTrainer(m16, lr=1e-3, weight_decay=0.01)
    tr16.train_step(lambda: m16.clm_loss(ids.to(DEV), labels.to(DEV)))
    tr = FlatTrainer(m, lr=1e-3, weight_decay=0.01, compute_dtype=torch.float32)
    assert tr.arena.shadow is None
    opt = torch.optim.AdamW(list(sd.values()), lr=1e-3, weight_decay=0.01)
    c = O.Cfg.of(cfg)
    for step in range(3):
        loss = tr.train_step(lambda: m.clm_loss(ids.to(DEV), labels.to(DEV)))
        full = dict(sd)
        full["lm_head.decoder.bias"] = sd["lm_head.bias"]
        opt.zero_grad()
        ref = O.clm_loss(O.decoder_forward(full, c, ids, None, "rope", None).logits, labels)
        ref.backward()
        opt.step()
        tr16.train_step(lambda: m16.clm_loss(ids.to(DEV), labels.to(DEV)))
        print(f"step {step}: HIP fp32 loss {loss.item():.7f}  oracle loss {ref.item():.7f}")
        assert abs(loss.item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item())), (step, loss.item(), ref.item())
    for name in ("all_layer.1.feed_forward.out.weight", "all_layer.0.attention.query.weight", "lm_head.dense.weight",
                 "word_embeddings.weight"):
        if name not in sd:
            continue
        w = dict(m.named_parameters())[name].detach().float().cpu()
        wr = sd[name].detach()
        # Adam normalises every element's step to ~lr: where the gradient is at the fp32 noise floor the direction
        # itself is noise, so the bulk is compared tightly and the worst element loosely
        assert (w - wr).abs().mean() < 2e-6, (name, (w - wr).abs().mean())
        assert (w - wr).abs().max() < 2 * 3 * 1e-3 + 1e-5, (name, (w - wr).abs().max())
