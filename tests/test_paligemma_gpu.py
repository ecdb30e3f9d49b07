"""PaliGemma-shape path (BASELINE.json configs[4]) on the MI355X.

The layer blocks are pinned to the reference notebook's own classes (exec'd cells,
tests/golden/paligemma_blocks.npz); the cached decode loop around them has no importable
reference ("parity unpinned" for that part) and is checked against the oracle's restatement."""
import types

import numpy as np
import pytest
import torch

from oracle import vyom_oracle as O
from tests.golden import cases
from vyomai_amd import recipe

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def fill(mod, prefix):
    for n, t in mod.state_dict().items():
        if t.is_floating_point():
            t.copy_(T(recipe.param_value(prefix + n, tuple(t.shape))))
    return mod.to(DEV).eval()


def close(got, want, atol, what):
    got = got.detach().float().cpu().numpy()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want).max()
    assert np.isfinite(got).all() and err <= atol, f"{what}: max abs err {err:.3e} > {atol}"


def test_blocks_fp32_vs_notebook():
    from vyomai_amd.models import paligemma as P
    from vyomai_amd.layers.positional_embeddings import RopeTable
    g = dict(np.load("tests/golden/paligemma_blocks.npz"))
    scfg = P.SiglipVisionConfig(**cases.SIGLIP)
    layer = fill(P.SiglipEncoderLayer(scfg), "pg.siglip.")
    x = T(recipe.uniform("pg.siglip.x", (2, 20, scfg.hidden_size))).to(DEV)
    with torch.no_grad():
        y = layer(x)
    close(cases.sub2(y.reshape(-1, y.shape[-1])), g["siglip.layer"], 2e-5, "siglip layer (dh=72)")
    gcfg = types.SimpleNamespace(**cases.GEMMA)
    glayer = fill(P.GemmaDecoderLayer(gcfg, 0), "pg.gemma.")
    xg = T(recipe.uniform("pg.gemma.x", (2, 12, gcfg.hidden_size))).to(DEV)
    rope = RopeTable(P._angles(gcfg.head_dim, 64, gcfg.rope_theta))
    with torch.no_grad():
        for key, causal, pos0 in (("nomask", False, 0), ("causal", True, 0)):
            y = glayer(xg, rope, pos0, causal=causal)
            want = g[f"gemma.layer.{key}"]
            close(cases.sub2(y.reshape(-1, y.shape[-1])), want, 3e-5 * max(1.0, float(np.abs(want).max())),
                  f"gemma layer {key} (MQA, dh=256)")
        norm = fill(P.GemmaRMSNorm(gcfg.hidden_size, gcfg.rms_norm_eps), "pg.norm.")
        y = norm(xg)
        close(cases.sub2(y.reshape(-1, y.shape[-1])), g["gemma.rmsnorm"], 2e-6, "rmsnorm")


def test_gemma_cached_decode_matches_full_recompute():
    """KV-cache decode of the Gemma block stack == recomputing the whole sequence (fp32, token-exact
    hidden states to 1e-5), at reduced depth; MQA with head_dim 256."""
    from vyomai_amd.models import paligemma as P
    from vyomai_amd.layers.positional_embeddings import RopeTable
    small = dict(cases.GEMMA, num_hidden_layers=2, intermediate_size=1024, vocab_size=512)
    gcfg = types.SimpleNamespace(**small)
    layers = [fill(P.GemmaDecoderLayer(gcfg, i), f"pgs.{i}.") for i in range(2)]
    rope = RopeTable(P._angles(gcfg.head_dim, 64, gcfg.rope_theta))
    B, Lp, steps = 2, 9, 4
    xs = T(recipe.uniform("pgs.x", (B, Lp + steps, gcfg.hidden_size))).to(DEV)
    caches = [(torch.zeros(B, 1, 32, gcfg.head_dim, device=DEV), torch.zeros(B, 1, 32, gcfg.head_dim, device=DEV))
              for _ in layers]
    with torch.no_grad():
        h = xs[:, :Lp]
        for l, c in zip(layers, caches):
            h = l(h, rope, 0, causal=True, cache=c)
        outs = [h[:, -1:]]
        for s in range(steps):
            h = xs[:, Lp + s:Lp + s + 1]
            for l, c in zip(layers, caches):
                h = l(h, rope, Lp + s, cache=c)
            outs.append(h)
        full = xs
        for l in layers:
            full = l(full, rope, 0, causal=True)
    got = torch.cat(outs, dim=1)
    want = full[:, Lp - 1:]
    assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())
    # and against the CPU oracle restatement of the same blocks
    sd = {}
    for i in range(2):
        shapes = {k: v for k, v in cases.gemma_layer_shapes(f"L{i}.").items()}
        for n, shp in shapes.items():
            shp = tuple(1024 if d_ == 16384 else d_ for d_ in shp)
            sd[n] = T(recipe.param_value(f"pgs.{i}." + n[len(f"L{i}."):], shp))
    ref = xs.cpu()
    mask = T(cases.causal_additive(B, Lp + steps, 0, None))
    for i in range(2):
        ref = O.gemma_layer(sd, f"L{i}.", ref, 8, 1, 256, 1e-6, mask, 0)
    assert (full.cpu() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())


def test_paligemma_shape_generate_bf16_runs():
    """Reduced-depth PaliGemma-shaped model end to end in bf16: image -> 256 tokens -> prefix -> greedy
    decode; finite logits, right shapes, cache positions advance."""
    from vyomai_amd.models import paligemma as P
    vis = P.SiglipVisionConfig(**dict(cases.SIGLIP, num_hidden_layers=2))
    txt = types.SimpleNamespace(**dict(cases.GEMMA, num_hidden_layers=2, vocab_size=4096))
    m = P.PaliGemmaForConditionalGeneration(P.PaliGemmaShape(vis, txt, txt.hidden_size))
    for n, p in m.named_parameters():
        with torch.no_grad():
            p.copy_(T(recipe.param_value("pgm." + n, tuple(p.shape))))
    m = m.to(DEV).to(torch.bfloat16).eval()
    img = T(recipe.uniform("pgm.img", (1, 3, 224, 224), 0.5, 0.5)).to(DEV)
    ids = T(recipe.token_ids("pgm.ids", (1, 8), 3, 4096)).to(DEV)
    out = m.generate(img, ids, max_new_tokens=6, max_cache_len=384)
    assert out.shape == (1, 6) and int(out.min()) >= 0 and int(out.max()) < 4096


def test_native_gemma_step_matches_the_layer_by_layer_loop():
    """generate() drives the language model with one vy_gemma_decoder_step call per token; the same tokens and the
    same last-step logits must come out of the Python loop over GemmaDecoderLayer.forward (bf16, M = 1: both use the
    same kernels in the same order)."""
    from vyomai_amd.models import paligemma as P
    from vyomai_amd import ops
    vis = P.SiglipVisionConfig(**dict(cases.SIGLIP, num_hidden_layers=1))
    txt = types.SimpleNamespace(**dict(cases.GEMMA, num_hidden_layers=3, vocab_size=5000, intermediate_size=2048))
    m = P.PaliGemmaForConditionalGeneration(P.PaliGemmaShape(vis, txt, txt.hidden_size))
    for n, p in m.named_parameters():
        with torch.no_grad():
            p.copy_(T(recipe.param_value("pgn." + n, tuple(p.shape))))
    m = m.to(DEV).to(torch.bfloat16).eval()
    img = T(recipe.uniform("pgn.img", (2, 3, 224, 224), 0.5, 0.5)).to(DEV)
    ids = T(recipe.token_ids("pgn.ids", (2, 5), 3, 5000)).to(DEV)
    steps = 7
    got = m.generate(img, ids, max_new_tokens=steps, max_cache_len=300)
    # the loop the notebook runs (cell 30), layer by layer through the Python modules
    with torch.no_grad():
        dt = torch.bfloat16
        emb = ops.linear(m.vision_tower(img.to(dt)), m.multi_modal_projector.weight, m.multi_modal_projector.bias)
        hidden = torch.cat([emb, m.embed_tokens(ids)], dim=1)
        caches = [(torch.zeros(2, 1, 300, 256, dtype=dt, device=DEV), torch.zeros(2, 1, 300, 256, dtype=dt, device=DEV))
                  for _ in m.layers]
        pos = hidden.shape[1]
        out = m._decoder(hidden, 0, caches)
        want = [m.lm_head(out[:, -1:, :])[:, -1].float().argmax(-1)]
        for _ in range(steps - 1):
            out = m._decoder(m.embed_tokens(want[-1][:, None]), pos, caches)
            pos += 1
            want.append(m.lm_head(out)[:, -1].float().argmax(-1))
    assert torch.equal(got, torch.stack(want, dim=1))


def test_generate_refuses_to_overrun_the_cache():
    from vyomai_amd.models import paligemma as P
    vis = P.SiglipVisionConfig(**dict(cases.SIGLIP, num_hidden_layers=1))
    txt = types.SimpleNamespace(**dict(cases.GEMMA, num_hidden_layers=1, vocab_size=512, intermediate_size=512))
    m = P.PaliGemmaForConditionalGeneration(P.PaliGemmaShape(vis, txt, txt.hidden_size)).to(DEV).to(torch.bfloat16).eval()
    img = torch.rand(1, 3, 224, 224, device=DEV)
    ids = torch.randint(3, 512, (1, 4), device=DEV)
    with pytest.raises(ValueError):
        m.generate(img, ids, max_new_tokens=50, max_cache_len=300)     # 260 prefix + 49 > 300
    assert m.generate(img, ids, max_new_tokens=3, max_cache_len=300).shape == (1, 3)


def test_model_and_cached_greedy_loop_fp32_vs_notebook(golden):
    """The whole PaliGemma-shaped model -- SigLIP tower, projector (/ sqrt d), Gemma stack with 1-indexed rotary
    positions, tied head -- and the notebook's cached greedy loop (cell 30 around the notebook's own StaticCache,
    cell 28), at the true widths with 2 + 2 layers: logits to 2e-5 of the reference's, token ids bit-exact
    (tests/golden/paligemma_model.npz, made by exec'ing the notebook cells)."""
    from vyomai_amd.models import paligemma as P
    g = golden("paligemma_model")
    vis = P.SiglipVisionConfig(**dict(cases.SIGLIP, num_hidden_layers=2))
    txt = types.SimpleNamespace(**dict(cases.GEMMA, num_hidden_layers=2, vocab_size=cases.PG_SMALL_VOCAB))
    m = P.PaliGemmaForConditionalGeneration(P.PaliGemmaShape(vis, txt, txt.hidden_size))
    with torch.no_grad():
        for n, p in m.named_parameters():
            ref_name = m.reference_name(n)
            if ref_name.endswith("embed_tokens.weight"):
                # the notebook ties lm_head.weight to the embedding table: one tensor under two state_dict keys,
                # and the generator's fill wrote the later key's values last
                ref_name = "language_model.lm_head.weight"
            p.copy_(T(recipe.param_value("pgm." + ref_name, tuple(p.shape))))
    m = m.to(DEV).eval()
    img = T(recipe.uniform("pgm.img", (1, 3, 224, 224), 0.5, 0.5)).to(DEV)
    ids = T(recipe.token_ids("pgm.ids", (1, 8), 3, cases.PG_IMAGE_TOKEN)).to(DEV)
    with torch.no_grad():
        feats = m.image_features(img)
        close(cases.sub2(feats[0]), g["image_features"], 2e-5 * max(1.0, float(np.abs(g["image_features"]).max())),
              "projected image features")
        hidden = m.prefill(img, ids)
        logits = m.lm_head(hidden)
    scale = max(1.0, float(np.abs(g["prefill.logits.sub"]).max()))
    close(logits[0, ::7, ::5], g["prefill.logits.sub"], 3e-5 * scale, "prefill logits")
    close(logits[:, -1], g["logits.step0"], 3e-5 * scale, "first-token logits")
    toks = m.generate(img, ids, max_new_tokens=8, max_cache_len=288)
    assert np.array_equal(toks.cpu().numpy().reshape(-1), g["generated"].reshape(-1)), (toks, g["generated"])
    # and a state_dict keyed like the notebook's modules loads
    ref_sd = {m.reference_name(n): t.detach().clone() + 1.0 for n, t in m.state_dict().items()}
    before = m.norm.weight.detach().clone()
    m.load_reference_state_dict(ref_sd)
    assert torch.allclose(m.norm.weight, before + 1.0)


def test_single_sequence_step_lean_kernels_vs_general_and_fp32(monkeypatch):
    """B = 1 bf16 decode at the true Gemma-2B widths (d 2048, 8 x 256 query heads on one KV head, MLP 16384): the
    straight-line matrix-vector kernels with the RMSNorms folded into pre-scaled weights and the rotary embedding
    fused into the QKV product (vy_decode.hip) against the general chain (RMSNorm / QKV / RoPE launches), and both
    against the fp32 step on the same weights and cache contents: same error level."""
    import ctypes as C
    from vyomai_amd import _lib
    from vyomai_amd.decode_plan import GemmaDecodePlan
    from vyomai_amd.models import paligemma as P
    lib = _lib.load()
    lib.vy_debug_set_gemma_lean.argtypes = [C.c_int]
    vis = P.SiglipVisionConfig(**dict(cases.SIGLIP, num_hidden_layers=1))
    txt = types.SimpleNamespace(**dict(cases.GEMMA, num_hidden_layers=2, vocab_size=4096))

    def model(dt):
        m = P.PaliGemmaForConditionalGeneration(P.PaliGemmaShape(vis, txt, txt.hidden_size))
        for n, p in m.named_parameters():
            with torch.no_grad():
                p.copy_(T(recipe.param_value("pgl." + n, tuple(p.shape))))
        return m.to(DEV).to(dt).eval()

    def caches(dt):
        g = torch.Generator().manual_seed(3)
        return [(torch.randn(1, 1, 64, 256, generator=g).to(dt).to(DEV), torch.randn(1, 1, 64, 256, generator=g).to(dt).to(DEV))
                for _ in range(2)]

    x = (torch.randn(1, txt.hidden_size, generator=torch.Generator().manual_seed(4)) * 0.3)
    pos = 40
    m16, m32 = model(torch.bfloat16), model(torch.float32)
    res = {}
    try:
        for name, lean, prescale in (("lean", 1, "1"), ("lean_noscale", 1, "0"), ("general", 0, "0")):
            monkeypatch.setenv("VY_GEMMA_PRESCALE", prescale)
            lib.vy_debug_set_gemma_lean(lean)
            cs = caches(torch.bfloat16)
            plan = GemmaDecodePlan(m16, cs, 1)
            res[name] = (plan.step(x.to(torch.bfloat16).to(DEV), pos).float().clone(),
                         [c[0][:, :, pos].float().clone() for c in cs])
            torch.cuda.synchronize()
    finally:
        lib.vy_debug_set_gemma_lean(1)
    cs32 = caches(torch.float32)
    ref = GemmaDecodePlan(m32, cs32, 1).step(x.to(DEV), pos).float()
    e = {k: (v[0] - ref).abs().mean().item() for k, v in res.items()}
    scale = ref.abs().mean().item()
    assert e["general"] < 0.05 * scale + 1e-3, (e, scale)           # the bf16 error level of the general chain
    assert e["lean"] <= 1.3 * e["general"] + 2e-3 * scale, (e, scale)
    assert e["lean_noscale"] <= 1.3 * e["general"] + 2e-3 * scale, (e, scale)
    for name in ("lean", "lean_noscale"):   # rotated K rows written into the cache
        for a, b in zip(res[name][1], res["general"][1]):
            assert (a - b).abs().max().item() <= 0.05 * max(1.0, b.abs().max().item()), name
