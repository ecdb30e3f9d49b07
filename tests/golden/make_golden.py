"""Generate golden vectors by running the REAL reference (Ajax0564/VyomAI) on CPU.

Runs only in the build container, where the reference is mounted read-only:

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo \
        python tests/golden/make_golden.py

It imports the reference package, fills its modules with the deterministic recipe
(`vyomai_amd.recipe`), feeds recipe-derived inputs and stores the *outputs* as small
``.npz`` fixtures next to this file.  Nothing of the reference (source, bytecode) is
written anywhere; fixtures are data only.  The tests regenerate the inputs from the
same recipe, so only outputs (sub-sampled where large) are stored.
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import VyomAI  # noqa: E402  (the reference)
from VyomAI.layers import attention as ref_attn  # noqa: E402
from VyomAI.layers import ffn as ref_ffn  # noqa: E402
from VyomAI.layers import positional_embeddings as ref_pos  # noqa: E402
from VyomAI.layers.kv_cache import DynamicCache, StaticCache, StaticCacheOne, DynamicCacheOne  # noqa: E402
from VyomAI.models import decoder as ref_dec  # noqa: E402
from VyomAI.models.encoder import EncoderModel  # noqa: E402
from VyomAI.models.vision_encoder import Vit  # noqa: E402
from VyomAI.models.multimodel import VisionLanguageModel  # noqa: E402
from VyomAI.generation_utils import generate_multimodel, generate, generate_seq2seq  # noqa: E402
from VyomAI.models import encoder_decoder as ref_s2s  # noqa: E402

from vyomai_amd import recipe  # noqa: E402
from tests.golden import cases  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)


def T(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name: str, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().float().numpy() if v.is_floating_point() else v.detach().numpy()
        out[k] = np.ascontiguousarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  keys={len(out)}")


def filled(module, prefix=""):
    recipe.load_recipe_(module) if not prefix else None
    if prefix:
        sd = module.state_dict()
        for n, t in sd.items():
            if t.is_floating_point():
                t.copy_(T(recipe.param_value(prefix + n, tuple(t.shape))))
    return module.eval()


# ---------------------------------------------------------------------------
# A. module level
# ---------------------------------------------------------------------------


def module_level():
    out = {}
    for tag, cfg in (("micro", cases.micro_cfg()), ("wide", cases.wide_cfg())):
        B, L = cases.MODULE_BL[tag]
        d, h = cfg.hidden_size, cfg.num_attention_heads
        dh = d // h
        x = T(recipe.uniform(f"{tag}.x", (B, L, d)))
        res = T(recipe.uniform(f"{tag}.res", (B, L, d)))
        keypad = T(cases.keypad(B, L))
        add_mask = (1.0 - keypad[:, None, None, :].float()) * torch.finfo(torch.float32).min
        freqs = ref_pos.RotaryEmbedding(cfg)(cfg.max_position_embeddings)[:, :L]

        # rope alone
        q = T(recipe.uniform(f"{tag}.q", (B, h, L, dh)))
        k = T(recipe.uniform(f"{tag}.k", (B, h, L, dh)))
        qe, ke = ref_pos.apply_rotary_pos_emb(q, k, freqs)
        out[f"{tag}.rope.q"], out[f"{tag}.rope.k"] = qe, ke
        out[f"{tag}.rope.angles"] = freqs

        m = filled(ref_ffn.FeedForward(cfg), f"{tag}.ffn.")
        out[f"{tag}.ffn"] = m(x, res)
        m = filled(ref_attn.AttentionSelfOutput(cfg), f"{tag}.aso.")
        out[f"{tag}.aso"] = m(x, res)

        for name, cls in (("enc", ref_attn.EncoderAttention), ("encgqa", ref_attn.EncoderAttentionGqa),
                          ("vis", ref_attn.VisionAttention)):
            m = filled(cls(cfg, 0), f"{tag}.{name}.")
            out[f"{tag}.{name}.keypad"] = m(x, add_mask)
            out[f"{tag}.{name}.nomask"] = m(x, None)
            out[f"{tag}.{name}.keypad.rope"] = m(x, add_mask, freqs)

        # decoder attention with whole-model caches: prefill L then 3 single-token steps
        for name, cls, gqa in (("dec", ref_dec.DecoderAttention, False), ("decgqa", ref_dec.DecoderAttentionGqa, True)):
            for cname in ("static", "dynamic"):
                m = filled(cls(cfg, 0), f"{tag}.{name}.")
                c1 = cases.one_layer(cfg, gqa)
                cache = (StaticCacheOne(c1, max_cache_len=L + 3, batch_size=B) if cname == "static"
                         else DynamicCacheOne(c1))
                full_freqs = ref_pos.RotaryEmbedding(cfg)(cfg.max_position_embeddings)
                causal = cases.causal_additive(B, L, 0, None)
                y, _ = m(x, T(causal), full_freqs[:, :L], True, cache, 0)
                out[f"{tag}.{name}.{cname}.prefill"] = y
                for s in range(3):
                    xs = T(recipe.uniform(f"{tag}.xstep{s}", (B, 1, d)))
                    y, _ = m(xs, None, full_freqs[:, L + s:L + s + 1], True, cache, L + s)
                    out[f"{tag}.{name}.{cname}.step{s}"] = y
                if cname == "static":
                    out[f"{tag}.{name}.static.kcache"] = cache.key_cache[0][:, :, :L + 3].clone()
                    out[f"{tag}.{name}.static.vcache"] = cache.value_cache[0][:, :, :L + 3].clone()
            # causal + key padding + start_pos>0 chunked prefill (second chunk of 5 tokens)
            m = filled(cls(cfg, 0), f"{tag}.{name}.")
            c1 = cases.one_layer(cfg, gqa)
            cache = DynamicCacheOne(c1)
            full_freqs = ref_pos.RotaryEmbedding(cfg)(cfg.max_position_embeddings)
            kp = cases.keypad(B, L)
            y0, _ = m(x[:, :L - 5], T(cases.causal_additive(B, L - 5, 0, kp[:, :L - 5])), full_freqs[:, :L - 5], True, cache, 0)
            y1, _ = m(x[:, L - 5:], T(cases.causal_additive(B, 5, L - 5, kp)), full_freqs[:, L - 5:L], True, cache, L - 5)
            out[f"{tag}.{name}.chunk0"] = y0
            out[f"{tag}.{name}.chunk1"] = y1
    save("modules", **out)


# ---------------------------------------------------------------------------
# B/C. model level + token exact
# ---------------------------------------------------------------------------


def model_level():
    out = {}
    # config 1: EncoderModel(EncoderConfig(), 'absolute'), B=4, L=128
    cfg = VyomAI.EncoderConfig()
    ids = T(recipe.token_ids("enc.ids", (4, 128), 3, cfg.vocab_size))
    am = T(cases.keypad(4, 128)).float()
    for pos, at in (("absolute", None), ("rope", None), ("sinusoidal", "gqa"), ("rope", "gqa")):
        c = cases.with_kv(cfg, at)
        m = filled(EncoderModel(c, pos, at))
        y = m(ids, am).logits
        out[f"encoder.{pos}.{at}.full_nopad"] = cases.sub(m(ids, None).logits)
        out[f"encoder.{pos}.{at}.pad"] = cases.sub(y)
        out[f"encoder.{pos}.{at}.pad.sum"] = np.array([y.double().sum().item(), y.double().abs().sum().item()])

    # decoder, the reference tests' own inputs (tests/test_decoder.py:28-46)
    ids3, am3 = cases.reference_test_inputs()
    ids3, am3 = T(ids3), T(am3)
    for pos in ("absolute", "sinusoidal", "rope"):
        for at in (None, "gqa"):
            c = cases.with_kv(cases.test_cfg(), at)
            m = filled(ref_dec.DecoderModel(c, pos, at))
            o = m(ids3, am3)
            out[f"decoder.{pos}.{at}.hidden"] = o.hidden_state
            out[f"decoder.{pos}.{at}.logits"] = o.logits[:, :, ::97]
            # token-exact greedy generation, three cache modes, prompt of the reference test
            p = torch.tensor([[9226, 16, 5, 1296]], dtype=torch.long)
            a = torch.ones(1, 4, dtype=torch.long)
            out[f"decoder.{pos}.{at}.gen.nocache"] = m.generate(p, a, use_cache=False)
            out[f"decoder.{pos}.{at}.gen.dynamic"] = m.generate(p, a, use_cache=True)
            out[f"decoder.{pos}.{at}.gen.static"] = m.generate(p, a, use_cache=True, use_static_cache=True)
            # batch of 3 equal-length prompts, 6 new tokens
            pb = T(recipe.token_ids("dec.prompt3", (3, 9), 3, c.vocab_size))
            ab = torch.ones(3, 9, dtype=torch.long)
            out[f"decoder.{pos}.{at}.gen3.static"] = m.generate(pb, ab, max_len=6, use_cache=True, use_static_cache=True)
            out[f"decoder.{pos}.{at}.gen3.nocache"] = m.generate(pb, ab, max_len=6, use_cache=False)
            if pos == "rope" and at is None:
                # bf16-on-CPU run of the same model: the reference's own bf16 gap
                mb = filled(ref_dec.DecoderModel(c, pos, at)).to(torch.bfloat16)
                mb.emb_freq = mb.emb_freq  # plain tensor attr stays fp32 (decoder.py:301)
                ob = mb(ids3, am3)
                out["decoder.rope.None.hidden.bf16"] = ob.hidden_state.float()
                # generation_utils.generate (no cache), 4 tokens
                out["decoder.rope.None.utilsgen"] = generate(m, p, max_new_tokens=4)
    save("models_text", **out)

    out = {}
    vcfg = cases.vit_cfg()
    img = T(recipe.uniform("vit.img", (2, 3, 224, 224), 0.5, 0.5))
    vit = filled(Vit(vcfg))
    y = vit(img.clone()).logits
    out["vit.out"] = cases.sub(y)
    out["vit.cls"] = y[:, 0, :]
    for pos, at in (("absolute", None), ("rope", "gqa"), ("rope", None)):
        c = cases.with_kv(cases.test_cfg(), at)
        vlm = VisionLanguageModel(c, Vit(vcfg), pos, at)
        filled(vlm)
        o = vlm(pixel_values=img.clone(), decoder_input_ids=ids3[:2], decoder_attention_mask=am3[:2])
        out[f"vlm.{pos}.{at}.logits"] = o.logits[:, :, ::97]
        enc = vlm.get_encoder_output(pixel_values=img[:1].clone())
        out[f"vlm.{pos}.{at}.enc"] = enc
        idx = torch.tensor([[0]])
        out[f"vlm.{pos}.{at}.gen.nocache"] = generate_multimodel(vlm, enc, None, idx, max_new_tokens=8)
        vlm._clean_cache(); vlm._setup_cache(c)
        out[f"vlm.{pos}.{at}.gen.static"] = generate_multimodel(vlm, enc, None, idx, max_new_tokens=8, use_cache=True)
        vlm._clean_cache(); vlm._setup_cache(c, cls=DynamicCache)
        out[f"vlm.{pos}.{at}.gen.dynamic"] = generate_multimodel(vlm, enc, None, idx, max_new_tokens=8, use_cache=True)
    save("models_vision", **out)


# ---------------------------------------------------------------------------
# D. gradients of a one-layer loss through the reference
# ---------------------------------------------------------------------------


def gradients():
    out = {}
    torch.set_grad_enabled(True)
    for tag, cfg in (("micro", cases.micro_cfg()), ("wide", cases.wide_cfg())):
        B, L = cases.MODULE_BL[tag]
        d = cfg.hidden_size
        for at in (None, "gqa"):
            layer = filled(ref_dec.DecoderLayer(cfg, 0, at), f"{tag}.layer.{at}.")
            x = T(recipe.uniform(f"{tag}.x", (B, L, d))).requires_grad_(True)
            g = T(recipe.uniform(f"{tag}.gout", (B, L, d)))
            freqs = ref_pos.RotaryEmbedding(cfg)(cfg.max_position_embeddings)[:, :L]
            mask = T(cases.causal_additive(B, L, 0, cases.keypad(B, L)))
            y, _ = layer(x, mask, freqs)
            (y * g).sum().backward()
            out[f"{tag}.{at}.y"] = y
            out[f"{tag}.{at}.dx"] = x.grad
            for n, p in layer.named_parameters():
                gr = p.grad
                out[f"{tag}.{at}.d.{n}"] = gr if gr.numel() <= 4096 else cases.sub2(gr)
    torch.set_grad_enabled(False)
    save("grads", **out)


# ---------------------------------------------------------------------------
# D2. configs[3] training: reference autograd through (i) a one-layer Vit (patchify conv, cls token, the
#     in-place double position add, VisionAttention with the fused qkv Linear at L = 197, FeedForward) and
#     (ii) a VisionLanguageModel (1-layer Vit + 1-layer decoder, the image vector prepended as token 0) under
#     the captioning notebooks' loss cross_entropy(logits[:, 1:-1], ids[:, 1:])
#     (Examples/vyom-ai-accelerate-multimodel-2t4.ipynb cell 1).  eval() mode: dropout is the identity.
# ---------------------------------------------------------------------------


def _grad_sample(g: torch.Tensor):
    g = g.detach()
    if g.numel() <= 4096:
        return g
    g2 = g.reshape(g.shape[0], -1) if g.dim() != 3 else g.reshape(-1, g.shape[-1])
    return cases.sub2(g2)


def vision_gradients():
    out = {}
    torch.set_grad_enabled(True)
    vcfg = cases.vit_cfg()
    vcfg.num_hidden_layers = 1
    img = T(recipe.uniform("vgrad.img", (2, 3, 224, 224), 0.5, 0.5))
    vit = filled(Vit(vcfg), "vgrad.vit.")
    y = vit(img.clone()).logits
    g = T(recipe.uniform("vgrad.gout", tuple(y.shape)))
    (y * g).sum().backward()
    out["vit.y"] = cases.sub(y)
    for n, p in vit.named_parameters():
        out["vit.d." + n] = _grad_sample(p.grad)

    for pos, at in (("rope", None), ("absolute", "gqa")):
        c = cases.with_kv(cases.test_cfg(), at)
        c.num_hidden_layers, c.vocab_size = 1, 1031
        vlm = VisionLanguageModel(c, Vit(vcfg), pos, at)
        filled(vlm, f"vgrad.vlm.{pos}.{at}.")
        ids = T(recipe.token_ids("vgrad.ids", (2, 12), 3, c.vocab_size))
        am = torch.ones(2, 12, dtype=torch.long)
        am[1, 9:] = 0
        logits = vlm(pixel_values=img.clone(), decoder_input_ids=ids, decoder_attention_mask=am).logits
        tgt = ids[:, 1:].clone()
        tgt[am[:, 1:] == 0] = -100
        loss = torch.nn.functional.cross_entropy(logits[:, 1:-1].reshape(-1, logits.shape[-1]), tgt.reshape(-1),
                                                 ignore_index=-100)
        loss.backward()
        out[f"vlm.{pos}.{at}.loss"] = loss.detach().reshape(1)
        for n, p in vlm.named_parameters():
            if p.grad is not None:
                out[f"vlm.{pos}.{at}.d." + n] = _grad_sample(p.grad)
    torch.set_grad_enabled(False)
    save("grads_vision", **out)


# ---------------------------------------------------------------------------
# F. seq2seq: EncoderDecoderModel (cross-attention), forward + greedy generation in the three cache
#    modes + gradients of one Seq2SeqDecoderLayer.  The reference tests' own inputs
#    (tests/test_encoder_decoder.py:48-68).
# ---------------------------------------------------------------------------


def seq2seq():
    out = {}
    ids3, am3 = cases.reference_test_inputs()
    ids3, am3 = T(ids3), T(am3)
    for pos, at in (("absolute", None), ("sinusoidal", None), ("rope", None), ("rope", "gqa")):
        c = cases.with_kv(cases.test_cfg(), at)
        m = filled(ref_s2s.EncoderDecoderModel(c, c, None, pos, at, pos, at))
        o = m(input_ids=ids3, attention_mask=am3, decoder_input_ids=ids3, decoder_attention_mask=am3)
        out[f"s2s.{pos}.{at}.logits"] = o.logits[:, :, ::97]
        out[f"s2s.{pos}.{at}.enc"] = o.key_value_states[:, :, ::4]
        # no decoder mask, no encoder mask (both default to all ones)
        o2 = m(input_ids=ids3, decoder_input_ids=ids3[:, :9])
        out[f"s2s.{pos}.{at}.logits.nomask"] = o2.logits[:, :, ::97]
        # greedy generation from <s>, encoder row 0 (the reference StaticCache is batch-1 only)
        enc = m.get_encoder_output(ids3[:1], am3[:1]).logits
        start = torch.tensor([[0]], dtype=torch.long)
        out[f"s2s.{pos}.{at}.gen.nocache"] = generate_seq2seq(m, enc, am3[:1], start, max_new_tokens=7)
        m._setup_cache(c)
        out[f"s2s.{pos}.{at}.gen.static"] = generate_seq2seq(m, enc, am3[:1], start, max_new_tokens=7, use_cache=True)
        m._clean_cache()
        m._setup_cache(c, cls=DynamicCache)
        out[f"s2s.{pos}.{at}.gen.dynamic"] = generate_seq2seq(m, enc, am3[:1], start, max_new_tokens=7, use_cache=True)
        m._clean_cache()
    # gradients through one Seq2SeqDecoderLayer (self-attn -> cross-attn -> FFN), micro and true width
    torch.set_grad_enabled(True)
    for tag, cfg in (("micro", cases.micro_cfg()), ("wide", cases.wide_cfg())):
        B, L = cases.MODULE_BL[tag]
        S = L + 5
        d = cfg.hidden_size
        for at in (None, "gqa"):
            layer = filled(ref_s2s.Seq2SeqDecoderLayer(cfg, 0, at), f"{tag}.s2slayer.{at}.")
            x = T(recipe.uniform(f"{tag}.s2s.x", (B, L, d))).requires_grad_(True)
            enc = T(recipe.uniform(f"{tag}.s2s.enc", (B, S, d))).requires_grad_(True)
            g = T(recipe.uniform(f"{tag}.s2s.gout", (B, L, d)))
            freqs = ref_pos.RotaryEmbedding(cfg)(cfg.max_position_embeddings)[:, :L]
            mask = T(cases.causal_additive(B, L, 0, cases.keypad(B, L)))
            emask = T((1.0 - cases.keypad(B, S)[:, None, None, :].astype(np.float32)) * cases.FMIN)
            y = layer(x, mask, enc, emask, freqs)
            (y * g).sum().backward()
            out[f"grad.{tag}.{at}.y"] = y
            out[f"grad.{tag}.{at}.dx"] = x.grad
            out[f"grad.{tag}.{at}.denc"] = enc.grad
            for n, p_ in layer.named_parameters():
                gr = p_.grad
                out[f"grad.{tag}.{at}.d.{n}"] = gr if gr.numel() <= 4096 else cases.sub2(gr)
    torch.set_grad_enabled(False)
    save("seq2seq", **out)


# ---------------------------------------------------------------------------
# E. PaliGemma-shape blocks: the reference ships them only as notebook cells
#    (Examples/paligemma.ipynb cells 9, 11-13).  The class-definition cells are exec'd here
#    (plain torch + einops code) and run at the true widths on a few tokens.
# ---------------------------------------------------------------------------


def paligemma_blocks():
    import json
    import logging
    import math
    import types
    from einops import rearrange
    nb = json.load(open("/root/reference/Examples/paligemma.ipynb"))
    ns = {"rearrange": rearrange, "math": math, "Cache": object, "GemmaConfig": object,
          "logger": logging.getLogger("pg"), "Optional": Optional, "Tuple": Tuple, "dataclass": dataclass,
          "torch": torch, "nn": torch.nn}
    import typing
    ns.update({k: getattr(typing, k) for k in ("List", "Union", "Dict", "Any")})
    for i in (9, 11, 12, 13):
        exec("".join(nb["cells"][i]["source"]), ns)
    out = {}
    scfg = ns["SiglipVisionConfig"](**cases.SIGLIP)
    layer = filled(ns["SiglipEncoderLayer"](scfg), "pg.siglip.")
    x = T(recipe.uniform("pg.siglip.x", (2, 20, scfg.hidden_size)))
    out["siglip.layer"] = layer(x)
    gcfg = types.SimpleNamespace(**cases.GEMMA)
    glayer = filled(ns["GemmaDecoderLayer"](gcfg, 0), "pg.gemma.")
    xg = T(recipe.uniform("pg.gemma.x", (2, 12, gcfg.hidden_size)))
    pos = torch.arange(12)[None, :].expand(2, -1)
    out["gemma.layer.nomask"] = glayer(xg, attention_mask=None, position_ids=pos)[0]
    causal = T(cases.causal_additive(2, 12, 0, None))
    out["gemma.layer.causal"] = glayer(xg, attention_mask=causal, position_ids=pos)[0]
    out["gemma.layer.pos7"] = glayer(xg, attention_mask=causal, position_ids=pos + 7)[0]
    norm = filled(ns["GemmaRMSNorm"](gcfg.hidden_size, eps=gcfg.rms_norm_eps), "pg.norm.")
    out["gemma.rmsnorm"] = norm(xg)
    save("paligemma_blocks", **{k: cases.sub2(v.reshape(-1, v.shape[-1])) if v.numel() > 20000 else v for k, v in out.items()})


# ---------------------------------------------------------------------------
# E2. the PaliGemma MODEL and its cached greedy loop: notebook cells 9, 11-13, 15-17 (classes) and 28 (the
#     notebook's StaticCache on transformers.cache_utils.Cache) exec'd at the true widths with reduced depth
#     (2 SigLIP + 2 Gemma layers) and a small vocabulary; test_inference (cell 30) restated line by line around
#     them (it needs the remote processor only to build the inputs, which are synthetic here).
# ---------------------------------------------------------------------------


def paligemma_model():
    import json
    import logging
    import math
    import typing
    from einops import rearrange
    nb = json.load(open("/root/reference/Examples/paligemma.ipynb"))
    ns = {"rearrange": rearrange, "math": math, "logger": logging.getLogger("pg"), "Optional": Optional,
          "Tuple": Tuple, "dataclass": dataclass, "torch": torch, "nn": torch.nn}
    ns.update({k: getattr(typing, k) for k in ("List", "Union", "Dict", "Any")})
    # The notebook's StaticCache (cell 28) derives from transformers.cache_utils.Cache only to call a
    # no-argument super().__init__(); the transformers installed here (5.x) made that constructor take the
    # layer list, so the base is an empty class here and the cell's own import line is dropped.  Everything the
    # loop uses -- update() with cache_position, get_seq_length(), get_max_length() -- is the cell's own code.
    class Cache:   # noqa: D401
        def __init__(self):
            pass
    ns["Cache"] = Cache
    for i in (9, 11, 12, 13, 15, 16, 17, 28):
        src = "".join(nb["cells"][i]["source"]).replace("from transformers.cache_utils import Cache\n", "")
        exec(src, ns)
    vis = dict(cases.SIGLIP, num_hidden_layers=2)
    txt = {k: v for k, v in cases.GEMMA.items() if k != "pad_token_id"}
    txt.update(num_hidden_layers=2, vocab_size=cases.PG_SMALL_VOCAB)
    cfg = ns["PaliGemmaConfig"](vision_config=vis, text_config=txt, image_token_index=cases.PG_IMAGE_TOKEN,
                                vocab_size=cases.PG_SMALL_VOCAB, projection_dim=2048, hidden_size=2048, pad_token_id=0)
    model = ns["PaliGemmaForConditionalGeneration"](cfg)
    model.tie_weights()
    filled(model, "pgm.")
    n_img = (vis["image_size"] // vis["patch_size"]) ** 2
    img = T(recipe.uniform("pgm.img", (1, 3, 224, 224), 0.5, 0.5))
    text = T(recipe.token_ids("pgm.ids", (1, 8), 3, cases.PG_IMAGE_TOKEN))
    input_ids = torch.cat([torch.full((1, n_img), cases.PG_IMAGE_TOKEN, dtype=torch.long), text], dim=1)
    attention_mask = torch.ones_like(input_ids)
    out = {}
    # cell 30, test_inference, greedy branch
    cache = ns["StaticCache"](cfg.text_config, batch_size=1, device="cpu", dtype=torch.float32, max_cache_len=288)
    generated = []
    ids, am, pv = input_ids, attention_mask, img
    for step in range(8):
        o = model(input_ids=ids, pixel_values=pv, attention_mask=am, past_key_values=cache, use_cache=True)
        cache = o.past_key_values
        logits = o.logits[:, -1, :]
        if step < 3:
            out[f"logits.step{step}"] = logits
        if step == 0:
            out["prefill.logits.sub"] = o.logits[0, ::7, ::5]
            out["image_features"] = cases.sub2(o.image_hidden_states[0])
        nxt = torch.argmax(logits, dim=-1, keepdim=True)
        generated.append(nxt.squeeze(0))
        ids = nxt
        am = torch.cat([am, torch.ones((1, 1))], dim=-1)
        pv = None if False else pv   # the notebook passes pixel_values on every call (cell 30)
    out["generated"] = torch.cat(generated, dim=-1)
    save("paligemma_model", **out)


# ---------------------------------------------------------------------------
# G. sampling processors and speculative decoding
# ---------------------------------------------------------------------------


def sampling():
    import types
    from VyomAI import logits_processors as ref_lp
    from VyomAI import speculative_decoding as ref_sd

    out = {}
    logits = T(cases.sampling_logits())
    for name, (cls, args) in cases.PROCESSORS.items():
        proc = getattr(ref_lp, cls)(*args)
        out[f"proc.{name}.probs"] = proc(logits.clone())
        out[f"proc.{name}.masked"] = proc._process(logits.clone())
        out[f"proc.{name}.argmax"] = ref_lp.GreedyProcessor.sample(proc, proc(logits.clone()))

    class HFLike(torch.nn.Module):
        """What speculative_generate expects of a model (speculative_decoding.py:150-154), around the
        reference's own DecoderModel; use_cache=False semantics (the prefix is recomputed)."""

        def __init__(self, m):
            super().__init__()
            self.m, self.config, self.device = m, m.config, torch.device("cpu")

        def forward(self, input_ids, past_key_values=None, use_cache=False):
            o = self.m(input_ids, torch.ones_like(input_ids))
            return types.SimpleNamespace(logits=o.logits, past_key_values=past_key_values)

    draws = T(cases.speculative_draws())
    real_rand = torch.rand
    for name, c in cases.SPECULATIVE.items():
        tcfg = cases.with_kv(cases.test_cfg(), None)
        tcfg.num_hidden_layers = c["target_layers"]
        dcfg = cases.with_kv(cases.test_cfg(), None)
        dcfg.num_hidden_layers = c["drafter_layers"]
        target = HFLike(filled(ref_dec.DecoderModel(tcfg, "rope", None), "spec.target."))
        drafter = HFLike(filled(ref_dec.DecoderModel(dcfg, "rope", None), c["drafter_prefix"]))
        prompt = T(recipe.token_ids("spec.prompt", (1, c["prompt_len"]), 3, tcfg.vocab_size))
        state = {"i": 0}

        def fixed_rand(n, device=None):   # the acceptance draws of speculative_decoding.py:200, made reproducible
            a = draws[state["i"]:state["i"] + n].clone()
            state["i"] += n
            return a

        torch.rand = fixed_rand
        try:
            cls, args = c["processor"]
            ids, rate = ref_sd.speculative_generate(
                prompt, drafter, target, gamma=c["gamma"], logits_processor=getattr(ref_lp, cls)(*args),
                max_gen_len=c["max_gen_len"], eos_tokens_id=c["eos"], pad_token_id=2, use_cache=False,
                skip_sample_adjustment=c["skip"], first_target=c["first_target"])
        finally:
            torch.rand = real_rand
        out[f"spec.{name}.ids"] = np.array(ids, dtype=np.int64)
        out[f"spec.{name}.rate"] = np.array([rate], dtype=np.float64)
        out[f"spec.{name}.draws_used"] = np.array([state["i"]], dtype=np.int64)
        print(name, len(ids), rate, state["i"])
    save("sampling", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["modules", "models", "grads", "vgrads", "paligemma", "pgmodel", "seq2seq", "sampling"]
    if "modules" in which:
        module_level()
    if "models" in which:
        model_level()
    if "grads" in which:
        gradients()
    if "vgrads" in which:
        vision_gradients()
    if "paligemma" in which:
        paligemma_blocks()
    if "pgmodel" in which:
        paligemma_model()
    if "seq2seq" in which:
        seq2seq()
    if "sampling" in which:
        sampling()
