"""Seq2seq (encoder-decoder with cross-attention) on the MI355X vs the golden vectors the real
reference produced (tests/golden/seq2seq.npz): fp32 logits to 1e-5 (a few 1e-5 on the 50265-wide
logits, as for the decoder), greedy generate_seq2seq token ids bit-exact in the no-cache / static /
dynamic cache modes, bf16 within the bf16 bar, and the gradients of one Seq2SeqDecoderLayer
(w.r.t. decoder state, ENCODER output and every parameter) against the reference's autograd."""
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


def close(got, want, atol, what=""):
    got = got.detach().float().cpu().numpy()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), what
    err = np.abs(got - want).max()
    assert err <= atol, f"{what}: max abs err {err:.3e} > {atol}"


def rel_err(got, want):
    got = got.detach().float().cpu().numpy()
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-12))


def build(pos, at, dtype=torch.float32):
    import vyomai_amd as V
    cfg = cases.with_kv(cases.test_cfg(), at)
    cfg.hidden_dropout_prob = 0.0
    m = V.EncoderDecoderModel.from_config(cfg, cfg, None, pos, at, pos, at)
    recipe.load_recipe_(m)
    return m.to(DEV).to(dtype).eval(), cfg


@pytest.mark.parametrize("pos,at", [("absolute", None), ("sinusoidal", None), ("rope", None), ("rope", "gqa")])
def test_seq2seq_fp32(golden, pos, at):
    import vyomai_amd as V
    g = golden("seq2seq")
    m, cfg = build(pos, at)
    ids, am = cases.reference_test_inputs()
    ids, am = T(ids).to(DEV), T(am).to(DEV)
    with torch.no_grad():
        o = m(input_ids=ids, attention_mask=am, decoder_input_ids=ids, decoder_attention_mask=am)
        o2 = m(input_ids=ids, decoder_input_ids=ids[:, :9])
    assert list(o.logits.shape) == [3, 17, cfg.vocab_size] and list(o.key_value_states.shape) == [3, 17, 768]
    close(o.key_value_states[:, :, ::4], g[f"s2s.{pos}.{at}.enc"], 1e-5, "encoder output")
    close(o.logits[:, :, ::97], g[f"s2s.{pos}.{at}.logits"], 3e-5, "logits")
    close(o2.logits[:, :, ::97], g[f"s2s.{pos}.{at}.logits.nomask"], 3e-5, "logits without masks")
    # greedy generation, encoder row 0 (the reference's StaticCache is batch-1 only)
    with torch.no_grad():
        enc = m.get_encoder_output(ids[:1], am[:1]).logits
    start = torch.tensor([[0]], dtype=torch.long, device=DEV)
    t = V.generate_seq2seq(m, enc, am[:1], start, max_new_tokens=7)
    assert np.array_equal(t.cpu().numpy(), g[f"s2s.{pos}.{at}.gen.nocache"])
    m._setup_cache(cfg)
    t = V.generate_seq2seq(m, enc, am[:1], start, max_new_tokens=7, use_cache=True)
    assert np.array_equal(t.cpu().numpy(), g[f"s2s.{pos}.{at}.gen.static"])
    m._clean_cache()
    m._setup_cache(cfg, cls=V.DynamicCache)
    t = V.generate_seq2seq(m, enc, am[:1], start, max_new_tokens=7, use_cache=True)
    assert np.array_equal(t.cpu().numpy(), g[f"s2s.{pos}.{at}.gen.dynamic"])
    m._clean_cache()
    # the reference's dense additive (B,1,1,S) encoder mask is accepted by the layer API too
    with torch.no_grad():
        dense = ((1.0 - am[:, None, None, :].float()) * torch.finfo(torch.float32).min)
        h1 = m.decoder(input_ids=ids, attention_mask=am, encoder_hidden_state=o.key_value_states,
                       encoder_attention_mask=dense)
        from vyomai_amd.layers.mask import AttnMask
        h2 = m.decoder(input_ids=ids, attention_mask=am, encoder_hidden_state=o.key_value_states,
                       encoder_attention_mask=AttnMask.from_padding(am, causal=False, start_pos=0, query_len=17))
    assert (h1 - h2).abs().max().item() < 2e-5


def test_seq2seq_bf16(golden):
    g = golden("seq2seq")
    m, cfg = build("rope", None, BF)
    ids, am = cases.reference_test_inputs()
    ids, am = T(ids).to(DEV), T(am).to(DEV)
    with torch.no_grad():
        o = m(input_ids=ids, attention_mask=am, decoder_input_ids=ids, decoder_attention_mask=am)
    want = g["s2s.rope.None.logits"]
    err = np.abs(o.logits[:, :, ::97].float().cpu().numpy() - want)
    # bf16 storage of 8 layers of activations: the decoder's own bf16 bar (mean << max)
    assert err.mean() < 0.03 and err.max() < 0.5, (err.mean(), err.max())


@pytest.mark.parametrize("at", [None, "gqa"])
def test_seq2seq_layer_gradients_vs_reference(golden, at):
    from vyomai_amd.layers.mask import AttnMask
    from vyomai_amd.layers.positional_embeddings import RopeSlice, RopeTable
    from vyomai_amd.models.encoder_decoder import Seq2SeqDecoderLayer
    g = golden("seq2seq")
    tag = "wide"
    cfg = cases.wide_cfg()
    cfg.hidden_dropout_prob = 0.0
    B, L = cases.MODULE_BL[tag]
    S = L + 5
    d = cfg.hidden_size
    dh = d // cfg.num_attention_heads
    layer = Seq2SeqDecoderLayer(cfg, 0, at)
    for n, t in layer.state_dict().items():
        t.copy_(T(recipe.param_value(f"{tag}.s2slayer.{at}." + n, tuple(t.shape))))
    layer = layer.to(DEV).train()
    x = T(recipe.uniform(f"{tag}.s2s.x", (B, L, d))).to(DEV).to(BF).requires_grad_(True)
    enc = T(recipe.uniform(f"{tag}.s2s.enc", (B, S, d))).to(DEV).to(BF).requires_grad_(True)
    gout = T(recipe.uniform(f"{tag}.s2s.gout", (B, L, d))).to(DEV).to(BF)
    mask = AttnMask.from_padding(T(cases.keypad(B, L)).to(DEV), causal=True, start_pos=0, query_len=L)
    emask = AttnMask.from_padding(T(cases.keypad(B, S)).to(DEV), causal=False, start_pos=0, query_len=L)
    freqs = RopeSlice(RopeTable(O.rotary_angles(dh, cfg.max_position_embeddings)), 0, L)
    y = layer(x, mask, enc, emask, freqs)
    (y.float() * gout.float()).sum().backward()
    assert rel_err(y, g[f"grad.{tag}.{at}.y"]) < 3e-2
    assert rel_err(x.grad, g[f"grad.{tag}.{at}.dx"]) < 5e-2, rel_err(x.grad, g[f"grad.{tag}.{at}.dx"])
    assert rel_err(enc.grad, g[f"grad.{tag}.{at}.denc"]) < 5e-2, rel_err(enc.grad, g[f"grad.{tag}.{at}.denc"])
    for n, p in layer.named_parameters():
        want = g[f"grad.{tag}.{at}.d.{n}"]
        got = p.grad if p.grad.numel() <= 4096 else cases.sub2(p.grad)
        if np.abs(want).max() < 1e-5:
            # softmax is invariant to a bias shared by every key: the reference's key-bias gradient
            # is rounding noise around an exact zero -- compare absolutely
            assert got.detach().float().abs().max().item() < 2e-3, (n, got.abs().max().item())
            continue
        e = rel_err(got, want)
        assert e < 6e-2, (n, e)


def test_seq2seq_training_step_runs():
    """FlatTrainer on the encoder-decoder model: loss decreases over a few AdamW steps (gradients reach
    the encoder through the cross-attention)."""
    import vyomai_amd as V
    from vyomai_amd.training import FlatTrainer
    cfg = cases.test_cfg()
    cfg.num_hidden_layers, cfg.vocab_size, cfg.hidden_dropout_prob = 2, 1031, 0.0
    m = V.EncoderDecoderModel.from_config(cfg, cfg, None, "rope", None, "rope", None)
    recipe.load_recipe_(m)
    m = m.to(DEV).train()
    src = T(recipe.token_ids("s2s.src", (4, 40), 3, cfg.vocab_size)).to(DEV)
    tgt = T(recipe.token_ids("s2s.tgt", (4, 32), 3, cfg.vocab_size)).to(DEV)
    am = T(cases.keypad(4, 40)).to(DEV)
    tr = FlatTrainer(m, lr=2e-3)
    enc_w = m.encoder.all_layer[0].attention.query.weight
    before = enc_w.detach().clone()
    losses = [tr.train_step(lambda: m.seq2seq_loss(src, am, tgt, tgt)).item() for _ in range(6)]
    assert losses[-1] < losses[0] - 0.05, losses
    assert (enc_w.detach() - before).abs().max().item() > 0, "the encoder received no update"
