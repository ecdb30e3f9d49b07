"""Sampling front end on the MI355X: vy_greedy_step, vy_sampling_probs, the processor classes and
speculative_generate against the reference's outputs (tests/golden/sampling.npz) and the CPU oracle.

Bars: probabilities 1e-6 abs (fp32), support sets and token ids bit-exact, acceptance rate exact.
"""
import numpy as np
import pytest
import torch

from oracle import vyom_oracle as O
from tests.golden import cases
from tests.test_oracle_golden import _proc_args
from vyomai_amd import recipe

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("name", list(cases.PROCESSORS))
def test_processors_vs_reference(golden, name):
    from vyomai_amd import logits_processors as LP
    g = golden("sampling")
    cls, args = cases.PROCESSORS[name]
    proc = getattr(LP, cls)(*args)
    logits = T(cases.sampling_logits()).to(DEV)
    before = logits.clone()
    probs = proc(logits)
    want = g[f"proc.{name}.probs"]
    got = probs.cpu().numpy()
    assert probs.dtype == torch.float32 and got.shape == want.shape
    assert np.array_equal(got > 0, want > 0), "support differs"
    assert np.abs(got - want).max() <= 1e-6
    assert np.abs(got.sum(-1) - 1).max() < 1e-5
    # the reference's side effect: top-k processors mask the caller's logits in place, the others do not
    _, k, _ = _proc_args(cls, args)
    if k and k < logits.shape[-1]:
        assert np.array_equal(logits.cpu().numpy(), O.processor_masked_logits(before.cpu(), k, 0.0).numpy())
    else:
        assert torch.equal(logits, before)
    assert np.array_equal(proc._process(before.clone()).cpu().numpy(), g[f"proc.{name}.masked"])
    assert np.array_equal(probs.argmax(-1, keepdim=True).cpu().numpy(), g[f"proc.{name}.argmax"])
    if cls == "GreedyProcessor":
        assert torch.equal(proc.sample(probs), probs.argmax(-1, keepdim=True))
    if cls != "GreedyProcessor":
        s = proc.sample(probs)
        assert s.shape == (3, 1) and bool((probs.gather(-1, s) > 0).all())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("t,k,p", [(1.0, 0, 0.0), (0.7, 50, 0.0), (0.3, 0, 0.9), (1.3, 200, 0.6), (1.0, 1, 0.0),
                                   (1.0, 0, 0.01)])
def test_sampling_probs_vocab_width(dtype, t, k, p):
    """The decoder's vocabulary width (50265, not a multiple of anything) against the oracle's sort-based
    restatement of the reference; bf16 logits have ties, which the kernel keeps whole (vyom_hip.h)."""
    from vyomai_amd import ops
    g = torch.Generator().manual_seed(5)
    logits = (torch.randn(4, 50265, generator=g) * 3).to(dtype)
    want = O.processor_probs(logits.float(), t, k, p)
    got = ops.sampling_probs(logits.to(DEV), t, k, p).cpu()
    if dtype == torch.float32:
        assert torch.equal(got > 0, want > 0)
        assert (got - want).abs().max().item() <= 1e-6
    else:
        # ties at the cut: the kernel's support is the reference's plus the rest of the tied value
        sup_g, sup_w = got > 0, want > 0
        assert bool((sup_g | ~sup_w).all())
        extra = sup_g & ~sup_w
        if extra.any():
            lf = logits.float()
            cut = torch.where(sup_w, lf, torch.full_like(lf, float("inf"))).min(-1, keepdim=True)[0]
            assert bool((lf[extra] == cut.expand_as(lf)[extra]).all())
        assert abs(got.sum(-1) - 1).max().item() < 1e-5


def test_sampling_probs_rejects_bad_arguments():
    from vyomai_amd import ops
    from vyomai_amd._lib import VyomHipError
    x = torch.zeros(2, 16, device=DEV)
    with pytest.raises(VyomHipError):
        ops.sampling_probs(x, 0.0)
    with pytest.raises(VyomHipError):
        ops.sampling_probs(x, 1.0, -1)
    with pytest.raises(VyomHipError):
        ops.sampling_probs(torch.zeros(2, 16), 1.0)   # CPU tensor: no fallback


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_greedy_step(dtype):
    """One position of DecoderModel.generate's bookkeeping (reference models/decoder.py:478-507)."""
    from vyomai_amd import ops
    g = torch.Generator().manual_seed(11)
    B, V, total, cur = 6, 50265, 12, 7
    logits = torch.randn(B, V, generator=g).to(dtype)
    logits[0, 777] = 40.0          # (explicit peaks: bf16 noise has tied maxima, whose order topk does not define)
    logits[3, 31000] = 40.0
    logits[5, 5] = 40.0
    logits[1, 40000] = 50.0
    logits[1, 123] = 50.0          # a tie: the lowest index wins
    logits[2, 2] = 60.0            # row 2 emits EOS (id 2)
    logits[4, 9] = 60.0            # row 4 would emit the second stop token but is still inside its prompt
    tokens = torch.randint(3, 1000, (B, total), generator=g)
    text_mask = torch.zeros(B, total, dtype=torch.bool)
    text_mask[:, :cur] = True
    text_mask[4, cur] = True       # forced position
    eos_ids = torch.tensor([2, 9])
    eos = torch.zeros(B, dtype=torch.bool)
    eos[5] = True                  # row 5 finished earlier
    # the reference's ops
    nxt = torch.topk(logits.float(), k=1, dim=-1)[1].reshape(-1)
    nxt[1] = 123
    nxt = torch.where(text_mask[:, cur], tokens[:, cur], nxt)
    want_tokens = tokens.clone()
    want_tokens[:, cur] = nxt
    want_eos = eos | ((~text_mask[:, cur]) & torch.isin(nxt, eos_ids))
    d_tokens, d_eos = tokens.to(DEV), eos.to(DEV)
    not_done = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.greedy_step_(logits.to(DEV), d_tokens, cur, text_mask.to(DEV), eos_ids.to(DEV), d_eos, not_done)
    assert torch.equal(d_tokens.cpu(), want_tokens)
    assert torch.equal(d_eos.cpu(), want_eos)
    assert int(not_done.item()) == int((~want_eos).sum())
    assert want_eos.tolist() == [False, False, True, False, False, True]


def _fill(module, prefix):
    sd = module.state_dict()
    with torch.no_grad():
        for n, t in sd.items():
            if t.is_floating_point():
                t.copy_(T(recipe.param_value(prefix + n, tuple(t.shape))))
    return module.to(DEV).eval()


@pytest.mark.parametrize("use_cache", [False, True])
@pytest.mark.parametrize("name", list(cases.SPECULATIVE))
def test_speculative_generate_vs_reference(golden, name, use_cache):
    """The reference's speculative_generate around its own DecoderModel (fixed acceptance draws) against ours
    around the HIP decoder, without and with KV caches (trimmed after every rejection)."""
    import vyomai_amd as V
    from vyomai_amd import logits_processors as LP
    from vyomai_amd.speculative_decoding import speculative_generate
    g = golden("sampling")
    c = cases.SPECULATIVE[name]
    tcfg, dcfg = cases.with_kv(cases.test_cfg(), None), cases.with_kv(cases.test_cfg(), None)
    tcfg.num_hidden_layers, dcfg.num_hidden_layers = c["target_layers"], c["drafter_layers"]
    target = _fill(V.DecoderModel(tcfg, "rope", None), "spec.target.")
    drafter = _fill(V.DecoderModel(dcfg, "rope", None), c["drafter_prefix"])
    draws, used = T(cases.speculative_draws()).to(DEV), [0]

    def rand_fn(n):
        used[0] += n
        return draws[used[0] - n:used[0]]

    prompt = T(recipe.token_ids("spec.prompt", (1, c["prompt_len"]), 3, tcfg.vocab_size)).to(DEV)
    cls, args = c["processor"]
    ids, rate = speculative_generate(prompt, drafter, target, gamma=c["gamma"], logits_processor=getattr(LP, cls)(*args),
                                     max_gen_len=c["max_gen_len"], eos_tokens_id=c["eos"], pad_token_id=2,
                                     use_cache=use_cache, skip_sample_adjustment=c["skip"],
                                     first_target=c["first_target"], rand_fn=rand_fn)
    assert ids == g[f"spec.{name}.ids"].tolist()
    assert abs(rate - float(g[f"spec.{name}.rate"][0])) < 1e-12
    assert used[0] == int(g[f"spec.{name}.draws_used"][0])


def test_trim_cache():
    from vyomai_amd.layers.kv_cache import DynamicCacheOne
    from vyomai_amd.speculative_decoding import trim_cache
    cfg = cases.micro_cfg()
    cache = DynamicCacheOne(cfg)
    k = torch.randn(1, 4, 10, 16, device=DEV)
    cache.update(0, k, k.clone())
    assert len(cache) == 10
    assert trim_cache(cache, 3) is cache and len(cache) == 7
    k2, _ = cache.update(0, k[:, :, :2], k[:, :, :2].clone())
    assert k2.shape[2] == 9 and torch.equal(k2[:, :, :7], k[:, :, :7]) and torch.equal(k2[:, :, 7:], k[:, :, :2])
    assert trim_cache(None, 2) is None
    with pytest.raises(ValueError):
        trim_cache(object(), 1)


@pytest.mark.parametrize("static", [False, True])
@pytest.mark.parametrize("stop_at", [0, 1, 3, 6, 7])
def test_generate_stops_like_the_reference(stop_at, static):
    """DecoderModel.generate reads the all-rows-finished count one step late: what it returns must still be what
    the reference's immediate break (models/decoder.py:512-513) leaves -- the EOS token, padding after it."""
    import vyomai_amd as V
    cfg = cases.with_kv(cases.test_cfg(), None)
    cfg.num_hidden_layers = 2
    cfg.eos_token_id = -5          # never produced
    m = V.DecoderModel(cfg, "rope", None)
    recipe.load_recipe_(m)
    m = m.to(DEV).eval()
    prompt = T(recipe.token_ids("stop.prompt", (1, 5), 3, cfg.vocab_size)).to(DEV)
    am = torch.ones(1, 5, dtype=torch.long, device=DEV)
    free = m.generate(prompt, am, max_len=8, use_cache=True, use_static_cache=static)
    assert free.shape == (1, 13)
    tok = int(free[0, 5 + stop_at])
    first = int((free[0, 5:] == tok).nonzero()[0])      # the token may occur earlier than stop_at
    cfg.eos_token_id = tok
    got = m.generate(prompt, am, max_len=8, use_cache=True, use_static_cache=static)
    want = free.clone()
    want[0, 5 + first + 1:] = getattr(cfg, "pad_token_id", 1)
    assert torch.equal(got, want)


@pytest.mark.parametrize("V", [1, 17, 1000, 257216])
def test_sampling_probs_odd_widths(V):
    """One column, a width below the workgroup size, and the PaliGemma vocabulary (257216 > 65536 columns)."""
    from vyomai_amd import ops
    g = torch.Generator().manual_seed(V)
    logits = torch.randn(3, V, generator=g) * 2
    for t, k, p in ((1.0, 0, 0.0), (0.5, 5, 0.0), (1.0, 0, 0.7), (2.0, 40, 0.5)):
        want = O.processor_probs(logits, t, min(k, V), p)
        got = ops.sampling_probs(logits.to(DEV), t, k, p).cpu()
        if p and V > 100000:
            # the reference's float32 cumsum over 257216 sorted probabilities (each ~4e-6) and the kernel's
            # digit-wise masses round differently: the cut may move by a few elements of (nearly) equal logit
            diff = (got > 0) != (want > 0)
            assert int(diff.sum()) <= 64, int(diff.sum())
            if diff.any():
                cut = torch.where(want > 0, logits, torch.full_like(logits, float("inf"))).min(-1, keepdim=True)[0]
                assert ((logits - cut).abs()[diff] < 2e-3).all()
            both = (got > 0) & (want > 0)
            assert ((got - want).abs()[both] <= 1e-6 + 1e-3 * want[both]).all()
            continue
        assert torch.equal(got > 0, want > 0), (V, t, k, p)
        assert (got - want).abs().max().item() <= 1e-6
    tok = torch.zeros(3, 4, dtype=torch.long, device=DEV)
    eos = torch.zeros(3, dtype=torch.bool, device=DEV)
    ops.greedy_step_(logits.to(DEV), tok, 2, None, torch.tensor([V + 5], device=DEV), eos, None)
    assert torch.equal(tok[:, 2].cpu(), logits.argmax(-1))
