"""CPU-only checks (no compute calls): the C-ABI library loads and exports every symbol that
include/vyom_hip.h declares, host-side logic (mask descriptors, caches, state_dict layout, error
behaviour) matches the reference's, and the product path FAILS LOUDLY without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests.golden import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from vyomai_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vyom_hip.h")).read()
    declared = set(re.findall(r"\b(vy_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vy_dtype", "vy_status", "vy_act"}
    assert declared == set(_lib.ALL_SYMBOLS), declared ^ set(_lib.ALL_SYMBOLS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().vy_abi_version() == 5


def test_argument_errors_are_reported_without_a_gpu():
    """Argument validation happens before any launch: a NULL operand is a VY_ERR_ARG."""
    from vyomai_amd import _lib
    with pytest.raises(_lib.VyomHipError, match="null operand"):
        _lib.call("vy_linear_fwd", None, 8, None, 8, None, None, 0, None, 8, None, 4, 8, 8, 0, 1, None)
    with pytest.raises(_lib.VyomHipError, match="multiple of 8"):
        buf = (ctypes.c_char * 4096)()
        p = (ctypes.addressof(buf) + 15) // 16 * 16
        _lib.call("vy_linear_fwd", p, 12, p, 12, None, None, 0, p, 16, None, 4, 8, 12, 0, 1, None)
    # the data-parallel surface: nothing but argument checks can run here (no GPU, no communicator)
    lib = _lib.load()
    assert lib.vy_ddp_world() == 0 and lib.vy_ddp_rank() == -1
    with pytest.raises(_lib.VyomHipError, match="no communicator|RCCL is not available"):
        _lib.call("vy_ddp_all_reduce_async", p, 16, 0, None)
    with pytest.raises(_lib.VyomHipError, match="bad arguments|RCCL is not available"):
        _lib.call("vy_ddp_init", p, 3, 2)
    _lib.call("vy_ddp_destroy")


def test_no_cpu_fallback():
    import vyomai_amd as V
    from vyomai_amd._lib import VyomHipError
    cfg = cases.micro_cfg()
    m = V.DecoderModel(cfg, "rope", None).eval()
    with pytest.raises(VyomHipError, match="MI355X"):
        m(torch.zeros(1, 4, dtype=torch.long))


def test_state_dict_layout_matches_reference():
    import vyomai_amd as V
    for pos in ("absolute", "sinusoidal", "rope"):
        for at in (None, "gqa"):
            cfg = cases.with_kv(cases.micro_cfg(), at)
            m = V.DecoderModel(cfg, pos, at)
            want = cases.text_model_shapes(cfg, pos, at, head=True)
            got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
            assert got == {k: tuple(v) for k, v in want.items()}, (pos, at)
            e = V.EncoderModel(cfg, pos, at)
            want = cases.text_model_shapes(cfg, pos, at, head=False)
            assert {k: tuple(v.shape) for k, v in e.state_dict().items()} == {k: tuple(v) for k, v in want.items()}
    v = V.Vit(cases.vit_cfg())
    assert {k: tuple(t.shape) for k, t in v.state_dict().items()} == {k: tuple(s) for k, s in cases.vit_shapes(cases.vit_cfg()).items()}
    # the vocabulary bias is ONE tied parameter under two keys (reference models/decoder.py:263-265)
    m = V.DecoderModel(cases.micro_cfg(), "rope", None)
    assert m.lm_head.bias is m.lm_head.decoder.bias
    # seq2seq: encoder.*, decoder.* (attention + cross_attention + feed_forward), lm_head.vocab tied to
    # lm_head.bias (reference models/encoder_decoder.py:86-99, 261-284)
    for pos, at in (("absolute", None), ("rope", "gqa")):
        cfg = cases.with_kv(cases.micro_cfg(), at)
        s2s = V.EncoderDecoderModel.from_config(cfg, cfg, None, pos, at, pos, at)
        want = cases.s2s_model_shapes(cfg, pos, at)
        assert {k: tuple(t.shape) for k, t in s2s.state_dict().items()} == {k: tuple(v) for k, v in want.items()}
        assert s2s.lm_head.bias is s2s.lm_head.vocab.bias


def test_constructor_errors_match_reference():
    import vyomai_amd as V
    from vyomai_amd.layers.attention import EncoderAttention, EncoderAttentionGqa
    bad = cases.micro_cfg()
    bad.hidden_size = 65
    with pytest.raises(ValueError, match="not a multiple of the number of attention"):
        EncoderAttention(bad, 0)
    bad = cases.micro_cfg()
    bad.num_key_value_heads = 3
    with pytest.raises(ValueError, match="num_key_value_heads"):
        EncoderAttentionGqa(bad, 0)
    from vyomai_amd.layers.positional_embeddings import AbsoluteEncoding
    with pytest.raises(ValueError, match="max_position_embeddings"):
        AbsoluteEncoding(cases.micro_cfg())(10_000)
    from vyomai_amd.models.decoder import DecoderAttention
    att = DecoderAttention(cases.micro_cfg(), 0)
    with pytest.raises(ValueError, match="need to pass kv_cache"):
        att(torch.zeros(1, 2, 64), None, None, True, None, 0)


def test_mask_descriptor_dense_equals_reference_mask():
    from vyomai_amd.layers.mask import AttnMask
    for B, L, start in ((2, 16, 0), (3, 5, 11)):
        kp = cases.keypad(B, L + start)
        m = AttnMask.from_padding(torch.from_numpy(kp), causal=True, start_pos=start, query_len=L)
        want = cases.causal_additive(B, L, start, kp)
        assert np.array_equal(m.dense(torch.float32).numpy(), want)
    m = AttnMask.from_padding(None, causal=True, start_pos=0, query_len=4)
    assert np.array_equal(m.dense(torch.float32, batch=2).numpy(), cases.causal_additive(2, 4, 0, None))
    with pytest.raises(ValueError):
        AttnMask.from_padding(torch.ones(2, 7), causal=True, start_pos=0, query_len=5)


def test_kv_caches_bookkeeping_on_cpu():
    """update()/get()/len semantics of the four cache classes (pure tensor bookkeeping)."""
    import vyomai_amd as V
    cfg = cases.with_kv(cases.micro_cfg(), None)
    cfg.num_hidden_layers = 2
    B, h, dh = 2, cfg.num_attention_heads, 16
    k1, v1 = torch.randn(B, h, 5, dh), torch.randn(B, h, 5, dh)
    k2, v2 = torch.randn(B, h, 1, dh), torch.randn(B, h, 1, dh)
    dyn = V.DynamicCacheOne(cfg)
    assert len(dyn) == 0
    with pytest.raises(ValueError):
        dyn.get(0)
    ka, va = dyn.update(0, k1, v1, 0)
    assert torch.equal(ka, k1) and len(dyn) == 5
    ka, va = dyn.update(0, k2, v2, 5)
    assert torch.equal(ka, torch.cat([k1, k2], 2)) and torch.equal(va, torch.cat([v1, v2], 2))
    assert dyn.get_seq_length(0) == 6 and dyn.get_seq_length(1) == 0
    for _ in range(80):  # capacity doubling keeps earlier entries
        ka, _ = dyn.update(0, k2, v2, 0)
    assert ka.shape[2] == 86 and torch.equal(ka[:, :, :5], k1)
    st = V.StaticCacheOne(cfg, max_cache_len=8, batch_size=B)
    st.device = torch.device("cpu")
    st.key_cache = [t.cpu() for t in st.key_cache]
    st.value_cache = [t.cpu() for t in st.value_cache]
    ka, va = st.update(1, k1, v1, 0)
    assert ka.shape == (B, h, 5, dh) and torch.equal(ka, k1)
    ka, va = st.update(1, k2, v2, 5)
    assert ka.shape[2] == 6 and torch.equal(ka[:, :, 5:], k2)
    with pytest.raises(ValueError):
        st.update(1, torch.randn(B, h, 4, dh), torch.randn(B, h, 4, dh), 6)
    one = V.StaticCache(cfg)
    with pytest.raises(AssertionError):
        one.update(k1, v1, 0)  # batch 1 only, like the reference (:137)
    ka, _ = one.update(k1[:1], v1[:1], 0)
    assert ka.shape == (1, h, 5, dh) and len(one) == cfg.max_position_embeddings
    d1 = V.DynamicCache(cfg)
    d1.update(k1, v1)
    ka, _ = d1.update(k2, v2)
    assert len(d1) == 6 and torch.equal(d1.get()[0], ka)


def test_flat_arena_packs_qkv_adjacent():
    import vyomai_amd as V
    from vyomai_amd.training import FlatArena
    cfg = cases.with_kv(cases.micro_cfg(), "gqa")
    m = V.DecoderModel(cfg, "rope", "gqa")
    before = {k: v.clone() for k, v in m.state_dict().items()}
    arena = FlatArena(m, shadow_dtype=None)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    att = m.all_layer[0].attention
    w, b = att._packed()
    assert w.shape == (64 + 2 * 32, 64) and w.data_ptr() == att.query.weight.data_ptr()
    assert torch.equal(w[64:96], att.key.weight) and torch.equal(b[96:], att.value.bias)
    assert att.query.weight.grad.data_ptr() == arena.grad.data_ptr() + 4 * arena.offsets[
        [id(p) for p in arena.params].index(id(att.query.weight))]


def test_recipe_is_deterministic_and_fp32_exact():
    from vyomai_amd import recipe
    a = recipe.uniform("x", (3, 5), 0.25)
    b = recipe.uniform("x", (3, 5), 0.25)
    assert a.dtype == np.float32 and np.array_equal(a, b) and np.abs(a).max() <= 0.25
    assert not np.array_equal(a, recipe.uniform("y", (3, 5), 0.25))
    ids = recipe.token_ids("t", (4, 9), 3, 50)
    assert ids.min() >= 3 and ids.max() < 50
    assert np.array_equal(recipe.param_value("lm_head.decoder.bias", (7,)), recipe.param_value("lm_head.bias", (7,)))


def test_sampling_front_end_host_logic():
    """Processors, cache trimming and the residual normalisation of speculative decoding: host behaviour that needs
    no GPU -- and no CPU fallback for what does."""
    import ctypes as C
    from vyomai_amd import _lib, logits_processors as LP
    from vyomai_amd.layers.kv_cache import DynamicCacheOne, StaticCacheOne
    from vyomai_amd.speculative_decoding import norm_fn, trim_cache
    with pytest.raises(TypeError):
        LP.LogitsProcessor(1.0)                       # abstract in the reference (abc.ABC)
    p = LP.TopKNucleusProcessor(0.9, 40, 0.8)
    assert (p.temperature, p.top_k, p.top_p, p.stochastic) == (0.9, 40, 0.8, True)
    assert isinstance(p, LP.MultinomialProcessor) and not LP.GreedyProcessor().stochastic
    assert LP.GreedyProcessor().temperature == 1 and not hasattr(LP.GreedyProcessor(), "top_k")
    with pytest.raises(_lib.VyomHipError, match="MI355X"):
        LP.NucleusProcessor(0.2, 0.9)(torch.zeros(2, 16))   # CPU logits: no fallback
    probs = torch.tensor([[0.1, 0.7, 0.2]])
    assert LP.GreedyProcessor().sample(probs).tolist() == [[1]]
    assert LP.MultinomialProcessor(1.0).sample(torch.tensor([[0.0, 1.0, 0.0]])).tolist() == [[1]]
    # trim_cache (reference speculative_decoding.py:9-71): dynamic caches shrink, unknown types are refused
    cfg = cases.micro_cfg()
    cache = DynamicCacheOne(cfg)
    k = torch.randn(1, 2, 6, 16)
    cache.update(0, k, k.clone())
    assert len(trim_cache(cache, 2)) == 4 and trim_cache(cache, 0) is cache and trim_cache(None, 3) is None
    k2, _ = cache.update(0, k[:, :, :1], k[:, :, :1].clone())
    assert k2.shape[2] == 5 and torch.equal(k2[:, :, 4], k[:, :, 0])
    assert trim_cache(StaticCacheOne(cfg, max_cache_len=8, batch_size=1), 3) is not None
    with pytest.raises(ValueError, match="Unsupported cache type"):
        trim_cache([], 1)
    x = torch.tensor([[0.5, -1.0, 1.5, float("nan")]])
    assert torch.equal(norm_fn(x), torch.tensor([[0.25, 0.0, 0.75, 0.0]]))
    # argument validation of the new entry points happens before any launch
    buf = (C.c_char * 4096)()
    a = (C.addressof(buf) + 15) // 16 * 16
    with pytest.raises(_lib.VyomHipError, match="temperature"):
        _lib.call("vy_sampling_probs", a, 64, 1, 8, 0, 0.0, 0, 0.0, a, 64, None)
    with pytest.raises(_lib.VyomHipError, match="1..8 descriptors"):
        _lib.call("vy_linear_wgrad_grouped", a, 0, 1, None)
    with pytest.raises(_lib.VyomHipError, match="bad arguments"):
        _lib.call("vy_greedy_step", a, 8, 1, 8, 0, a, 4, 9, None, 0, None, 0, a, None, None)   # cur_pos beyond the row


def test_deferred_weight_gradient_bookkeeping():
    """autograd_train._WgradGroup decides from shapes alone which weight gradients are deferred into a grouped
    launch, and marks them so that the reducer ignores autograd's premature hook (training.BucketReducer._hook)."""
    from vyomai_amd import autograd_train as AT
    g = AT._WgradGroup()
    w = torch.empty(768, 768)
    assert g.wants(torch.empty(32, 512, 768), w, None)                 # 16384 rows
    assert g.wants(torch.empty(2112, 768), w, None)                    # a captioning decoder's rows
    assert not g.wants(torch.empty(1024, 768), w, None)                # too few rows for 256 x 256 tiles to pay
    assert not g.wants(torch.empty(16384, 768), w, torch.ones(1))      # device-scalar alpha: the LM head's own path
    assert not g.wants(torch.empty(16384, 50265), torch.empty(50265, 768), None)   # the vocabulary has its own variant
    assert not g.wants(torch.empty(16384, 64), torch.empty(64, 64), None)
    from vyomai_amd.training import BucketReducer
    seen = []

    class Fake:
        mark_ready = lambda self, p: seen.append(p)
    p = torch.nn.Parameter(torch.zeros(4))
    p._vy_deferred = True
    BucketReducer._hook(Fake(), p)
    assert not seen
    p._vy_deferred = False
    BucketReducer._hook(Fake(), p)
    assert seen == [p]


def test_bench_starts_its_own_ranks_and_never_misreports_n_gpus():
    """`python bench.py --gpus N` without a launcher used to run ONE rank and print n_gpus: 1.  Now it starts
    the N ranks itself (a child torch.distributed.run), and refuses a world size that differs from --gpus."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--rehearse-launch", "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True,
                         timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    env["WORLD_SIZE"] = "4"
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-launch"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and not bad.stdout.strip()


def test_wrapped_projection_is_refused_not_ignored():
    """The reference's adapters replace attention.query by a wrapper holding the Linear at `.linear`
    (layers/adapters.py:21-24): the fused QKV path must say that it cannot honour it, not silently drop it."""
    import torch.nn as nn
    import vyomai_amd as V
    from vyomai_amd._lib import VyomHipError

    class LoraLike(nn.Module):
        def __init__(self, linear):
            super().__init__()
            self.linear = linear
            self.lora_a = nn.Parameter(torch.zeros(4, linear.in_features))
            self.lora_b = nn.Parameter(torch.zeros(linear.out_features, 4))

    cfg = cases.micro_cfg()
    layer = V.DecoderModel(cfg, "rope", None).all_layer[0]
    layer.attention._packed()                       # plain projections: fine
    layer.attention.query = LoraLike(layer.attention.query)
    with pytest.raises(VyomHipError, match="LoraLike"):
        layer.attention._packed()
    with pytest.raises(VyomHipError, match="merge the low-rank update"):
        layer.attention._params()
