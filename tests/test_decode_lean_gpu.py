"""The decode-only kernels (vyomai_amd/csrc/vy_decode.hip) behind vy_decoder_step at the benchmark width
(d = 768, 12 heads of 64; reference models/decoder.py:430-514 with a StaticCacheOne):

* against the general launchers (vy_debug_set_decode_lean(0)) on the same plan: same arithmetic per output up
  to the fp32 summation order of the split-K partial tiles -- logits, hidden state and the K/V rows written
  agree to bf16 rounding;
* against the fp32 path (pinned to the reference at 1e-5 in test_models_gpu.py) on the same weights: the bf16
  error is no larger than the general bf16 kernels';
* hipGraph replay (device-side position) gives the same tokens as eager launches.
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(attn, pos_kind, layers=2, dtype=torch.bfloat16):
    import vyomai_amd as V
    from vyomai_amd import recipe
    cfg = V.EncoderConfig(num_hidden_layers=layers, max_position_embeddings=256, hidden_dropout_prob=0.0)
    if attn == "gqa":
        cfg.num_key_value_heads = 4
    m = V.DecoderModel(cfg, pos_kind, None if attn == "none" else "gqa")
    recipe.load_recipe_(m)
    return cfg, m.to(DEV).to(dtype).eval()


def _plan(cfg, m, B, dtype, cap=96, seed=0):
    from vyomai_amd.decode_plan import DecodePlan
    from vyomai_amd.layers.kv_cache import StaticCacheOne
    cache = StaticCacheOne(cfg, max_cache_len=cap, batch_size=B, dtype=dtype)
    plan = DecodePlan(m, cache, B, dtype, torch.device("cuda", 0))
    g = torch.Generator().manual_seed(seed)
    for i in range(len(cache.key_cache)):
        cache.key_cache[i].copy_(torch.randn(cache.key_cache[i].shape, generator=g).to(dtype))
        cache.value_cache[i].copy_(torch.randn(cache.value_cache[i].shape, generator=g).to(dtype))
    return plan, cache


def _set_lean(v):
    from vyomai_amd import _lib
    lib = _lib.load()
    lib.vy_debug_set_decode_lean.argtypes = [C.c_int]
    lib.vy_debug_set_decode_lean(int(v))


def _step(plan, cache, x, pos, lean):
    _set_lean(lean)
    try:
        logits, hidden = plan.step(x, pos, want_hidden=True)
        torch.cuda.synchronize()
        kv = [(k[:, :, pos].float().clone(), v[:, :, pos].float().clone())
              for k, v in zip(cache.key_cache, cache.value_cache)]
        return logits.float().clone(), hidden.float().clone(), kv
    finally:
        _set_lean(1)


@pytest.mark.parametrize("attn,pos_kind,B", [("none", "rope", 32), ("none", "rope", 7), ("gqa", "rope", 17),
                                             ("none", "absolute", 32)])
def test_lean_step_matches_general_and_fp32(attn, pos_kind, B):
    torch.manual_seed(1)
    cfg, m = _model(attn, pos_kind)
    x = (torch.randn(B, cfg.hidden_size) * 0.5).to(torch.bfloat16).to(DEV)
    pos = 37
    plan, cache = _plan(cfg, m, B, torch.bfloat16)
    lg_lean, hd_lean, kv_lean = _step(plan, cache, x, pos, 1)
    plan2, cache2 = _plan(cfg, m, B, torch.bfloat16)
    lg_gen, hd_gen, kv_gen = _step(plan2, cache2, x, pos, 0)
    # same inputs, same rounding points: differences are fp32 summation order seen through bf16 rounding
    for (k1, v1), (k2, v2) in zip(kv_lean, kv_gen):
        assert (k1 - k2).abs().max() <= 4e-2 and (v1 - v2).abs().max() <= 4e-2
    assert (k1 - k2).abs().mean() <= 1e-3
    assert (hd_lean - hd_gen).abs().mean() <= 5e-3, (hd_lean - hd_gen).abs().mean()
    assert (lg_lean - lg_gen).abs().mean() <= 2e-2, (lg_lean - lg_gen).abs().mean()
    # fp32 path on the same weights and cache contents
    cfg32, m32 = _model(attn, pos_kind, dtype=torch.float32)
    plan32, cache32 = _plan(cfg32, m32, B, torch.float32)
    lg32, hd32, _ = _step(plan32, cache32, x.float(), pos, 1)
    e_lean = (hd_lean - hd32).abs().mean().item()
    e_gen = (hd_gen - hd32).abs().mean().item()
    assert e_lean <= 1.25 * e_gen + 1e-3, (e_lean, e_gen)
    el, eg = (lg_lean - lg32).abs().mean().item(), (lg_gen - lg32).abs().mean().item()
    assert el <= 1.25 * eg + 2e-3, (el, eg)


def test_lean_graph_replay_matches_eager(monkeypatch):
    from vyomai_amd import recipe
    cfg, m = _model("none", "rope")
    ids = torch.from_numpy(recipe.token_ids("lean.ids", (5, 20), 3, cfg.vocab_size)).to(DEV)
    am = torch.ones_like(ids)
    monkeypatch.setenv("VY_DECODE_GRAPH", "1")
    t_graph = m.generate(ids, am, max_len=10, use_cache=True, use_static_cache=True)
    monkeypatch.setenv("VY_DECODE_GRAPH", "0")
    t_eager = m.generate(ids, am, max_len=10, use_cache=True, use_static_cache=True)
    assert torch.equal(t_graph, t_eager)
