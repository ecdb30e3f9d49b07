"""Two-lane training forward (ops.lanes: two batch halves on two HIP streams writing the same full-batch tensors)
against the one-stream forward: the same kernels on row halves, so loss and every gradient must be bit-identical."""
import os

import pytest
import torch

from tests.golden import cases

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _grads(model, ids, labels, am, lanes):
    os.environ["VY_LANES"] = lanes
    try:
        model.zero_grad(set_to_none=True)
        loss = model.clm_loss(ids, labels, am)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    finally:
        os.environ.pop("VY_LANES", None)


@pytest.mark.parametrize("attn", [None, "gqa"])
@pytest.mark.parametrize("B,L,pad", [(4, 64, False), (3, 48, True), (2, 128, True)])
def test_two_lane_forward_is_bit_identical(attn, B, L, pad):
    import vyomai_amd as V
    from vyomai_amd import recipe
    cfg = cases.micro_cfg()
    cfg.hidden_dropout_prob = 0.0
    cfg.max_position_embeddings = 256
    torch.manual_seed(0)
    m = V.DecoderModel(cfg, "rope", attn)
    recipe.load_recipe_(m)
    m = m.to(DEV).to(torch.bfloat16).train()
    ids = torch.randint(3, cfg.vocab_size, (B, L), device=DEV)
    am = None
    labels = ids.clone()
    if pad:
        am = torch.ones(B, L, dtype=torch.long, device=DEV)
        am[0, L - 7:] = 0
        am[B - 1, L - 19:] = 0
        labels[am == 0] = -100
    # the forward itself: bit for bit
    outs = []
    for lanes in ("0", "1"):
        os.environ["VY_LANES"] = lanes
        try:
            o = m(ids, am)
            outs.append((o.hidden_state.detach().clone(), o.logits.detach().clone()))
        finally:
            os.environ.pop("VY_LANES", None)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # loss and gradients: the loss sum and the weight gradients are accumulated with fp32 atomics (order differs from run
    # to run in either mode), so equal to fp32 rounding of the sums
    l0, g0 = _grads(m, ids, labels, am, "0")
    l1, g1 = _grads(m, ids, labels, am, "1")
    assert abs(float(l0) - float(l1)) < 2e-6 * abs(float(l0))
    assert g0.keys() == g1.keys() and len(g0) > 10
    for n in g0:
        a, b = g0[n].float(), g1[n].float()
        assert torch.allclose(a, b, rtol=0, atol=1e-2 * float(a.abs().max()) + 1e-12), n


def test_lanes_region_joins_before_whole_ops_and_keeps_tensors_alive():
    """A region entered without autograd (nothing saves the intermediates): outputs equal the one-stream run, also when
    the caller drops intermediates at once and a non-split op follows a split one."""
    from vyomai_amd import ops
    torch.manual_seed(1)
    B, L, d = 4, 96, 256
    x = torch.randn(B, L, d, device=DEV).bfloat16()
    w = (torch.randn(d, d, device=DEV) / 16).bfloat16()
    b = torch.randn(d, device=DEV).bfloat16()
    g, be = torch.randn(d, device=DEV).bfloat16(), torch.randn(d, device=DEV).bfloat16()

    def chain():
        y = x
        for _ in range(6):
            s = ops.linear(y, w, b, residual=y)
            y, _, _ = ops.layernorm(s, g, be, 1e-5)
            del s
        z = ops.rmsnorm(y, g, 1e-6)          # not lane-aware: the main stream must have joined lane 1
        return ops.linear(z, w, b)

    ref = chain()
    with ops.lanes(B, L, True):
        out = chain()
    torch.cuda.synchronize()
    # (outside the region these 384-row GEMMs split K over workgroups, inside they do not: equal up to the fp32 summation order)
    assert (ref.float() - out.float()).abs().max() <= 2 ** -5 * max(1.0, float(ref.float().abs().max()))
    assert not ops._Lanes.active and ops._Lanes.keep == []
