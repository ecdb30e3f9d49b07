"""smoke(): one small decoder forward + cached greedy decode on cuda:0 through the HIP path,
checked against the CPU oracle (checker only) on the same recipe weights."""
import numpy as np
import torch

from oracle import vyom_oracle as O
from tests.golden import cases
from vyomai_amd import recipe


def run(verbose: bool = False) -> None:
    import vyomai_amd as V
    cfg = cases.test_cfg()
    cfg.num_hidden_layers = 2
    cfg.vocab_size = 1031
    m = V.DecoderModel(cfg, "rope", None)
    recipe.load_recipe_(m)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to("cuda").eval()
    ids = torch.from_numpy(recipe.token_ids("smoke.ids", (2, 24), 3, cfg.vocab_size))
    am = torch.ones(2, 24, dtype=torch.long)
    with torch.no_grad():
        out = m(ids.cuda(), am.cuda())
        toks = m.generate(ids.cuda(), am.cuda(), max_len=4, use_cache=True, use_static_cache=True)
    torch.cuda.synchronize()
    c = O.Cfg.of(cfg)
    ref = O.decoder_forward(sd, c, ids, am, "rope", None)
    err = (out.hidden_state.cpu() - ref.hidden_state).abs().max().item()
    ref_toks = O.decoder_generate(sd, c, ids, am.float(), 4, "rope", None, use_cache=True, use_static_cache=True)
    if verbose:
        print(f"smoke: hidden max abs err vs oracle {err:.2e}; tokens equal: {torch.equal(toks.cpu(), ref_toks)}")
    assert err < 1e-5, err
    assert torch.equal(toks.cpu(), ref_toks)
    # bf16 path runs and stays finite
    mb = m.to(torch.bfloat16)
    with torch.no_grad():
        ob = mb(ids.cuda(), am.cuda())
    assert torch.isfinite(ob.logits.float()).all()
    if verbose:
        print("smoke: bf16 path max |hidden - fp32 oracle| =",
              f"{(ob.hidden_state.float().cpu() - ref.hidden_state).abs().max().item():.2e}")
