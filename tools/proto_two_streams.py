"""Prototype: one training step as TWO half-batches on two streams, interleaved layer by layer (forward creation order
alternates, so autograd's backward alternates too), against the ordinary step.  Measures only; the reducer / optimizer
overlap are switched off in both arms."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VY_WGRAD_GROUP", os.environ.get("GROUP", "0"))   # grouped wgrads mix tensors of both streams: off here
import vyomai_amd as V
from vyomai_amd import recipe, autograd_train as AT
from vyomai_amd.training import FlatTrainer

cfg = V.EncoderConfig(num_hidden_layers=12, max_position_embeddings=1024, hidden_dropout_prob=0.0)
m = V.DecoderModel(cfg, "rope", None)
recipe.load_recipe_(m)
m = m.to("cuda").train()
tr = FlatTrainer(m, lr=5e-5, weight_decay=0.01, overlap_optimizer=False)
torch.manual_seed(1234)
ids = torch.randint(3, cfg.vocab_size, (32, 512), device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def fwd_split(parts):
    main = torch.cuda.current_stream()
    streams = [s1, s2][:len(parts)]
    hs, fr, mk = [], [], []
    for st, x in zip(streams, parts):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            h = m._embed(m.word_embeddings, x)
            h, f = m._positions(h, 0, x.shape[1])
            hs.append(h); fr.append(f); mk.append(m.create_mask_for_decoder(input_ids=x, attention_mask=None, start_pos=0))
    for layer in m.all_layer:
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                hs[i], _ = layer(hs[i], mk[i], freqs=fr[i], use_cache=False, kv_cache=None, start_pos=0)
    losses = []
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            losses.append(m.lm_head.loss(hs[i], parts[i], -100))
    for st in streams:
        main.wait_stream(st)
    return sum(losses) / len(losses)


def step(split):
    tr.zero_grad()
    AT._WT.refresh()
    if split:
        loss = fwd_split([ids[:16], ids[16:]])
    else:
        loss = m.clm_loss(ids, ids)
    tr.reducer.enabled = False
    loss.backward()
    tr.reducer.enabled = True
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    return loss


for split in (False, True, False, True):
    for _ in range(3):
        l = step(split)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        l = step(split)
    torch.cuda.synchronize()
    print(f"split={split}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per fwd+bwd (no optimizer), loss {l.item():.4f}")
