"""BASELINE.json configs[4]: PaliGemma-shape model (SigLIP-So400m 27L d=1152 + Gemma-2B 18L d=2048, MQA dh=256,
vocabulary 257216), random-init bf16, one 224x224 image + 8 text tokens, KV-cache greedy decode of 64 tokens
on one MI355X (Examples/paligemma.ipynb cell 30: max_cache_len 384).  Reports prefill and decode rates."""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(new_tokens: int = 64, dev=None):
    from vyomai_amd import shapes as cases
    from vyomai_amd.models import paligemma as P
    dev = dev or torch.device("cuda", 0)
    torch.manual_seed(0)
    vis = P.SiglipVisionConfig(**cases.SIGLIP)
    txt = types.SimpleNamespace(**cases.GEMMA)
    with torch.device(dev):
        m = P.PaliGemmaForConditionalGeneration(P.PaliGemmaShape(vis, txt, txt.hidden_size))
    m = m.to(torch.bfloat16).eval()
    for p in m.parameters():
        if p.dim() > 1:
            torch.nn.init.normal_(p, std=0.02)
    nparam = sum(p.numel() for p in m.parameters())
    img = torch.rand(1, 3, 224, 224, device=dev)
    ids = torch.randint(3, txt.vocab_size, (1, 8), device=dev)

    def go(n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        m.generate(img, ids, max_new_tokens=n, max_cache_len=384)
        torch.cuda.synchronize()
        return time.perf_counter() - t

    go(4)
    t1 = go(1)
    tn = go(new_tokens)
    per = (tn - t1) / (new_tokens - 1)
    streamed = (nparam - sum(p.numel() for p in m.vision_tower.parameters())) * 2
    return {"workload": "configs[4]: PaliGemma shape (SigLIP-So400m 27L + Gemma-2B 18L, MQA dh=256, vocab 257216), random init, "
                        "bf16, B=1, 224x224 image + 8 text tokens, KV-cache greedy decode",
            "parameters": nparam, "prefill_plus_first_token_ms": round(t1 * 1e3, 2), "ms_per_token": round(per * 1e3, 3),
            "decode_tokens_per_sec": round(1 / per, 1), "new_tokens": new_tokens,
            "weight_stream_TBps": round(streamed / per * 1e-12, 2)}


if __name__ == "__main__":
    r = run()
    print(f"PaliGemma shape: {r['parameters']/1e9:.2f} G parameters; vision tower + 264-token prefill + first token "
          f"{r['prefill_plus_first_token_ms']:.1f} ms; decode {r['ms_per_token']:.3f} ms/token = "
          f"{r['decode_tokens_per_sec']:.1f} tokens/s (B=1); weight stream per token {r['weight_stream_TBps']:.2f} TB/s")
