"""configs[3]: ViT-B/16 + vision-language decoder, caption training step through FlatTrainer (B=64 images of
224x224, 32 caption tokens, 12-layer text decoder, bf16 kernels + fp32 masters, AdamW)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def step_flops(B: int, d: int = 768, vit_tokens: int = 197, cap_tokens: int = 33, layers: int = 12, vocab: int = 50265):
    """Algorithmic FLOPs of one caption-training step (forward + backward = 3 x forward; the patch projection has no
    input gradient: 2 x): per token and layer 24 d^2 of projections / feed-forward + attention counted as executed --
    4 S d in the ViT (every key visible), 2 S d in the causal decoder (BASELINE.md section 4's convention)."""
    mv, md = B * vit_tokens, B * cap_tokens
    vit = layers * mv * (24 * d * d + 4 * vit_tokens * d)
    dec = layers * md * (24 * d * d + 2 * cap_tokens * d) + md * (2 * d * d + 2 * d * vocab)
    patch = B * (vit_tokens - 1) * 2 * d * d
    return 3.0 * (vit + dec) + 2.0 * patch


def run(B: int = 64, steps: int = 8, dev: str = "cuda"):
    """One process per GPU: under torch.distributed every rank runs B images per step (weak scaling), gradients go
    through the trainer's bucketed all-reduce; the time is the maximum over the ranks, the rates are whole-job."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    import vyomai_amd as V
    from vyomai_amd import recipe
    from vyomai_amd.training import FlatTrainer
    from vyomai_amd.shapes import vit_b16_config
    vcfg = vit_b16_config(12, 0.0)       # ViT-B/16: 12 layers, d = 768, 12 heads, MLP 3072, 197 tokens
    cfg = V.EncoderConfig(num_hidden_layers=12, hidden_dropout_prob=0.0)
    vlm = V.VisionLanguageModel(cfg, V.Vit(vcfg), "rope", None)
    recipe.load_recipe_(vlm)
    vlm = vlm.to(dev).train()
    img = torch.from_numpy(recipe.uniform(f"vit.img.{rank}", (B, 3, 224, 224), 0.5, 0.5)).to(dev).to(torch.bfloat16)
    ids = torch.from_numpy(recipe.token_ids(f"cap.ids.{rank}", (B, 32), 3, cfg.vocab_size)).to(dev)
    tr = FlatTrainer(vlm, lr=1e-4)

    def loss_fn():
        if os.environ.get("VLM_TORCH_LOSS"):
            logits = vlm(pixel_values=img, decoder_input_ids=ids).logits          # (B, 33, V): image token first
            lg = logits[:, 1:-1].float().reshape(-1, logits.shape[-1])
            return torch.nn.functional.cross_entropy(lg, ids[:, 1:].reshape(-1))
        return vlm.caption_loss(img, ids)     # the same loss with the head and the cross-entropy fused

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # (no host read of the loss inside the loop: the next step's launches are queued while this one runs, as in the
    # headline loop -- a per-step .item() left the GPU idle while the host queued ~600 launches)
    losses = []
    warm = 3
    for s in range(warm + steps):
        if s == warm:
            barrier(); t0 = time.perf_counter()
        losses.append(tr.train_step(loss_fn))
    barrier()
    dt = (time.perf_counter() - t0) / steps
    losses = [float(x.item()) for x in losses]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return {"workload": "configs[3]: ViT-B/16 (12L, 224x224/16, 197 tokens) + 12L d=768 vision-language decoder (33 tokens), caption training, AdamW, bf16 kernels + fp32 masters",
            "batch_per_gpu": B, "n_gpus": world, "ms_per_step": round(dt * 1e3, 2), "images_per_sec": round(world * B / dt, 1),
            "decoder_tokens_per_sec": round(world * B * 33 / dt, 1), "scaling": "weak", "steps": steps, "warmup": warm,
            "roofline": {"bound": "mfma", "achieved": round(world * step_flops(B) / dt * 1e-12, 1), "peak": 2500.0 * world,
                         "unit": "TFLOP/s", "frac": round(step_flops(B) / dt * 1e-12 / 2500.0, 4),
                         "flops_per_step_per_gpu": step_flops(B),
                         "note": "whole training step (forward + backward + AdamW) over the algorithmic FLOPs of the model; "
                                 "kernel profile: profiles/r03_vlm_kernel_stats.csv"},
            "losses": [round(x, 4) for x in losses]}


if __name__ == "__main__":
    r = run(int(os.environ.get("B", "64")))
    print("losses", r["losses"])
    print(f"caption training B={r['batch_per_gpu']}: {r['ms_per_step']:.1f} ms/step, {r['images_per_sec']:.0f} images/s, "
          f"{r['decoder_tokens_per_sec']:.0f} decoder tokens/s")
