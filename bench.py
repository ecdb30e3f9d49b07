"""Benchmark of the VyomAI transformer hot path on MI355X (driver contract: see the task brief).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* = one CLM training step (forward + backward + gradient all-reduce + fused AdamW) of the
GPT-style decoder of BASELINE.json configs[1]: 12 layers, d=768, 12 heads, RoPE, vocab 50265,
per-GPU batch 32 x seq 512, bf16 kernels with fp32 master weights, synthetic tokens, recipe
weights.  `value` = whole-job training tokens/s (weak scaling: per-GPU work fixed).  After the
timed training region the same run measures KV-cache greedy decode (512-token prompts, 128 new
tokens, static cache) and reports it as `decode_tokens_per_sec`.

Extra objects on the JSON line:
  roofline      dominant kernel = the bf16 MFMA GEMM (forward projections/FFN of one step); achieved
                = algorithmic FLOPs / HIP-event time of those launches, measured live on the launch
                stream; peak = 2500 TFLOP/s dense bf16 (MI355X_MICROARCH.md).  `block_forward`
                reports the attention+FFN block forward (the north-star 40 % target) the same way.
  cpu_baseline  the CPU oracle (plain-torch restatement of the reference, pinned to reference-made
                golden vectors) timed on the host cores on a bounded sample (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import contextlib
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--decode-tokens", type=int, default=128)
    ap.add_argument("--attn", default="none", choices=["none", "gqa"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the configs[3] / configs[4] side measurements")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="launcher rehearsal (runs without a GPU): every rank joins the process group, meets the barrier and "
                         "rank 0 prints a line with n_gpus = the world size -- no kernels, not a benchmark line")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal on a one-GPU box: a ONE-rank process group whose bucket collectives are really issued "
                         "(RCCL init, async all-reduce, stream waits, barrier); not a benchmark configuration")
    return ap.parse_args()


def make_cfg(a):
    from vyomai_amd import EncoderConfig
    cfg = EncoderConfig(num_hidden_layers=a.layers, max_position_embeddings=1024, hidden_dropout_prob=0.0)
    if a.attn == "gqa":
        cfg.num_key_value_heads = 4
    return cfg


def block_flops_per_token(d, S):
    """forward FLOPs of one attention+FFN block per token, causal attention counted as 2*S*d
    (BASELINE.md section 4)."""
    return 24 * d * d + 2 * S * d


def event_time_us(fn, iters, stream):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    s.record(stream)
    for _ in range(iters):
        fn()
    e.record(stream)
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def roofline_probe(cfg, B, L, dev):
    """Time the dominant kernel and the block forward on the stream they are launched on."""
    from vyomai_amd import ops
    d, h = cfg.hidden_size, cfg.num_attention_heads
    dh = d // h
    M = B * L
    bf = torch.bfloat16
    st = torch.cuda.current_stream()
    g = torch.Generator().manual_seed(7)
    r = lambda *s: torch.randn(*s, generator=g).to(bf).to(dev)
    x = r(M, d)
    x3 = x.view(B, L, d)
    wqkv, bqkv = r(3 * d, d) / math.sqrt(d), r(3 * d)
    wo, bo = r(d, d) / math.sqrt(d), r(d)
    w1, b1 = r(4 * d, d) / math.sqrt(d), r(4 * d)
    w2, b2 = r(d, 4 * d) / math.sqrt(4 * d), r(d)
    gam, bet = r(d), r(d)
    cos, sin = ops.rope_tables(dh, 1024, dev)
    q = torch.empty(B, h, L, dh, dtype=bf, device=dev)
    k, v = torch.empty_like(q), torch.empty_like(q)
    o = torch.empty(B, L, d, dtype=bf, device=dev)
    s1 = torch.empty(M, d, dtype=bf, device=dev)
    hm = torch.empty(M, 4 * d, dtype=bf, device=dev)

    def gemms():
        ops.qkv_rope(x3, wqkv, bqkv, h, h, dh, cos, sin, 0, q, k, v)
        ops.linear(x, wo, bo, residual=x, out=s1)
        ops.linear(x, w1, b1, act=1, out=hm)
        ops.linear(hm, w2, b2, residual=x, out=s1)

    def block():
        ops.qkv_rope(x3, wqkv, bqkv, h, h, dh, cos, sin, 0, q, k, v)
        ops.attention(q, k, v, causal=True, out=o)
        ops.linear(o.view(M, d), wo, bo, residual=x, out=s1)
        ops.layernorm(s1, gam, bet, 1e-5)
        ops.linear(s1, w1, b1, act=1, out=hm)
        ops.linear(hm, w2, b2, residual=x, out=s1)
        ops.layernorm(s1, gam, bet, 1e-5)

    t_gemm = event_time_us(gemms, 10, st)
    t_blk = event_time_us(block, 10, st)
    f_gemm = 24.0 * d * d * M
    f_blk = block_flops_per_token(d, L) * M
    # HBM-side bytes per launch come from a rocprofv3 --pmc run of the same four launches (rocprofv3 cannot wrap
    # this process from the inside): tools/regen_profiles.sh writes profiles/rNN_gemm_pmc.json every round; the
    # newest one is reported, with its name
    traffic, traffic_file, alg_bytes = None, None, None
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_pmc.json")), reverse=True):
        try:
            with open(fn) as f:
                j = json.load(f)
            traffic = round(j["avg_hbm_bytes_per_launch"])
            alg_bytes = round(j.get("algorithmic_bytes_per_launch", 0)) or None
            traffic_file = os.path.relpath(fn, ROOT)
            break
        except (OSError, KeyError, ValueError):
            continue
    return {
        "bound": "mfma", "kernel": "gemm_nt_bf16_x3m16_kernel / gemm_nt_bf16_m16_kernel<256,256> (4 launches of one layer: qkv+rope, out+res, "
                                   "ffn1+gelu, ffn2+res)",
        "achieved": round(f_gemm / t_gemm * 1e-6, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
        "frac": round(f_gemm / t_gemm * 1e-6 / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
        "traffic_note": f"HBM bytes/launch from rocprofv3 PMC (2*FETCH_SIZE+WRITE_SIZE), {traffic_file}; "
                        f"algorithmic bytes/launch {alg_bytes}",
        "measured_ceilings": measured_ceilings(dev),
        "avg_launch_us": round(t_gemm / 4, 1), "flops_per_launch": f_gemm / 4,
        "block_forward": {"us": round(t_blk, 1), "achieved": round(f_blk / t_blk * 1e-6, 1),
                          "frac": round(f_blk / t_blk * 1e-6 / PEAK_BF16_TFLOPS, 4),
                          "flops": f_blk, "convention": "24 d^2 + 2 S d per token (causal)"},
    }


def measured_ceilings(dev):
    """What this device delivers on the two bounding resources, measured in the same run next to the spec peaks
    (SURVEY section 8d): the matrix pipe alone (six independent MFMA chains per wave, 8 waves per CU, non-trivial
    operands, no memory traffic) and a plain 16-byte-per-lane copy of 1 GiB (read + write bytes counted)."""
    import ctypes as C
    from vyomai_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream()
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    wgs, iters = 256 * 4, 4000
    lib.vy_debug_mfma_peak.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.vy_debug_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    t = event_time_us(lambda: lib.vy_debug_mfma_peak(wgs, iters, sink.data_ptr(), st.cuda_stream), 3, st)
    mfma = wgs * 8 * iters * 6 * 2.0 * 32 * 32 * 16 / t * 1e-6
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    a.zero_()
    t = event_time_us(lambda: lib.vy_debug_copy(a.data_ptr(), b.data_ptr(), n, st.cuda_stream), 5, st)
    del a, b
    return {"mfma_bf16_TFLOPs": round(mfma, 1), "mfma_spec_TFLOPs": PEAK_BF16_TFLOPS,
            "hbm_copy_GBs": round(2.0 * n / t * 1e-3, 1), "hbm_spec_GBs": PEAK_HBM_GBS,
            "note": "MFMA-only loop (6 chains x 8 waves per CU) and a 1 GiB copy (read + write bytes), this run"}


def host_cores() -> int:
    """CPU threads this process may really use: affinity mask and cgroup quota, not the node size."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(cfg, seq):
    """The oracle (CPU restatement of the reference, pinned to reference-made golden vectors) timed on the host
    cores, the two legs SURVEY section 8d / BASELINE.md section 5 name: forward+backward of the 12-layer decoder at
    B=4 x seq (training tokens/s) and a B=1 cached greedy decode of 32 tokens behind a 64-token prompt (decode
    tokens/s).  Bounded: at most ~25 s of CPU work in all."""
    from oracle import vyom_oracle as O
    from vyomai_amd import recipe
    from vyomai_amd.shapes import text_model_shapes
    nthreads = host_cores()
    torch.set_num_threads(nthreads)
    shapes = text_model_shapes(cfg, "rope", None, head=True)
    sd = {}
    for n, s in shapes.items():
        if n.endswith("decoder.bias"):
            continue
        sd[n] = torch.from_numpy(recipe.param_value(n, s)).requires_grad_(True)
    sd["lm_head.decoder.bias"] = sd["lm_head.bias"]
    c = O.Cfg.of(cfg)
    B = 4
    ids = torch.from_numpy(recipe.token_ids("bench.cpu", (B, seq), 3, cfg.vocab_size))
    t0 = time.time()
    n = 0
    while True:
        out = O.decoder_forward(sd, c, ids, None, "rope", None, fused_sdpa=True)
        loss = O.clm_loss(out.logits, ids)
        loss.backward()
        n += 1
        if time.time() - t0 > 12 or n >= 3:
            break
    dt = time.time() - t0
    # decode leg
    with torch.no_grad():
        sdn = {k: v.detach() for k, v in sd.items()}
        prompt = ids[:1, :64]
        new = 32
        O.decoder_generate(sdn, c, prompt, torch.ones(1, 64), max_len=2, pos_type="rope", attn_type=None,
                           use_cache=True, use_static_cache=True, eos_id=-1)   # warm-up
        t1 = time.time()
        O.decoder_generate(sdn, c, prompt, torch.ones(1, 64), max_len=1, pos_type="rope", attn_type=None,
                           use_cache=True, use_static_cache=True, eos_id=-1)
        t_pre = time.time() - t1
        t1 = time.time()
        O.decoder_generate(sdn, c, prompt, torch.ones(1, 64), max_len=new, pos_type="rope", attn_type=None,
                           use_cache=True, use_static_cache=True, eos_id=-1)
        t_all = time.time() - t1
    dec = (new - 1) / max(t_all - t_pre, 1e-9)
    return {"value": round(n * B * seq / dt, 1), "unit": "tokens/s", "cores": nthreads, "kind": "port",
            "sample": f"{n} fwd+bwd passes of the 12L decoder oracle at B={B} x seq={seq}, fp32, no optimizer step",
            "decode": {"value": round(dec, 1), "unit": "tokens/s",
                       "sample": f"B=1 greedy decode, static cache, {new} tokens behind a 64-token prompt (token loop only)"}}


def build_text_model(cfg, attn, B, L, dev, rank):
    import vyomai_amd as V
    from vyomai_amd import recipe
    from vyomai_amd.training import FlatTrainer
    with contextlib.redirect_stdout(sys.stderr):   # (the constructors print the reference's notices; stdout is the JSON line's)
        model = V.DecoderModel(cfg, "rope", None if attn == "none" else "gqa")
    recipe.load_recipe_(model)
    model = model.to(dev).train()
    trainer = FlatTrainer(model, lr=5e-5, weight_decay=0.01)
    torch.manual_seed(1234 + rank)
    ids = torch.randint(3, cfg.vocab_size, (B, L), device=dev)
    return model, trainer, ids


def time_training(trainer, model, ids, steps, warmup, barrier, dev, dist_on):
    """warmup untimed steps, then exactly `steps` steps between two barrier + synchronize pairs -> (max over ranks of
    the wall time, last loss)."""
    import torch.distributed as dist

    def step():
        return trainer.train_step(lambda: model.clm_loss(ids, ids))

    loss = None
    for _ in range(warmup):
        loss = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return float(tmax.item()), loss


def time_decode(model, cfg, attn, ids, new_tokens, dev, world, dist_on):
    """KV-cache greedy decode behind the (B, L) prompts, static cache: time per token step from two runs (1 and n new
    tokens), and the HBM roofline of a step -- every layer / head weight once (the embedding table only B rows) + the
    K/V cache of the average context of the timed steps, over the measured step time."""
    import torch.distributed as dist
    B, L = ids.shape
    model.eval()
    with torch.no_grad():
        am = torch.ones(B, L, dtype=torch.long, device=dev)

        def gen(n):
            torch.cuda.synchronize()
            t = time.perf_counter()
            model.generate(ids, am, max_len=n, use_cache=True, use_static_cache=True)
            torch.cuda.synchronize()
            return time.perf_counter() - t
        gen(2)
        t1 = gen(1)
        tn = gen(new_tokens)
    model.train()
    per_tok = (tn - t1) / max(1, new_tokens - 1)
    tm = torch.tensor([per_tok], device=dev, dtype=torch.float64)
    if dist_on:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    per_tok = float(tm.item())
    d_, nl, h = cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads
    hk = getattr(cfg, "num_key_value_heads", h) if attn == "gqa" else h
    dh = d_ // h
    wbytes = (nl * (8 * d_ * d_ + 2 * d_ * d_ + 2 * d_ * hk * dh) + d_ * d_ + cfg.vocab_size * d_) * 2
    ctx = L + (new_tokens + 1) / 2.0
    kvbytes = 2 * nl * B * ctx * hk * dh * 2
    return {"tokens_per_sec": round(world * B / per_tok, 1), "ms_per_token_step": round(per_tok * 1e3, 3),
            "prefill_plus_first_token_ms": round(t1 * 1e3, 2), "batch": B, "prompt": L,
            "new_tokens": new_tokens, "cache": "StaticCacheOne", "scaling": "replicas only",
            "roofline": {"bound": "hbm", "achieved": round((wbytes + kvbytes) / per_tok * 1e-9, 1),
                         "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round((wbytes + kvbytes) / per_tok * 1e-9 / PEAK_HBM_GBS, 4),
                         "bytes_per_step": int(wbytes + kvbytes),
                         "note": f"algorithmic bytes: weights once ({wbytes / 1e6:.0f} MB) + K/V of the mean context "
                                 f"({kvbytes / 1e6:.0f} MB); a step is a chain of dependent launches, only the attention "
                                 "launch is bandwidth-bound (DESIGN section 3)"}}


def gqa_side_config(a, dev, rank, world, barrier, dist_on):
    """BASELINE.md section 4 / SURVEY section 8d name two attention types for the headline: attention_type=None (the
    headline itself) and 'gqa' with 4 key/value heads (reference models/decoder.py:116-201).  The same step and the same
    decode on the GQA model, a few steps."""
    a2 = argparse.Namespace(**vars(a))
    a2.attn = "gqa"
    cfg = make_cfg(a2)
    model, trainer, ids = build_text_model(cfg, "gqa", a.batch, a.seq, dev, rank)
    steps = min(a.steps, 8)
    dt, loss = time_training(trainer, model, ids, steps, 2, barrier, dev, dist_on)
    out = {"workload": "configs[1] with attention_type='gqa' (num_key_value_heads=4): the same CLM training step and KV-cache "
                       "greedy decode", "ms_per_step": round(dt / steps * 1e3, 3),
           "tokens_per_sec": round(world * a.batch * a.seq * steps / dt, 1), "loss": round(float(loss.item()), 4)}
    if not a.no_decode:
        dec = time_decode(model, cfg, "gqa", ids, a.decode_tokens, dev, world, dist_on)
        out["decode"] = {k: dec[k] for k in ("tokens_per_sec", "ms_per_token_step", "prefill_plus_first_token_ms", "roofline")}
    del trainer, model
    torch.cuda.empty_cache()
    return out


def spawn_ranks(a) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, as a CHILD
    torch.distributed.run process (nothing in this process has touched the GPU yet; a process that has must never exec
    another program), hand its stdout -- rank 0's one JSON line -- through and return its exit code."""
    import socket
    import subprocess
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    return subprocess.call(cmd, env=env)


def rehearse_launch(a) -> None:
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.backend if a.backend != "nccl" or torch.cuda.is_available() else "gloo")
        dist.barrier()
        t = torch.ones(1)
        if dist.get_backend() == "nccl":
            t = t.cuda(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        dist.all_reduce(t)
        assert int(t.item()) == world
    if rank == 0:
        print(json.dumps({"metric": "launcher rehearsal", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "rehearsal": "launch only: process group, barrier and one all-reduce; no kernels"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    a = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        raise SystemExit(spawn_ranks(a))
    if int(env_world or "1") != a.gpus:
        # never print a line whose n_gpus differs from what was asked for
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={env_world} ranks")
    if a.rehearse_launch:
        return rehearse_launch(a)
    # stdout carries ONE JSON line: native libraries write there too (RCCL prints a five-line version banner to fd 1 when
    # the communicator is created), so fd 1 is pointed at stderr for the run and the line goes out through a saved copy
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)  # rehearsals with more ranks than GPUs share a device (gloo only)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
    elif a.force_dist:
        os.environ["VY_DDP_FORCE_COLLECTIVES"] = "1"
        kw = {"device_id": dev} if a.backend == "nccl" else {}
        import socket
        so = socket.socket()
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
        so.close()
        dist.init_process_group(a.backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, **kw)

    cfg = make_cfg(a)
    B, L = a.batch, a.seq
    dist_on = world > 1 or a.force_dist

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    model, trainer, ids = build_text_model(cfg, a.attn, B, L, dev, rank)
    dt, loss = time_training(trainer, model, ids, a.steps, a.warmup, barrier, dev, dist_on)
    tokens = world * B * L * a.steps
    final_loss = float(loss.item())

    # ---- decode: replicas only (independent sequences per GPU, no collective) -------------------
    dec = None
    if not a.no_decode:
        dec = time_decode(model, cfg, a.attn, ids, a.decode_tokens, dev, world, dist_on)

    roof = cpu = None
    others = None
    if rank == 0:
        roof = roofline_probe(cfg, B, L, dev)
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(cfg, L)
    if not a.no_other_configs:
        # the other BASELINE.json configurations, measured beside the headline (never part of `value`):
        # the training model is released first; a failure here leaves a note, not a broken bench line.
        # configs[3] (caption training, "B=64 on 1->8 GPUs") runs on every rank under the same process group;
        # configs[4] (single-sequence decode) on one GPU only.
        others = {}
        if world == 1:
            # SURVEY section 8d's second run: the same step with 25 % right padding (key-padding path of the attention
            # kernels, ignored label positions) -- a few steps on the trainer that is still alive
            try:
                am = torch.ones(B, L, dtype=torch.long, device=dev)
                am[:, L - L // 4:] = 0
                lab = ids.clone()
                lab[am == 0] = -100
                for _ in range(2):
                    trainer.train_step(lambda: model.clm_loss(ids, lab, am))
                torch.cuda.synchronize()
                t0p = time.perf_counter()
                for _ in range(3):
                    lp = trainer.train_step(lambda: model.clm_loss(ids, lab, am))
                torch.cuda.synchronize()
                dtp = (time.perf_counter() - t0p) / 3
                others["configs[1] with 25 % right padding"] = {
                    "workload": "the headline step with attention_mask: the last 128 of 512 positions padded (key-padding + causal mask, labels -100 there)",
                    "ms_per_step": round(dtp * 1e3, 3), "tokens_per_sec_incl_padding": round(B * L / dtp, 1), "loss": round(float(lp.item()), 4)}
            except Exception as ex:   # noqa: BLE001
                others["configs[1] with 25 % right padding"] = {"error": f"{type(ex).__name__}: {ex}"}
        del trainer, model
        torch.cuda.empty_cache()
        try:
            others["configs[1] gqa"] = gqa_side_config(a, dev, rank, world, barrier, dist_on)
        except Exception as ex:   # noqa: BLE001
            # (a side measurement must never cost the headline its line: an error here is recorded on every rank -- the
            # ranks run the same code on the same shapes, so they fail together or not at all)
            others["configs[1] gqa"] = {"error": f"{type(ex).__name__}: {ex}"}
        todo = [("configs[3]", "tools.bench_vlm_training")]
        if world == 1:
            todo.append(("configs[4]", "tools.bench_paligemma"))
        for key, mod in todo:
            try:
                import importlib
                with contextlib.redirect_stdout(sys.stderr):
                    others[key] = importlib.import_module(mod).run()
            except Exception as ex:   # noqa: BLE001
                others[key] = {"error": f"{type(ex).__name__}: {ex}"}
            torch.cuda.empty_cache()
    if rank == 0:
        d = cfg.hidden_size
        fl_step = 3.0 * (block_flops_per_token(d, L) * cfg.num_hidden_layers + 2 * d * d + 2 * d * cfg.vocab_size) * B * L
        line = {
            "metric": "training tokens/sec (12L d=768 seq=512 GPT-style decoder, CLM)",
            "value": round(tokens / dt, 1), "unit": "tokens/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"configs[1]: GPT-style Decoder {cfg.num_hidden_layers}L d=768 h=12 RoPE vocab 50265, "
                                   f"CLM training B={B}/GPU seq={L}, AdamW, bf16 kernels + fp32 masters; "
                                   f"then KV-cache greedy decode {a.decode_tokens} tok",
                       "global_batch": world * B, "seq_len": L, "parallelism": f"dp{world}",
                       "attention": a.attn, "weights": "deterministic recipe (vyomai_amd.recipe)"},
            "final_loss": round(final_loss, 4),
            "train_model_tflops": round(fl_step * world * a.steps / dt * 1e-12, 1),
            "decode_tokens_per_sec": dec["tokens_per_sec"] if dec else None,
            "decode": dec,
            "roofline": roof,
            "cpu_baseline": cpu,
            "other_configs": others,
        }
        if a.force_dist:
            line["rehearsal"] = "one-rank process group with the bucket collectives forced (--force-dist): not a benchmark line"
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
