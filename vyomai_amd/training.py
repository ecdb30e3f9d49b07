"""Data-parallel training engine for the VyomAI models on MI355X.

The reference trains through HuggingFace accelerate -> torch DDP (Examples/vyom-ai-decoder_clm.ipynb
cells 31, 33; Examples/vyomai-fused-kernals-2t4.ipynb cell 0).  This engine is the MI355X-first
replacement, one process per GPU:

  * FlatArena: every parameter becomes a view into ONE fp32 master buffer; gradients live in ONE
    fp32 buffer with the same layout (wgrad kernels accumulate straight into it), AdamW moments in
    two more, and the bf16 weights the kernels read in a fifth.  288 GB of HBM3E makes the extra
    copies free and turns the optimizer into a single kernel launch and the gradient exchange into
    a handful of large contiguous collectives.
  * BucketReducer: the gradient arena is cut into contiguous buckets in reverse registration order
    (= the order backward produces them).  When the last gradient of a bucket has been written the
    bucket is all-reduced asynchronously; RCCL runs on torch.distributed's own HIP stream, so the
    exchange overlaps the rest of backward.  xGMI is point-to-point (7 links x ~153 GB/s), so
    buckets are few and large (default 64 MB) rather than NVSwitch-sized.  Parameters that never
    receive a gradient (the reference needs find_unused_parameters=True for its multimodal model,
    Examples/vyom-ai-accelerate-multimodel-2t4.ipynb cell 1) are flushed at finish().
  * FlatTrainer: forward/backward on the HIP kernels + vy_adamw_step over the arenas.

The arena/reducer logic is device-agnostic (it is exercised with gloo on CPU in tests/); only the
fused optimizer step needs the GPU.
"""
from __future__ import annotations

import os

from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

ALIGN = 8  # elements: keeps every bf16 parameter 16-byte aligned inside the shadow arena


def _ordered_params(model: nn.Module) -> List[Tuple[str, nn.Parameter]]:
    """Unique parameters in registration order, except that the q/k/v projection weights (and
    biases) of each attention module are made adjacent so the packed [Wq;Wk;Wv] matrix the fused
    QKV kernel reads is one contiguous arena range."""
    from .layers.attention import _SelfAttentionBase

    named = list(model.named_parameters(remove_duplicate=True))
    by_id = {id(p): n for n, p in named}
    taken = set()
    out: List[Tuple[str, nn.Parameter]] = []
    groups: Dict[int, List[nn.Parameter]] = {}
    for mod in model.modules():
        if isinstance(mod, _SelfAttentionBase) and not mod._fused_qkv:
            members = mod._params()
            groups[id(members[0])] = members
    for n, p in named:
        if id(p) in taken:
            continue
        if id(p) in groups:
            for m in groups[id(p)]:
                out.append((by_id[id(m)], m))
                taken.add(id(m))
        else:
            out.append((n, p))
            taken.add(id(p))
    return out


class FlatArena:
    """fp32 master + fp32 grad (+ optional bf16 shadow) arenas; parameters and their .grad become
    views."""

    def __init__(self, model: nn.Module, shadow_dtype: Optional[torch.dtype] = torch.bfloat16):
        self.items = _ordered_params(model)
        params = [p for _, p in self.items]
        device = params[0].device
        self.offsets: List[int] = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self.master = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.shadow = torch.zeros(off, dtype=shadow_dtype, device=device) if shadow_dtype is not None else None
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                n = p.numel()
                self.master[o:o + n].copy_(p.detach().reshape(-1).float())
                p.data = self.master[o:o + n].view(p.shape)
                p.grad = self.grad[o:o + n].view(p.shape)
                p._vy_direct = True
        self.params = params
        self.refresh_shadow(install=True)
        self._install_packed(model)

    def refresh_shadow(self, install: bool = False) -> None:
        if self.shadow is None:
            return
        if self.master.is_cuda:
            from . import ops
            ops.cast(self.master, self.shadow)
        else:
            self.shadow.copy_(self.master)
        if install:
            for p, o in zip(self.params, self.offsets):
                p._vy_shadow = (p._version, self.shadow[o:o + p.numel()].view(p.shape))

    def _install_packed(self, model: nn.Module) -> None:
        """Point each attention module's packed weight/bias (and their bf16 shadows) at the arena."""
        from .layers.attention import _SelfAttentionBase

        index = {id(p): o for p, o in zip(self.params, self.offsets)}
        for mod in model.modules():
            if not isinstance(mod, _SelfAttentionBase) or mod._fused_qkv:
                continue
            ws = [mod.query.weight, mod.key.weight, mod.value.weight]
            n_rows, k = sum(w.shape[0] for w in ws), ws[0].shape[1]
            o = index[id(ws[0])]
            assert all(w.numel() % ALIGN == 0 for w in ws)
            mod._packed_w = self.master[o:o + n_rows * k].view(n_rows, k)
            if self.shadow is not None:
                mod._packed_w._vy_shadow = (mod._packed_w._version, self.shadow[o:o + n_rows * k].view(n_rows, k))
            if mod.attention_bias:
                bs = [mod.query.bias, mod.key.bias, mod.value.bias]
                assert all(b.numel() % ALIGN == 0 for b in bs)
                ob = index[id(bs[0])]
                mod._packed_b = self.master[ob:ob + n_rows]
                if self.shadow is not None:
                    mod._packed_b._vy_shadow = (mod._packed_b._version, self.shadow[ob:ob + n_rows])
            else:
                mod._packed_b = None

    def zero_grad(self) -> None:
        self.grad.zero_()


class BucketReducer:
    """Bucketed, overlapped gradient all-reduce over a flat gradient arena."""

    def __init__(self, arena: FlatArena, process_group=None, bucket_bytes: int = 64 << 20,
                 average: bool = True):
        self.arena = arena
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        # buckets over the arena in REVERSE order: backward reaches the last layers first
        self.buckets: List[Tuple[int, int]] = []      # (start, end) element ranges
        self.bucket_of: Dict[int, int] = {}
        self.expect: List[int] = []
        cap = max(1, bucket_bytes // 4)
        end = arena.numel
        cur_start = end
        count = 0
        members: List[int] = []
        for idx in range(len(arena.params) - 1, -1, -1):
            o = arena.offsets[idx]
            cur_start = o
            members.append(idx)
            count += 1
            if end - cur_start >= cap or idx == 0:
                b = len(self.buckets)
                self.buckets.append((cur_start, end))
                self.expect.append(count)
                for m in members:
                    self.bucket_of[id(arena.params[m])] = b
                end, count, members = cur_start, 0, []
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._works: List = []
        self.launch_order: List[int] = []
        self._seen: set = set()
        # called as on_bucket(b, work) when bucket b's gradients are final on this rank and its
        # all-reduce (work, None when world == 1) has been enqueued -- the trainer steps the bucket
        self.on_bucket = None
        for p in arena.params:
            p._vy_ready = self.mark_ready
            # gradients produced by torch autograd itself (embedding tables) arrive through the tape
            p.register_post_accumulate_grad_hook(self._hook)

    def _hook(self, p) -> None:
        # (a weight whose gradient launch has been deferred -- autograd_train._WgradGroup -- reports itself
        # when that launch has been enqueued; the tape's hook comes too early for it)
        if not getattr(p, "_vy_deferred", False):
            self.mark_ready(p)

    def reset(self) -> None:
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._works = []
        self.launch_order = []
        self._seen = set()

    def mark_ready(self, p) -> None:
        b = self.bucket_of[id(p)]
        if self._launched[b]:
            return
        if id(p) in self._seen:   # a parameter reports once per backward, however many ops fed it
            if os.environ.get("VY_DEBUG_READY"):
                print("double ready:", [n for n, q in self.arena.items if q is p])
            return
        self._seen.add(id(p))
        self._pending[b] += 1
        if self._pending[b] >= self.expect[b]:
            self._launch(b)

    def _launch(self, b: int) -> None:
        self._launched[b] = True
        self.launch_order.append(b)
        work = None
        if self.world > 1:
            s, e = self.buckets[b]
            view = self.arena.grad[s:e]
            work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._works.append(work)
        if self.on_bucket is not None:
            self.on_bucket(b, work)

    def finish(self) -> float:
        """Flush buckets whose parameters never got a gradient, wait for the exchange and return
        the scale the optimizer must apply to turn the summed gradients into the average."""
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        for w in self._works:
            w.wait()
        self._works = []
        return (1.0 / self.world) if (self.average and self.world > 1) else 1.0


class FlatTrainer:
    """AdamW data-parallel trainer: bf16 HIP kernels, fp32 master weights, fused optimizer."""

    def __init__(self, model: nn.Module, lr: float = 5e-5, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.01, compute_dtype: torch.dtype = torch.bfloat16,
                 process_group=None, bucket_bytes: int = 64 << 20, overlap_optimizer: bool = True):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        for m in model.modules():
            if hasattr(m, "compute_dtype"):
                m.compute_dtype = compute_dtype
        self.arena = FlatArena(model, shadow_dtype=compute_dtype)
        self.m = torch.zeros_like(self.arena.master)
        self.v = torch.zeros_like(self.arena.master)
        self.reducer = BucketReducer(self.arena, process_group, bucket_bytes)
        self.step_count = 0
        # AdamW is an HBM stream (30 B/param), the backward GEMMs are not HBM-bound: each gradient
        # bucket is stepped on a side stream as soon as it is final (and reduced), under the rest of
        # backward.  A bucket is final only when every layer that owns a parameter in it has finished
        # its backward, so nothing that still runs reads the weights being rewritten.
        if os.environ.get("VY_OVERLAP_OPT") == "0":   # A/B knob
            overlap_optimizer = False
        self._side = torch.cuda.Stream() if (overlap_optimizer and self.arena.master.is_cuda) else None
        self._stepped: set = set()
        self._scale = (1.0 / self.reducer.world) if (self.reducer.average and self.reducer.world > 1) else 1.0
        if self._side is not None:
            self.reducer.on_bucket = self._bucket_final

    def zero_grad(self) -> None:
        from .autograd_train import _wgrad_group
        _wgrad_group.discard()   # (only an aborted backward can have left deferred weight gradients behind)
        self.arena.zero_grad()
        self.reducer.reset()
        self._stepped = set()
        self._backward_calls = 0

    def backward(self, loss: torch.Tensor) -> None:
        if self.reducer.world > 1 and getattr(self, "_backward_calls", 0) >= 1:
            # a bucket is all-reduced as soon as its gradients are final, i.e. during the first backward pass
            raise RuntimeError("gradient accumulation over several backward passes is not supported with more than "
                               "one rank (the buckets are reduced during the first pass)")
        if self._side is not None and getattr(self, "_backward_calls", 0) >= 1:
            # the per-bucket AdamW of the overlapped optimizer runs as soon as a bucket's gradients are final,
            # i.e. during the FIRST backward pass after zero_grad()
            raise RuntimeError("gradient accumulation over several backward passes needs "
                               "FlatTrainer(..., overlap_optimizer=False)")
        self._backward_calls = getattr(self, "_backward_calls", 0) + 1
        loss.backward()

    def _adamw(self, lo: int, hi: int, step: int) -> None:
        from . import ops
        a = self.arena
        ops.adamw_step(a.master[lo:hi], a.grad[lo:hi], self.m[lo:hi], self.v[lo:hi],
                       None if a.shadow is None else a.shadow[lo:hi], self.lr, self.betas[0], self.betas[1],
                       self.eps, self.weight_decay, step, self._scale)

    def _bucket_final(self, b: int, work) -> None:
        """Reducer callback (during backward): step bucket b on the side stream."""
        lo, hi = self.reducer.buckets[b]
        ev = torch.cuda.current_stream().record_event()   # after the kernels that wrote its gradients
        self._side.wait_event(ev)
        with torch.cuda.stream(self._side):
            if work is not None:
                work.wait()                                # the side stream waits for the all-reduce
            self._adamw(lo, hi, self.step_count + 1)
        self._stepped.add(b)

    def optimizer_step(self) -> None:
        from .autograd_train import WEIGHT_EPOCH

        scale = self.reducer.finish()   # flushes buckets without gradients (their callbacks run here too)
        assert abs(scale - self._scale) < 1e-12
        self.step_count += 1
        if self._side is None:
            self._adamw(0, self.arena.numel, self.step_count)
        else:
            for b, (lo, hi) in enumerate(self.reducer.buckets):
                if b not in self._stepped:
                    self._adamw(lo, hi, self.step_count)
            torch.cuda.current_stream().wait_stream(self._side)
        WEIGHT_EPOCH[0] += 1

    def train_step(self, loss_fn: Callable[[], torch.Tensor]) -> torch.Tensor:
        """zero_grad -> loss_fn() -> backward (overlapped all-reduce) -> fused AdamW."""
        self.zero_grad()
        loss = loss_fn()
        self.backward(loss)
        self.optimizer_step()
        return loss.detach()
