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
# ... and every parameter that is not a follower inside a packed q/k/v group STARTS on a 128-byte line of the bf16 arena
# (64 elements).  The vocabulary bias (50265 elements) used to leave everything registered after it -- the LM head's
# projections -- 64 bytes off a line: every 1536-byte weight row then straddled 13 lines instead of 12 and each 128-byte
# row segment an LDS-DMA instruction fetches came from two lines (the decode step's vocabulary projection: 28.8 us inside
# bench.py against 18.5 us on line-aligned weights, profiles/r03_*).
LINE = int(os.environ.get("VY_ARENA_LINE", "64"))   # (8: the round-2 layout, for A/B runs)


def _ordered_params(model: nn.Module) -> List[Tuple[str, nn.Parameter]]:
    """Unique parameters in registration order, except that the q/k/v projection weights (and
    biases) of each attention module are made adjacent so the packed [Wq;Wk;Wv] matrix the fused
    QKV kernel reads is one contiguous arena range."""
    from .layers.attention import _SelfAttentionBase

    named = list(model.named_parameters(remove_duplicate=True))
    by_id = {id(p): n for n, p in named}
    taken = set()
    out: List[Tuple[str, nn.Parameter]] = []
    followers: set = set()       # ids of packed-group members behind the first: they must follow it without a gap
    groups: Dict[int, List[nn.Parameter]] = {}
    for mod in model.modules():
        if isinstance(mod, _SelfAttentionBase) and not mod._fused_qkv:
            members = mod._params()
            groups[id(members[0])] = members
    for n, p in named:
        if id(p) in taken:
            continue
        if id(p) in groups:
            for j, m in enumerate(groups[id(p)]):
                out.append((by_id[id(m)], m))
                taken.add(id(m))
                if j:
                    followers.add(id(m))
        else:
            out.append((n, p))
            taken.add(id(p))
    _ordered_params.followers = followers
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
        followers = getattr(_ordered_params, "followers", set())
        for p in params:
            if id(p) not in followers:
                off = (off + LINE - 1) // LINE * LINE
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
                p._vy_shadow = (p._version, self.shadow[o:o + p.numel()].view(p.shape), True)

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
            sw = sb = None
            if self.shadow is not None:
                sw = self.shadow[o:o + n_rows * k].view(n_rows, k)
            if mod.attention_bias:
                bs = [mod.query.bias, mod.key.bias, mod.value.bias]
                assert all(b.numel() % ALIGN == 0 for b in bs)
                ob = index[id(bs[0])]
                mod._packed_b = self.master[ob:ob + n_rows]
                if self.shadow is not None:
                    sb = self.shadow[ob:ob + n_rows]
            else:
                mod._packed_b = None
            if self.shadow is not None:
                # keyed on the member versions, pinned to the arena (layers.attention._packed_shadow)
                mod._vy_pshadow = (tuple(p._version for p in mod._params()), sw, sb, mod._packed_w.data_ptr(), True)

    def zero_grad(self) -> None:
        self.grad.zero_()


class _StreamWork:
    """What torch.distributed's Work is to the reducer: wait() makes the CURRENT stream wait for the collective."""

    def __init__(self, event: torch.cuda.Event) -> None:
        self.event = event

    def wait(self) -> None:
        torch.cuda.current_stream().wait_event(self.event)


class NativeComm:
    """The library's own RCCL communicator (vy_ddp_*, include/vyom_hip.h) behind the reducer instead of
    torch.distributed.all_reduce: one communicator per process on the current device, the bucket all-reduce enqueued on
    a dedicated HIP stream that first waits for the compute stream.  torch.distributed (any backend) is used once, to hand
    rank 0's 128-byte id to the other ranks.  Opt-in: FlatTrainer(native_rccl=True) or VY_DDP_NATIVE=1."""

    def __init__(self, device: torch.device, process_group=None) -> None:
        from . import _lib
        self._lib = _lib
        # the RCCL that ships with torch is already mapped into this process (libtorch_hip links it): make ITS symbols the
        # ones vy_ddp_* binds to (dlsym on the global scope), rather than mapping a second copy from /opt/rocm
        import ctypes
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            _lib.call("vy_ddp_unique_id", uid.data_ptr())
        if world > 1:
            backend = dist.get_backend(process_group)
            t = uid.to(device) if backend == "nccl" else uid
            dist.broadcast(t, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                           group=process_group)
            uid = t.cpu()
        torch.cuda.set_device(device)
        _lib.call("vy_ddp_init", uid.data_ptr(), rank, world)
        self.world, self.rank = world, rank
        self.stream = torch.cuda.Stream(device=device)
        self.issued = 0   # vy_ddp_all_reduce_async calls

    def all_reduce(self, view: torch.Tensor) -> _StreamWork:
        self.stream.wait_stream(torch.cuda.current_stream())
        self._lib.call("vy_ddp_all_reduce_async", view.data_ptr(), view.numel(), self._lib.dtype_code(view.dtype),
                       self.stream.cuda_stream)
        self.issued += 1
        return _StreamWork(self.stream.record_event())

    def close(self) -> None:
        self._lib.call("vy_ddp_destroy")


class BucketReducer:
    """Bucketed, overlapped gradient all-reduce over a flat gradient arena.

    comm_dtype: None reduces the fp32 arena in place (652 MB per step for the 12-layer decoder);
    torch.bfloat16 reduces a bf16 copy of each bucket (326 MB) and writes the sum back in fp32 --
    a ring all-reduce over xGMI is bound by ONE ~153 GB/s link, so halving the bytes halves the
    exchange."""

    def __init__(self, arena: FlatArena, process_group=None, bucket_bytes: int = 64 << 20,
                 average: bool = True, comm_dtype: Optional[torch.dtype] = None, native: Optional[NativeComm] = None):
        self.arena = arena
        self.pg = process_group
        self.native = native   # the library's own RCCL communicator instead of torch.distributed.all_reduce
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # rehearsal knob: issue the bucket collectives even in a one-rank group (the RCCL calls, their stream
        # semantics and the per-bucket optimizer behind them run on a one-GPU box exactly as they do at N > 1)
        self.force = (dist.is_initialized() or native is not None) and os.environ.get("VY_DDP_FORCE_COLLECTIVES") == "1"
        self.average = average
        self.comm_dtype = comm_dtype
        self.enabled = True   # False during the non-final micro-steps of gradient accumulation
        self.collectives = 0  # all-reduces issued since construction, by either transport (tools/check_rccl_one_rank.py)
        # buckets over the arena in REVERSE order: backward reaches the last layers first.  Only
        # parameters that require gradients are waited for; a frozen one never reports.
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
            if arena.params[idx].requires_grad:
                count += 1
            if end - cur_start >= cap or idx == 0:
                b = len(self.buckets)
                self.buckets.append((cur_start, end))
                self.expect.append(count)
                for m in members:
                    self.bucket_of[id(arena.params[m])] = b
                end, count, members = cur_start, 0, []
        self._comm: Dict[int, torch.Tensor] = {}
        self.on_bucket = None
        self.reset()
        # called as on_bucket(b, work) when bucket b's gradients are final on this rank and its
        # all-reduce (work, None when world == 1) has been enqueued -- the trainer steps the bucket
        for p in arena.params:
            p._vy_ready = self.mark_ready
            if p.requires_grad:
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
        self._ready = [False] * len(self.buckets)
        self._next = 0               # lowest bucket index that has not gone out yet (collective order, see _launch_ready)
        self.global_flags = None     # after finish() under world > 1: device fp32 [n_params], > 0 where ANY rank had a gradient
        self._works: List = []
        self.launch_order: List[int] = []
        self._seen: set = set()
        self.touched: set = set()    # ids of the parameters that received a gradient since reset()
        self.dirty = False

    def mark_ready(self, p) -> None:
        self.touched.add(id(p))
        if not self.enabled:
            return
        self.dirty = True
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
            self._ready[b] = True
            self._launch_ready()

    def _launch_ready(self) -> None:
        """Collectives must be issued in the SAME order on every rank.  A rank whose batch leaves a parameter without a
        gradient (the reference's multimodal model, find_unused_parameters=True) completes that bucket only at
        finish(), while the other ranks complete it during backward: so with more than one rank buckets go out
        strictly by index (0 = the last-registered parameters, which backward reaches first) -- a late bucket holds
        back the ones behind it on that rank, never reorders them.  One rank alone has no collective to order and
        hands every bucket to the optimizer the moment it is final."""
        if self.world > 1 or self.force:
            while self._next < len(self.buckets) and self._ready[self._next]:
                self._launch(self._next)
        else:
            for b, r in enumerate(self._ready):
                if r and not self._launched[b]:
                    self._launch(b)

    def _launch(self, b: int, notify: bool = True):
        self._launched[b] = True
        self.launch_order.append(b)
        while self._next < len(self.buckets) and self._launched[self._next]:
            self._next += 1
        work = None
        if self.world > 1 or self.force:
            s, e = self.buckets[b]
            view = self.arena.grad[s:e]
            if self.comm_dtype is not None:
                buf = self._comm.get(b)
                if buf is None:
                    buf = self._comm[b] = torch.empty(e - s, dtype=self.comm_dtype, device=view.device)
                _convert(view, buf)
                view = buf
            if self.native is not None:
                work = self.native.all_reduce(view)
            else:
                work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self.collectives += 1
            self._works.append((b, work))
        if notify and self.on_bucket is not None:
            self.on_bucket(b, work)
        return work

    def _exchange_flags(self) -> None:
        """Which parameters received a gradient on ANY rank (one small MAX all-reduce per step, issued by every rank at
        the same point: after the last bucket).  The all-reduce gives such a parameter the same averaged gradient on
        every rank, so every rank must update it -- also the ranks whose own batch never reached it; a parameter no
        rank touched is updated by none (torch DDP leaves globally unused gradients alone as well).  The flags stay on
        the device: the optimizer launches of locally untouched parameters are gated by them (vy_adamw_step_gated), no
        host round trip."""
        if not (self.world > 1 or self.force):
            self.global_flags = None
            return
        a = self.arena
        host = torch.tensor([1.0 if id(p) in self.touched else 0.0 for p in a.params], dtype=torch.float32)
        if a.grad.is_cuda:
            flags = host.pin_memory().to(a.grad.device, non_blocking=True)
        else:
            flags = host
        if self.native is not None:
            # the native surface only sums: a sum of 0/1 flags is > 0 exactly where the MAX is 1
            self.native.all_reduce(flags).wait()
        else:
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.pg)
        self.collectives += 1
        self.global_flags = flags

    def writeback(self, b: int) -> None:
        """comm_dtype buckets: the reduced low-precision sum back into the fp32 arena (the caller has
        made the current stream wait for the bucket's all-reduce)."""
        if self.comm_dtype is not None and (self.world > 1 or self.force) and b in self._comm:
            s, e = self.buckets[b]
            _convert(self._comm[b], self.arena.grad[s:e])

    def finish(self) -> float:
        """Flush buckets whose parameters never got a gradient, wait for the exchange and return
        the scale the optimizer must apply to turn the summed gradients into the average."""
        late = [(b, self._launch(b, notify=False)) for b in range(len(self.buckets)) if not self._launched[b]]
        self._exchange_flags()
        if self.on_bucket is not None:
            for b, work in late:
                self.on_bucket(b, work)
        for b, w in self._works:
            w.wait()
            if self.on_bucket is None:   # (an on_bucket consumer writes back on its own stream)
                self.writeback(b)
        self._works = []
        return (1.0 / self.world) if (self.average and self.world > 1) else 1.0


def _convert(src: torch.Tensor, dst: torch.Tensor) -> None:
    if src.is_cuda:
        from . import ops
        ops.cast(src, dst)
    else:
        dst.copy_(src)


class FlatTrainer:
    """AdamW data-parallel trainer: bf16 HIP kernels, fp32 master weights, fused optimizer.

    Mirrors what the reference's loops get from accelerate (Examples/vyom-ai-decoder_clm.ipynb cell 31:
    Accelerator(gradient_accumulation_steps=2); Examples/vyomai-fused-kernals-2t4.ipynb cell 0:
    clip_grad_norm_):
      accumulate_steps  k > 1: backward() scales the loss by 1/k, gradients of k micro-steps add up in
                        the arena, the buckets are reduced (and stepped) during the LAST micro-step's
                        backward only; zero_grad() / optimizer_step() are no-ops in between, as under
                        `with accelerator.accumulate(model)`.
      max_grad_norm     global L2 clip over the reduced gradient arena (torch.nn.utils.clip_grad_norm_
                        semantics: coefficient min(1, max_norm / (norm + 1e-6))) folded into the AdamW
                        kernel's gradient scale on the device -- needs every gradient before the first
                        update, so the per-bucket overlapped optimizer is off when it is set.
    Parameters with requires_grad=False, and parameters that received no gradient in a step, are not
    updated at all (torch.optim.AdamW skips p.grad is None: no weight decay either)."""

    def __init__(self, model: nn.Module, lr: float = 5e-5, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.01, compute_dtype: torch.dtype = torch.bfloat16,
                 process_group=None, bucket_bytes: int = 64 << 20, overlap_optimizer: bool = True,
                 accumulate_steps: int = 1, max_grad_norm: Optional[float] = None,
                 grad_comm_dtype: Optional[torch.dtype] = None, native_rccl: Optional[bool] = None):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        for m in model.modules():
            if hasattr(m, "compute_dtype"):
                m.compute_dtype = compute_dtype
        # fp32 compute (the parity path) reads the masters themselves: no shadow arena
        self.arena = FlatArena(model, shadow_dtype=None if compute_dtype == torch.float32 else compute_dtype)
        self.m = torch.zeros_like(self.arena.master)
        self.v = torch.zeros_like(self.arena.master)
        # native_rccl: the exchange goes through the library's own RCCL communicator (vy_ddp_*) instead of
        # torch.distributed.all_reduce (None: the VY_DDP_NATIVE environment variable decides; default off)
        if native_rccl is None:
            native_rccl = os.environ.get("VY_DDP_NATIVE") == "1"
        self.native = None
        if native_rccl and self.arena.master.is_cuda and (dist.is_initialized() or os.environ.get("VY_DDP_FORCE_COLLECTIVES") == "1"):
            self.native = NativeComm(self.arena.master.device, process_group)
        self.reducer = BucketReducer(self.arena, process_group, bucket_bytes, comm_dtype=grad_comm_dtype, native=self.native)
        self.step_count = 0
        self.accumulate_steps = max(1, int(accumulate_steps))
        self.max_grad_norm = max_grad_norm
        self.last_grad_norm: Optional[torch.Tensor] = None   # device scalar (pre-clip norm of the last step)
        self._micro = 0            # micro-steps taken since the last optimizer step
        # AdamW is an HBM stream (30 B/param), the backward GEMMs are not HBM-bound: each gradient
        # bucket is stepped on a side stream as soon as it is final (and reduced), under the rest of
        # backward.  A bucket is final only when every layer that owns a parameter in it has finished
        # its backward, so nothing that still runs reads the weights being rewritten.
        if os.environ.get("VY_OVERLAP_OPT") == "0":   # A/B knob
            overlap_optimizer = False
        if max_grad_norm is not None:
            overlap_optimizer = False
        self._side = torch.cuda.Stream() if (overlap_optimizer and self.arena.master.is_cuda) else None
        self._stepped: set = set()
        self._scale = (1.0 / self.reducer.world) if (self.reducer.average and self.reducer.world > 1) else 1.0
        if self._side is not None:
            self.reducer.on_bucket = self._bucket_final
        self._needs_zero = False   # an optimizer step has consumed the arena: zero_grad() must come next

    # ---- arena ranges that an update may touch ---------------------------------------------------
    def _ranges(self, lo: int, hi: int, touched_only: bool) -> List[Tuple[int, int]]:
        """Merged element ranges inside [lo, hi) of the parameters that are trainable (and, when
        touched_only, received a gradient since zero_grad())."""
        a = self.arena
        out: List[Tuple[int, int]] = []
        import bisect
        i = bisect.bisect_right(a.offsets, lo) - 1
        i = max(i, 0)
        while i < len(a.params) and a.offsets[i] < hi:
            p, o = a.params[i], a.offsets[i]
            e = a.offsets[i + 1] if i + 1 < len(a.params) else a.numel
            if o >= lo and p.requires_grad and (not touched_only or id(p) in self.reducer.touched):
                if out and out[-1][1] == o:
                    out[-1] = (out[-1][0], e)
                else:
                    out.append((o, e))
            i += 1
        return out

    def zero_grad(self) -> None:
        if 0 < self._micro < self.accumulate_steps:   # in the middle of an accumulation window: keep adding
            return
        self._micro = 0            # (a finished window that was never stepped is dropped here: its gradients are cleared)
        from .autograd_train import _wgrad_group
        _wgrad_group.discard()   # (only an aborted backward can have left deferred weight gradients behind)
        self.arena.zero_grad()
        self.reducer.reset()
        self._stepped = set()
        self._needs_zero = False

    def backward(self, loss: torch.Tensor) -> None:
        if self._needs_zero:
            # the reducer still holds the previous step's state: every bucket counts as launched and
            # stepped, so this backward would neither reduce nor update anything -- silently
            raise RuntimeError("FlatTrainer.backward() after optimizer_step() without zero_grad(): call "
                               "zero_grad() (or use train_step()) so the gradient arena and the bucket "
                               "reducer start from a clean state")
        k = self.accumulate_steps
        if self._micro >= k and not (self._side is None and self.reducer.world == 1 and not self.reducer.force):
            # a second backward before optimizer_step(): every bucket of the first pass has already gone out (and, with
            # the overlapped optimizer, been stepped), so these gradients would be added on top of an already reduced
            # sum and never exchanged -- the ranks would drift apart without a word
            raise RuntimeError(f"FlatTrainer.backward() called {self._micro + 1} times before optimizer_step() with "
                               f"accumulate_steps={k}: construct the trainer with accumulate_steps=<number of "
                               "micro-batches> (the loss is scaled by 1/k for you) instead of accumulating by hand")
        last = (self._micro + 1) >= k
        # buckets are reduced -- and, with the overlapped optimizer, stepped -- as soon as their gradients
        # are final, which is only true in the last micro-step's backward
        self.reducer.enabled = last
        if last:
            self.reducer._seen = set()
        self._micro += 1
        (loss / k if k > 1 else loss).backward()
        self.reducer.enabled = True

    def _adamw(self, lo: int, hi: int, step: int, scale_dev: Optional[torch.Tensor] = None,
               gate: Optional[torch.Tensor] = None) -> None:
        """gate: one fp32 on the device; the launch leaves the range alone when it is 0 (a parameter without a gradient
        on this rank whose fate depends on the other ranks: BucketReducer._exchange_flags)."""
        from . import ops
        a = self.arena
        ops.adamw_step(a.master[lo:hi], a.grad[lo:hi], self.m[lo:hi], self.v[lo:hi],
                       None if a.shadow is None else a.shadow[lo:hi], self.lr, self.betas[0], self.betas[1],
                       self.eps, self.weight_decay, step, self._scale, scale_dev=scale_dev, gate=gate)

    def _step_range(self, lo: int, hi: int, step: int, scale_dev: Optional[torch.Tensor] = None) -> None:
        """AdamW over the trainable parameters inside [lo, hi) that received a gradient -- on this rank (merged plain
        launches) or, judged on the device by the exchanged flags, on any other rank (one gated launch each)."""
        for s, e in self._ranges(lo, hi, touched_only=True):
            self._adamw(s, e, step, scale_dev)
        flags = self.reducer.global_flags
        if flags is None:
            return
        a = self.arena
        import bisect
        i = max(bisect.bisect_right(a.offsets, lo) - 1, 0)
        while i < len(a.params) and a.offsets[i] < hi:
            p, o = a.params[i], a.offsets[i]
            if o >= lo and p.requires_grad and id(p) not in self.reducer.touched:
                e = a.offsets[i + 1] if i + 1 < len(a.params) else a.numel
                self._adamw(o, e, step, scale_dev, gate=flags[i:i + 1])
            i += 1

    def _bucket_final(self, b: int, work) -> None:
        """Reducer callback (during backward): step bucket b on the side stream."""
        lo, hi = self.reducer.buckets[b]
        ev = torch.cuda.current_stream().record_event()   # after the kernels that wrote its gradients
        self._side.wait_event(ev)
        with torch.cuda.stream(self._side):
            if work is not None:
                work.wait()                                # the side stream waits for the all-reduce
                self.reducer.writeback(b)
            self._step_range(lo, hi, self.step_count + 1)
        self._stepped.add(b)

    def _sumsq(self) -> torch.Tensor:
        from . import ops
        return ops.sumsq(self.arena.grad)

    def grad_norm(self) -> torch.Tensor:
        """L2 norm of the (averaged) gradient arena as a device scalar (after the exchange)."""
        return self._sumsq().sqrt() * self._scale

    def optimizer_step(self) -> bool:
        """-> True when an update was applied (False inside an accumulation window)."""
        from .autograd_train import WEIGHT_EPOCH

        if self._micro < self.accumulate_steps:
            return False
        self._micro = 0
        scale = self.reducer.finish()   # flushes buckets without gradients (their callbacks run here too)
        assert abs(scale - self._scale) < 1e-12
        self.step_count += 1
        if self._side is None:
            scale_dev = None
            if self.max_grad_norm is not None:
                # clip coefficient on the device: min(1, max_norm / (norm + 1e-6)); no host sync
                norm = self._sumsq().sqrt() * self._scale
                self.last_grad_norm = norm
                scale_dev = torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0).to(torch.float32).reshape(1)
            self._step_range(0, self.arena.numel, self.step_count, scale_dev)
        else:
            for b, (lo, hi) in enumerate(self.reducer.buckets):
                if b not in self._stepped:
                    self._step_range(lo, hi, self.step_count)
            torch.cuda.current_stream().wait_stream(self._side)
        WEIGHT_EPOCH[0] += 1
        self._needs_zero = True
        return True

    def train_step(self, loss_fn: Callable[[], torch.Tensor]) -> torch.Tensor:
        """zero_grad -> loss_fn() -> backward (overlapped all-reduce) -> fused AdamW.  With
        accumulate_steps = k this is ONE micro-step; the update happens on every k-th call."""
        self.zero_grad()
        loss = loss_fn()
        self.backward(loss)
        self.optimizer_step()
        return loss.detach()

    # ---- checkpoint / external writes -----------------------------------------------------------
    def resync(self, reset_moments: bool = False) -> None:
        """After anything but this trainer wrote the parameters in place (load_state_dict, a manual
        re-initialisation): recast the fp32 masters into the compute-dtype arena the kernels read and
        invalidate every cached W^T."""
        from .autograd_train import WEIGHT_EPOCH
        self.arena.refresh_shadow(install=True)
        self.arena._install_packed(self.model)
        if reset_moments:
            self.m.zero_()
            self.v.zero_()
            self.step_count = 0
        WEIGHT_EPOCH[0] += 1

    def load_state_dict(self, state_dict, strict: bool = True, reset_moments: bool = False):
        """model.load_state_dict() into the master arena (the parameters are views of it), then resync()."""
        out = self.model.load_state_dict(state_dict, strict=strict)
        self.resync(reset_moments=reset_moments)
        return out

    def optimizer_state_dict(self) -> dict:
        return {"step": self.step_count, "m": self.m.clone(), "v": self.v.clone(),
                "names": [n for n, _ in self.arena.items], "offsets": list(self.arena.offsets)}

    def load_optimizer_state_dict(self, sd: dict) -> None:
        if list(sd["offsets"]) != list(self.arena.offsets):
            raise ValueError("optimizer state was saved for a different parameter layout")
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.step_count = int(sd["step"])
