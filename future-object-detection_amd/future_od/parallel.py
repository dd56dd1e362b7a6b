"""Data-parallel wrapper: one process per GPU, gradients averaged over RCCL/xGMI.

`FodDataParallel` subclasses torch's DistributedDataParallel so that `isinstance(model,
DistributedDataParallel)` checks and `.module` accesses in the reference's Trainer / optimizer setup
(reference future_od/trainer.py:40-42, runs/_helper.py:85-87) keep working, and its constructor still
verifies shapes across ranks and broadcasts rank 0's parameters.  The per-step work of torch's reducer is
NOT used:

  * torch's reducer hooks every parameter: per gradient one copy into a bucket and one divide by the world
    size -- ~1100 extra small kernels (+4.4 ms of GPU time) and ~6 ms of host time per step on this model
    before any byte moves (measured with a one-rank RCCL group: 36.6 -> 41.1 ms/step); its runtime-statistics
    logger additionally blocks the host on CUDA events in the first 10 iterations and every 100th;
  * here every parameter gradient is already a slice of ONE flat fp32 buffer, the zero arena the `_acc`
    kernels accumulate into (`native/functional.py`), filled in backward execution order: transformer
    (decoder -> encoder), then the backbone sweep layer4 -> layer2.  `GradientReducer` averages that buffer
    in place with a handful of large all-reduces (`ReduceOp.AVG`) on a side stream:
        flush A  when the backbone's backward starts: every transformer gradient is final (~120 MB);
                 it overlaps the whole backbone backward (~13 ms of GPU time);
        flush B  inside the backbone sweep whenever another `bucket_cap_mb` of gradients is complete;
        flush C  at the end of backward: the tail, and any gradient that does not live in the arena (the
                 first step, before the arena exists; parameters whose gradient autograd summed itself);
    xGMI is point-to-point (a ring all-reduce is bound by one ~153 GB/s link), so few, large messages;
  * gradients hold the average over ranks when `backward()` returns (stream-ordered: the compute stream waits
    for the side stream), exactly DDP's contract, so any optimizer / clipping code works unchanged;
  * no `device_ids`: with them DDP walks the input dict every step and moves every tensor to the device,
    including the loader's HOST copies of the annotations (`_host_annotations`) that exist so that the step
    never reads them back;
  * buffers are never broadcast (FrozenBatchNorm statistics do not change; the reference re-broadcasts
    106 240 floats before every forward).

Why in-place reduction of arena regions is safe: a region is flushed only when every kernel that writes it has
been queued on the compute stream (the side stream waits on an event recorded at the flush) and nothing writes
it again -- parameters used several times (query_scale, the shared decoder norm) live in the transformer, whose
gradients are all final at flush A (autograd runs nodes in reverse creation order and the backbone is the first
node created); backbone weights are used once.  Temporaries that share the arena are consumed by kernels queued
before the flush; averaging them afterwards is harmless.
"""
import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel


class GradientReducer:
    def __init__(self, params, process_group=None, bucket_mb=48):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self.native_avg = dist.get_backend(process_group) == "nccl"
        self.comm = None                 # side stream, created on first use (CUDA only)
        self.enabled = True
        self.stats = {"arena_flushes": 0, "arena_elems": 0, "stragglers": 0}      # of the last backward
        self._checks_left = 3
        self._verdict = None
        self._stats = dict(self.stats)
        self._reset()

    def _check_layout(self):
        """In-place averaging of arena regions assumes that every rank allocated the same gradients in the same
        order.  Verified in the first steps: the flush counts and sizes must agree across ranks.  The verdict of a
        step is read at the end of the NEXT backward (through pinned memory and an event), so the check never stalls
        the launching thread on the GPU."""
        self._read_layout_verdict()
        s = self.stats
        mine = torch.tensor([s["arena_flushes"], s["arena_elems"], s["stragglers"]], dtype=torch.int64)
        dev = self.comm.device if self.comm is not None else torch.device("cpu")
        lo, hi = mine.to(dev), mine.to(dev)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if dev.type == "cuda":
            both = torch.empty((2, 3), dtype=torch.int64, pin_memory=True)
            both.copy_(torch.stack([lo, hi]), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            both, ev = torch.stack([lo, hi]), None
        self._verdict = (mine, both, ev)
        if ev is None:
            self._read_layout_verdict()          # CPU tensors: nothing to wait for

    def _read_layout_verdict(self):
        if self._verdict is None:
            return
        (mine, both, ev), self._verdict = self._verdict, None
        if ev is not None:
            ev.synchronize()
        if not torch.equal(both[0], both[1]):
            raise RuntimeError(f"FodDataParallel: ranks disagree on the gradient layout (this rank {mine.tolist()}, "
                               f"min {both[0].tolist()}, max {both[1].tolist()}): the autograd graphs differ between ranks")

    def _reset(self):
        self.start = 0                   # arena offset when this backward started: only [start, flushed) is ever averaged in place
        self.flushed = 0                 # arena elements already queued for reduction in this backward
        self.arena_buf = None
        self.armed = False               # end-of-backward callback queued
        self.accumulating = False        # gradients from an earlier (un-synchronised) backward are still in place
        self._unpack = []

    # ---- called during backward (native/backbone.py, the hook FodDataParallel puts on the loss) -------
    def arm(self):
        """Queue the end-of-backward callback (must be called from inside a backward pass)."""
        if self.enabled and not self.armed:
            self.armed = True
            self._stats = {"arena_flushes": 0, "arena_elems": 0, "stragglers": 0}
            # Gradient accumulation (a backward under no_sync(), then this one, no zero_grad in between): the arena
            # still holds the earlier gradients and autograd will ADD this pass's into the same slices at its own
            # pace -- an in-place flush would average slices that are not final yet and race with those adds.  Then
            # nothing is averaged in place: every gradient goes through the packed path at the end of backward.
            self.accumulating = any(p.grad is not None for p in self.params)
            from future_od.native import functional as Fn
            self.start = self.flushed = Fn.ARENA.off if Fn.ARENA.active else 0
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)

    def _all_reduce(self, t):
        if self.native_avg:
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
        else:                            # gloo (CPU tests, 1-GPU rehearsal): sum, then scale
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t.div_(self.world)

    def _on_side_stream(self, tensors):
        """Average each of `tensors` in place, ordered after everything queued so far on the compute stream."""
        if not tensors:
            return
        if tensors[0].is_cuda:
            dev = tensors[0].device
            if self.comm is None:
                self.comm = torch.cuda.Stream(device=dev)
            ev = torch.cuda.current_stream(dev).record_event()
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                for t in tensors:
                    t.record_stream(self.comm)
                    self._all_reduce(t)
        else:
            for t in tensors:
                self._all_reduce(t)

    def flush(self, arena, min_elems=1):
        """Average the arena region filled since the last flush, if it holds at least `min_elems` elements."""
        from future_od.native import functional as Fn
        Fn.WGRADS.flush()                # weight gradients still waiting for their launch are part of the region
        if not self.enabled or not arena.active or arena.buf is None:
            return
        self.arm()
        if self.accumulating or arena.off - self.flushed < min_elems:
            return
        self.arena_buf = arena.buf
        region = arena.buf[self.flushed:arena.off]
        self._stats["arena_flushes"] += 1
        self._stats["arena_elems"] += region.numel()
        self.flushed = arena.off
        self._on_side_stream([region])

    def maybe_flush(self, arena):
        self.flush(arena, self.bucket_elems)

    # ---- end of backward -------------------------------------------------------------------------
    def _in_flushed_arena(self, g):
        if self.arena_buf is None or not g.is_cuda or g.dtype != torch.float32:
            return False
        lo = self.arena_buf.data_ptr()
        return lo + 4 * self.start <= g.data_ptr() < lo + 4 * self.flushed

    def _reduce_stragglers(self):
        todo = [p.grad for p in self.params if p.grad is not None and not self._in_flushed_arena(p.grad)]
        self._stats["stragglers"] = len(todo)
        if not todo:
            return
        if len(todo) <= 8 and all(g.is_contiguous() for g in todo):
            self._on_side_stream(todo)
            return
        by_dtype = {}
        for g in todo:
            by_dtype.setdefault(g.dtype, []).append(g)
        for grads in by_dtype.values():      # first step (no arena yet): one packed all-reduce per dtype
            flat = torch.cat([g.reshape(-1) for g in grads])
            self._on_side_stream([flat])
            self._unpack.append((flat, grads))

    def finish(self):
        """Runs once at the end of the backward pass (autograd engine callback)."""
        try:
            if not self.enabled:
                return
            from future_od.native import functional as Fn
            self.flush(Fn.ARENA)
            self._reduce_stragglers()
            if self.comm is not None:
                torch.cuda.current_stream(self.comm.device).wait_stream(self.comm)
            for flat, grads in self._unpack:
                off = 0
                for g in grads:
                    g.copy_(flat[off:off + g.numel()].view(g.shape))
                    off += g.numel()
            self.stats = self._stats
            if self._checks_left > 0:
                self._checks_left -= 1
                self._check_layout()
            else:
                self._read_layout_verdict()      # of the last checked step
        finally:
            self._reset()


class _ArmHook:
    def __init__(self, red):
        self.red = red

    def __call__(self, grad):
        self.red.arm()
        return None


def _tensors_requiring_grad(obj):
    if isinstance(obj, torch.Tensor):
        if obj.requires_grad:
            yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_requiring_grad(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_requiring_grad(v)


class _NoSync:
    def __init__(self, ddp):
        self.ddp = ddp

    def __enter__(self):
        self.prev = self.ddp._sync_enabled
        self.ddp._sync_enabled = False

    def __exit__(self, *exc):
        self.ddp._sync_enabled = self.prev


class FodDataParallel(DistributedDataParallel):
    def __init__(self, module, device=None, bucket_cap_mb=48, find_unused_parameters=False, process_group=None):
        """A DistributedDataParallel by type (the reference's `isinstance` checks and `.module` accesses), WITHOUT torch's
        reducer: DistributedDataParallel.__init__ builds it, and with it stashes every parameter's AccumulateGrad node
        on the default stream -- unused weight here (this class averages the gradients itself), and the one thing that
        keeps a backward pass from being captured into a hipGraph on another stream (future_od/graph.py).  What the
        constructor is needed for is done directly: rank 0's parameters and buffers are broadcast, shapes verified.
        FOD_DDP_TORCH_INIT=1 runs torch's constructor instead (then captured steps need the bare model)."""
        import os
        self.light = os.environ.get("FOD_DDP_TORCH_INIT", "0") != "1"
        if self.light:
            torch.nn.Module.__init__(self)
            self.module = module
            self.process_group = process_group if process_group is not None else dist.group.WORLD
            self.device_ids = None
            self.output_device = None
            self.broadcast_buffers = False
            self.find_unused_parameters = False
            self.static_graph = False
            self._verify_and_broadcast()
        else:
            super().__init__(module, broadcast_buffers=False, bucket_cap_mb=bucket_cap_mb, find_unused_parameters=False,
                             process_group=process_group)
            from future_od.native import functional as Fn
            Fn.WGRADS.enabled = False    # torch's reducer hooks see gradients on arrival: none may be filled in later
        self.require_backward_grad_sync = False          # torch's reducer stays idle (as under no_sync())
        self.grad_reducer = GradientReducer(self.module.parameters(), self.process_group, bucket_cap_mb)
        self._sync_enabled = True

    def _verify_and_broadcast(self):
        """Every rank must hold the same parameter shapes (a mismatch would deadlock or corrupt the first collective);
        then rank 0's parameters and buffers go to everyone, as DistributedDataParallel's constructor does."""
        tensors = [p.data for p in self.module.parameters()] + [b.data for b in self.module.buffers()]
        sig = torch.tensor([len(tensors), sum(t.numel() for t in tensors)], dtype=torch.int64)
        dev = tensors[0].device if tensors else torch.device("cpu")
        backend = dist.get_backend(self.process_group)
        sig = sig.to(dev) if (backend == "nccl" or dev.type == "cuda") else sig
        lo, hi = sig.clone(), sig.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.process_group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.process_group)
        if not torch.equal(lo, hi):
            raise RuntimeError(f"FodDataParallel: ranks hold different models (tensors / elements: this rank "
                               f"{sig.tolist()}, min {lo.tolist()}, max {hi.tolist()})")
        src = dist.get_global_rank(self.process_group, 0) if self.process_group is not dist.group.WORLD else 0
        with torch.no_grad():
            for t in tensors:
                dist.broadcast(t, src=src, group=self.process_group)

    def no_sync(self):
        """Gradient accumulation without communication, as DistributedDataParallel.no_sync()."""
        return _NoSync(self)

    def forward(self, *inputs, **kwargs):
        from future_od.native import functional as Fn
        out = self.module(*inputs, **kwargs)
        red = self.grad_reducer
        red.enabled = self._sync_enabled and torch.is_grad_enabled()
        Fn.set_grad_sync(red if red.enabled else None)
        if red.enabled:
            for t in _tensors_requiring_grad(out):       # the loss: arms the end-of-backward reduction
                t.register_hook(_ArmHook(red))
        return out
