"""A whole training step as ONE hipGraph: zero_grad, forward (backbone, encoder, decoder, device-side Hungarian
matching, set loss, post-processing / AP bookkeeping), backward, gradient clipping and AdamW.

Why: the reference's step (future_od/trainer.py:171-189) is ~1400 kernel launches behind ~16 us of Python each
(autograd Function bookkeeping, output allocation, shape checks); every BASELINE configuration except the headline
was bound by the launching thread (~21 ms per step whatever the image size).  Replaying a captured step
costs the host one call.  What made the step capturable:

  * the matcher never leaves the device (fod_pack_targets, fod_match_cost, fod_lap_solve_batch_dev): no D2H copy, no
    scipy, no parked stream, target counts and `num_boxes` read from device memory;
  * the optimizer's step count / bias corrections live on the device (FusedAdamW.enable_device_step), learning rates
    are read from a device table that `sync_hyperparams` rewrites in place;
  * every kernel takes raw pointers + the stream, allocates nothing and keeps no state (include/fod.h conventions).

Data parallel (`GraphedStep(bare_model, opt, data_parallel=True)`): the step is TWO graphs with the collectives between them, launched eagerly --
    [graph A: zero_grad, forward, backward]  ->  all_reduce(AVG) of the flat gradient arena (one large message, plus the
    few gradients autograd allocated itself)  ->  [graph B: clip + AdamW]
(FOD_GRAPH_OVERLAP=1 splits graph A at the backbone's output and sends the transformer's gradients while the backbone's
backward runs; off by default, see __init__).
No collective is captured (RCCL inside a hipGraph could not be verified on the one-GPU boxes this was built on; a hang
there would void a whole scaling run): the loss normaliser `num_boxes`, the only collective of the forward pass, is
all-reduced from the batch's annotations BEFORE graph A and fed in as an input (`data["_num_boxes"]`).  The gradient
average is not overlapped with the backward pass (215 MB over xGMI, ~1-2 ms at 8 GPUs), which costs less than the ~6 ms
per step the eagerly launched data-parallel step loses to the launching thread (bench.py --no-graph).

Dropout (train mode, p > 0): the masks are stateless hashes of (seed, index); a captured kernel carries its call's
seed as a constant, so the kernels additionally mix in a device scalar (`ops.DROP_BASE`) that the graph advances once
per replay -- new masks every step, the same masks in the forward and the backward of one step.

Restrictions (checked): one device per process, static shapes -- a batch with other shapes is re-captured.
"""
import torch
import torch.distributed as dist

from future_od.native import functional as Fn
from future_od.native import ops


def _stage_inputs(g, data):
    """Copy the batch into the graph's input buffers -- except entries that are the very tensors staged last time and
    have not been written since (same object, same version counter): a resident, unchanged batch costs nothing."""
    seen = g.setdefault("staged", {})
    for k, v in g["static"].items():
        if isinstance(v, torch.Tensor) and k in data:
            src = data[k]
            if src.data_ptr() == v.data_ptr():
                continue
            tag = (id(src), src._version)
            if seen.get(k) == tag:
                continue
            v.copy_(src, non_blocking=True)
            seen[k] = tag
            g.setdefault("keep", {})[k] = src        # the id stays unique while the tensor is alive



class CaptureError(RuntimeError):
    """The step could not be CAPTURED (raised by GraphedStep / GraphedForward at the first call for a batch signature,
    with parameters and optimizer state rolled back): the caller may launch eagerly instead.  Errors of a replay are not
    of this type and are not to be swallowed."""


class _Bf16Pending:
    """An asynchronous bf16 all-reduce whose result still has to go back into the f32 gradient tensor."""

    def __init__(self, work, buf, dst, scale):
        self.work, self.buf, self.dst, self.scale = work, buf, dst, scale

    def wait(self):
        self.work.wait()
        self.dst.copy_(self.buf)
        if self.scale != 1.0:
            self.dst.mul_(self.scale)


class GraphedStep:
    """step = GraphedStep(model, optimizer);  post, loss, stats, od = step(data)

    The first call with a given batch signature runs `warmup` eager steps (they are real optimizer steps: nothing is
    rolled back) and captures one more; later calls copy the batch into the captured input buffers and replay.
    Returned tensors are the graph's static outputs: valid until the next call."""

    def __init__(self, model, optimizer, warmup=2, process_group=None, data_parallel=False, rollback_warmup=False):
        """`data_parallel=True` (or a `process_group`): `model` is the BARE model of this rank -- not a
        DistributedDataParallel wrapper: torch's wrapper stashes every parameter's AccumulateGrad node at construction,
        on the default stream, and a backward pass captured on another stream must not touch the default stream.  The
        ranks' parameters must be equal on entry (`broadcast_parameters`)."""
        self.model, self.opt, self.warmup = model, optimizer, max(int(warmup), 2)
        self.core = model
        self.group = process_group
        self.ddp = bool(data_parallel or process_group is not None)
        if getattr(model, "light", False) and hasattr(model, "module"):
            # FodDataParallel built without torch's reducer (parallel.py): nothing of it lives on the default stream, so
            # the wrapped model can be captured; its own reducer is simply not armed (the model is called directly)
            self.core, self.group, self.ddp = model.module, model.process_group, True
        self.world = dist.get_world_size(self.group) if self.ddp else 1
        self._graphs = {}
        self.replays = 0
        self.comm_stats = {"tensors": 0, "bytes": 0}           # what one step all-reduces (data parallel)
        # the eager warm-up steps before a capture are real optimizer steps on the batch that triggered it; with
        # rollback_warmup the parameters, moments and step counts are put back afterwards, so that a training loop
        # sees exactly one update per batch (the Trainer asks for this; the benchmark does not care)
        self.rollback_warmup = bool(rollback_warmup)
        # data parallel: split the captured backward at the backbone's output and average the transformer gradients
        # (120 MB) on RCCL's stream while the backbone's backward graph replays (north_star: "all-reduce ... overlapped
        # with backward on a side HIP stream").  What one rank can measure (tools/ddp_overlap_probe.py, one-rank RCCL
        # group, same process, interleaved): the split itself is free (A1 + A2 + B without communication = A + B: 20.92
        # vs 20.96 ms), the same collectives issued AFTER A2 cost what the unsplit step's do (21.35 vs 21.34), running the
        # transformer's collective BESIDE A2 costs +0.6 ms (21.96) whatever NCCL_MAX_NCHANNELS is -- but a one-rank
        # "all-reduce" is a full-chip element-wise kernel (oneRankReduce, 16384 x 512 threads, 0.3 ms) fighting the
        # backbone's kernels for every CU, where a ring step of a real group runs on a few channels.  So: on for
        # world > 1 with RCCL (FOD_GRAPH_OVERLAP=0 turns it off), off at one rank and for gloo (a two-rank gloo
        # rehearsal with full-size gradients stalled in the asynchronous collective, 6.7 s per step, round 2).
        import os
        env = os.environ.get("FOD_GRAPH_OVERLAP")
        # default (round 3): ON for a real multi-rank RCCL group, off for one rank and for gloo -- see the comment above
        # and profiles/r03e_ddp_overlap_probe_1rank.txt
        self.overlap = (env == "1") if env is not None else (self.ddp and self.world > 1 and self._native_avg())
        self.overlap_mode = os.environ.get("FOD_GRAPH_OVERLAP_MODE", "async")     # "after": experiment, see __call__
        self.grad_bf16 = os.environ.get("FOD_GRAD_BF16", "0") == "1"
        self._bf16_bufs = {}

    def broadcast_parameters(self, src=0):
        """Rank `src`'s parameters and buffers to every rank (what DistributedDataParallel's constructor does)."""
        with torch.no_grad():
            for t in list(self.core.parameters()) + list(self.core.buffers()):
                dist.broadcast(t.data, src=src, group=self.group)

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _signature(data):
        return tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in data.items() if isinstance(v, torch.Tensor)))

    def _check(self, data):
        if (isinstance(self.core, torch.nn.parallel.DistributedDataParallel)
                or (isinstance(self.model, torch.nn.parallel.DistributedDataParallel) and self.core is self.model)):
            raise RuntimeError("GraphedStep: pass the bare model (data_parallel=True / process_group=...), not a "
                               "DistributedDataParallel wrapper: its stashed AccumulateGrad nodes live on the default "
                               "stream and its reducer's collectives are not captured")
        for k, v in data.items():
            if isinstance(v, torch.Tensor) and not v.is_cuda:
                raise RuntimeError(f"GraphedStep: batch entry {k!r} is not on the device")

    # ---- the step, eager (warm-up; data parallel: the same pieces as the replayed step, so the ranks stay in step) ----
    def _eager(self, data):
        if self.ddp:
            if "_num_boxes" not in data:
                dev = data["active"].device
                data["_num_boxes"] = torch.ones(1, dtype=torch.float32, device=dev)
            self._global_num_boxes(data, data["_num_boxes"])
        outs = self._forward_backward(data)
        if self.ddp:
            for t in self._reduce_targets():
                self._all_reduce(t)
        self.opt.step()
        return outs

    # ---- the pieces that are captured ------------------------------------------------------------------------------
    def _dropout_active(self):
        if not self.core.training:
            return False
        drops = [m.p for m in self.core.modules() if isinstance(m, torch.nn.Dropout)]
        rates = [float(getattr(m, "droprate", 0.0) or 0.0) for m in self.core.modules()]
        return any(p > 0 for p in drops + rates)

    def _forward_backward(self, data):
        if ops.DROP_BASE is not None:
            # train mode: the dropout kernels mix this device scalar into the seeds baked into the graph, so every
            # replay draws new masks (forward and backward of one step read the same value)
            ops.DROP_BASE.add_(1)
        self.opt.zero_grad()
        Fn.set_grad_sync(None)          # an eager pass through a FodDataParallel wrapper may have left its reducer installed
        post, _state, loss, stats, od = self.core(data=data, distributed=False)
        loss.backward()
        return post, loss, stats, od

    def _global_num_boxes(self, data, out):
        """reference set_criterion.py:185-193: the number of target boxes averaged over the ranks, at least 1 -- from the
        batch's annotations, on the device, nothing read back."""
        n = (data["active"] == 1).sum().to(torch.float32).view(1)
        self._all_reduce(n, average=True)
        out.copy_(n.clamp_(min=1.0))

    def _native_avg(self):
        return dist.get_backend(self.group) == "nccl"

    def _all_reduce(self, t, average=True, async_op=False):
        group = self.group
        if self.grad_bf16 and average and t.dtype == torch.float32 and t.numel() >= (1 << 16):
            return self._all_reduce_bf16(t, async_op)
        if average and self._native_avg():
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group, async_op=async_op)
        # gloo (rehearsal on one GPU): sum, then scale (an asynchronous call leaves the scaling to its caller).  The
        # stream is drained first: a gloo collective issued behind a queued graph replay took seconds instead of 60 ms
        # (per-phase timings with FOD_GRAPH_TIMING=1); RCCL orders itself on the stream and needs no such wait
        if t.is_cuda and not async_op:
            torch.cuda.current_stream(t.device).synchronize()
        work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if average and not async_op:
            t.div_(self.world)
        return work

    def _all_reduce_bf16(self, t, async_op):
        """FOD_GRAD_BF16=1 ("bf16 gradient buckets halve it to 107.8 MB"): the large f32 gradient tensors
        travel as bf16 -- cast, average, cast back into the f32 tensor the optimizer reads.  Every rank ends with the
        SAME f32 values (the average of the ranks' bf16-rounded gradients, rounded to bf16 once more by the collective's
        output type), so replicas stay bit-equal; what changes is the gradient's precision (8 significant bits), which
        is why this is a switch and not the default."""
        buf = self._bf16_bufs.get(t.data_ptr())
        if buf is None or buf.numel() != t.numel():
            buf = self._bf16_bufs[t.data_ptr()] = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
        buf.copy_(t)
        if self._native_avg():
            work = dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        else:
            if not async_op:
                torch.cuda.current_stream(t.device).synchronize()
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            return _Bf16Pending(work, buf, t, 1.0 if self._native_avg() else 1.0 / self.world)
        t.copy_(buf)
        if not self._native_avg():
            t.div_(self.world)
        return None

    def _reduce_targets(self, arena_from=0, skip=()):
        """What a captured backward piece leaves to be averaged: the arena region that holds gradients as ONE flat tensor
        (everything in it that is not a gradient is a dead temporary: averaging it is harmless) plus the gradients that
        live elsewhere (sums autograd allocated itself).  Static addresses: every replay rewrites the same memory.
        `arena_from` / `skip`: only arena gradients from that element offset on, none of the tensors whose data_ptr is
        in `skip` -- the second piece of a split backward."""
        grads = [p.grad for p in self.core.parameters() if p.requires_grad and p.grad is not None]
        buf = Fn.ARENA.buf if Fn.ARENA.active else None
        targets, lo, hi = [], None, None
        for g in grads:
            inside = (buf is not None and g.is_cuda and g.dtype == torch.float32
                      and buf.data_ptr() <= g.data_ptr() < buf.data_ptr() + 4 * buf.numel())
            if inside:
                # the extent of the gradient's storage footprint inside the arena (channels_last views included)
                first = (g.data_ptr() - buf.data_ptr()) // 4
                if first < arena_from:
                    continue
                last = first + sum((n - 1) * s for n, s in zip(g.shape, g.stride())) + 1
                lo = first if lo is None else min(lo, first)
                hi = last if hi is None else max(hi, last)
            elif g.data_ptr() not in skip:
                targets.append(g)
        if lo is not None:
            targets.insert(0, buf[lo:hi])
        return targets

    def _snapshot(self):
        opt = self.opt
        params = [p for grp in opt.param_groups for p in grp["params"]]
        return {"params": [(p, p.detach().clone()) for p in params],
                "moments": {p: (st["exp_avg"].clone(), st["exp_avg_sq"].clone())
                            for p, st in opt.state.items() if "exp_avg" in st},
                "step_no": getattr(opt, "_step_no", 0)}

    def _restore(self, snap):
        opt = self.opt
        with torch.no_grad():
            for p, saved in snap["params"]:
                p.copy_(saved)
            for p, st in opt.state.items():
                if "exp_avg" in st:
                    old = snap["moments"].get(p)
                    if old is None:                      # created by the warm-up: back to "never stepped"
                        st["exp_avg"].zero_()
                        st["exp_avg_sq"].zero_()
                    else:
                        st["exp_avg"].copy_(old[0])
                        st["exp_avg_sq"].copy_(old[1])
        opt._step_no = snap["step_no"]
        if getattr(opt, "_dev_step", None) is not None:
            opt._dev_step.fill_(float(snap["step_no"]))
        Fn.PREP.mark_stale()

    def _capture(self, data):
        # the capture's eager warm-up steps are real optimizer steps: with rollback_warmup they are undone whether the
        # capture succeeds or not -- a caller that falls back to the eager step after a failed capture must not apply a
        # batch three or four times
        snap = self._snapshot() if self.rollback_warmup else None
        try:
            out = self._capture_inner(data)
        finally:
            if snap is not None:
                self._restore(snap)
        # the optimizer tables this graph's AdamW node reads: learning-rate changes go into THAT plan (__call__), and
        # the record keeps it alive for as long as the graph lives
        out["plan"] = getattr(self.opt, "_launched_plan", None)
        return out

    def _capture_agreed(self, data):
        """_capture(), with every failure INSIDE the capture (and only those: replay-time errors are not caught anywhere)
        turned into CaptureError -- after, under data parallel, the ranks have agreed on the outcome: a rank that alone
        fell back to the eagerly launched step would issue other collectives than its peers."""
        err, g = None, None
        try:
            g = self._capture(data)
        except Exception as e:                            # noqa: BLE001  (re-raised below as CaptureError)
            err = e
        if self.ddp:
            import torch.distributed as dist
            dev = next(v for v in data.values() if isinstance(v, torch.Tensor)).device
            ok = torch.tensor([0.0 if err is not None else 1.0], device=dev if dist.get_backend() != "gloo" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 0.0 and err is None:
                err = RuntimeError("another rank could not capture the step")
        if err is not None:
            raise CaptureError(f"{type(err).__name__}: {err}") from err
        return g

    def _capture_inner(self, data):
        self._check(data)
        dev = next(v for v in data.values() if isinstance(v, torch.Tensor)).device
        static = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in data.items()
                  if k != "_host_annotations"}
        if self._dropout_active() and (ops.DROP_BASE is None or ops.DROP_BASE.device != dev):
            ops.DROP_BASE = torch.zeros(1, dtype=torch.int64, device=dev)
        side = _warmup_stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(self.warmup):          # populates every cache (prepared-weight tables, optimizer plan, ...)
                self._eager(static)
            if getattr(self.opt, "_dev_step", None) is None:
                self.opt.enable_device_step(dev)
                self._eager(static)               # one eager step through the device-side bias corrections
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        Fn.WGRADS.prepare(dev)                    # pinned job tables for the captures below (none can be made inside)
        ops.preallocate_graph_workspaces(dev)     # scratch of the captured launches: allocated and zeroed out here
        if not self.ddp:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = self._forward_backward(static)
                self.opt.step()
            torch.cuda.synchronize(dev)
            # the capture itself executed nothing: parameters, moments and the device step count are those after the
            # warm-up steps; python's step counter ran one ahead during capture
            self.opt._step_no -= 1
            return {"graph": graph, "static": static, "outs": outs}
        # data parallel: the collectives stay outside the two graphs
        self._global_num_boxes(static, static["_num_boxes"])
        torch.cuda.synchronize(dev)
        # The backward pass in two pieces (native/functional.py: BackboneCut): A1 = forward + everything down to the
        # backbone's output (all transformer gradients), A2 = the backbone's backward.  The transformer gradients are
        # all-reduced while A2 runs -- the overlap the eager reducer has (parallel.py), without a collective in a graph.
        graph_a = torch.cuda.CUDAGraph()
        cut = Fn.BackboneCut() if self.overlap else None
        Fn.BACKBONE_CUT = cut
        try:
            # thread-local capture mode: the process group's watchdog thread polls its events while this thread captures
            with torch.cuda.graph(graph_a, capture_error_mode="thread_local"):
                outs = self._forward_backward(static)
        finally:
            Fn.BACKBONE_CUT = None
        first = self._reduce_targets()
        graph_a2, second = None, []
        if cut is not None and cut.pairs:
            off1 = Fn.ARENA.off if Fn.ARENA.active else 0
            graph_a2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_a2, pool=graph_a.pool(), capture_error_mode="thread_local"):
                cut.finish()
            second = self._reduce_targets(arena_from=off1, skip={t.data_ptr() for t in first})
        graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph_b, pool=graph_a.pool(), capture_error_mode="thread_local"):
            self.opt.step()
        torch.cuda.synchronize(dev)
        self.opt._step_no -= 1
        both = first + second
        self.comm_stats = {"tensors": len(both), "bytes": sum(t.numel() * t.element_size() for t in both),
                           "overlapped_bytes": sum(t.numel() * t.element_size() for t in first) if graph_a2 else 0}
        return {"graph": graph_a, "graph_bb": graph_a2, "graph_opt": graph_b, "targets": first, "targets_bb": second,
                "static": static, "outs": outs}

    # ------------------------------------------------------------------------------------------
    def __call__(self, data, sync=True):
        """One optimizer step on `data`.  `sync=False` (data parallel only) skips the gradient average -- the ranks'
        parameters then DIVERGE; it exists to time the step without communication."""
        token = getattr(self.opt, "_state_token", 0)
        if token != getattr(self, "_opt_token", token):
            self._retired_graphs = getattr(self, "_retired_graphs", []) + [self._graphs]
            self._graphs = {}                      # optimizer.load_state_dict() replaced the moment tensors: capture anew
        self._opt_token = token
        sig = self._signature(data)
        g = self._graphs.get(sig)
        if g is None:
            g = self._graphs[sig] = self._capture_agreed(data)
        _check_matcher(g)                                 # a failure of the previous replay's device solver: raised here
        _stage_inputs(g, data)
        self.opt.sync_hyperparams()                       # the current plan (eager steps) ...
        if g.get("plan") is not None:
            self.opt.sync_hyperparams(g["plan"])          # ... and the one baked into this graph, if it is another
        import os, time
        dbg = os.environ.get("FOD_GRAPH_TIMING") == "1"
        def tick(label, _t=[time.perf_counter()]):
            if dbg:
                torch.cuda.synchronize()
                now = time.perf_counter()
                print(f"[graph timing] {label}: {(now - _t[0]) * 1e3:.1f} ms", flush=True)
                _t[0] = now
        tick("staged")
        if self.ddp:
            self._global_num_boxes(g["static"], g["static"]["_num_boxes"])
        tick("num_boxes")
        g["graph"].replay()
        tick("graph A")
        if self.ddp:
            pending = []
            if sync:
                if g["graph_bb"] is not None and self.overlap_mode == "after":
                    pass                             # (experiment: split graphs, every collective behind A2)
                elif g["graph_bb"] is not None:
                    # issued behind graph A1 on the collective's own stream: runs while graph A2 does
                    pending = [self._all_reduce(t, async_op=True) for t in g["targets"]]
                else:
                    for t in g["targets"]:
                        self._all_reduce(t)
                        tick(f"all_reduce {t.numel()} elements")
            if g["graph_bb"] is not None:
                g["graph_bb"].replay()
                if sync:
                    if self.overlap_mode == "after":
                        for t in g["targets"]:
                            self._all_reduce(t)
                    for t in g["targets_bb"]:
                        self._all_reduce(t)
                    for w, t in zip(pending, g["targets"]):
                        w.wait()                     # stream-ordered for RCCL: the launching stream waits, not the host
                        if not self._native_avg() and not isinstance(w, _Bf16Pending):
                            t.div_(self.world)       # (gloo sums; a bf16 hand-off scales in its own wait)
            g["graph_opt"].replay()
        self.opt._step_no += 1
        self.replays += 1
        Fn.PREP.mark_stale()          # the replay changed the parameters without bumping their python-side versions
        return g["outs"]


def _check_matcher(g):
    from future_od.models.set_criterion import check_device_matcher
    dev = next(v for v in g["static"].values() if isinstance(v, torch.Tensor)).device
    check_device_matcher(dev)


_WARMUP_STREAMS = {}


def _warmup_stream(dev):
    """One side stream per device for every capture's eager warm-up steps (torch wants them off the default stream): the
    per-stream scratch buffers of native/ops.py are then allocated once, not once per capture."""
    key = (dev.type, dev.index)
    if key not in _WARMUP_STREAMS:
        _WARMUP_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _WARMUP_STREAMS[key]


class GraphedForward:
    """out = GraphedForward(model)(data): the evaluation pass (`with torch.no_grad(): model(data)`; forward, set loss,
    post-processing, AP bookkeeping) as one hipGraph per batch signature.  The reference's evaluation loop
    (trainer.py:171-178 without the backward) is ~450 launches behind ~16 us of Python each: bound by the launching
    thread on all but the largest configuration.  Parameters are read when the graph RUNS, so a graph captured once stays
    valid while training updates them in place; the compute-dtype weight copies the kernels read are refreshed (one eager
    launch) before a replay whenever a parameter has changed since.  Returns (post, loss, stats, od): the graph's static outputs, valid until the next call."""

    def __init__(self, model, warmup=1):
        self.model, self.warmup = model, max(int(warmup), 1)
        self._graphs = {}
        self.replays = 0

    def _run(self, data):
        with torch.no_grad():
            post, _state, loss, stats, od = self.model(data=data, distributed=False)
        return post, loss, stats, od

    def _capture(self, data):
        dev = next(v for v in data.values() if isinstance(v, torch.Tensor)).device
        static = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in data.items()
                  if k != "_host_annotations"}
        side = _warmup_stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(self.warmup + 1):
                # the second pass refreshes every prepared weight copy eagerly: its job table (pinned upload) must
                # exist before the capture -- nothing may be allocated on the host inside one
                self._run(static)
                Fn.PREP.mark_stale()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        ops.preallocate_graph_workspaces(dev)
        Fn.PREP.refresh()                         # the graph reads the prepared weight copies; they are refreshed
        with torch.cuda.graph(graph):             # eagerly before a replay, and only when a parameter has changed
            outs = self._run(static)
        torch.cuda.synchronize(dev)
        return {"graph": graph, "static": static, "outs": outs}

    def __call__(self, data):
        if self.model.training:
            raise RuntimeError("GraphedForward: evaluation only (model.eval()); training steps go through GraphedStep")
        sig = GraphedStep._signature(data)
        g = self._graphs.get(sig)
        if g is None:
            for k, v in data.items():
                if isinstance(v, torch.Tensor) and not v.is_cuda:
                    raise RuntimeError(f"GraphedForward: batch entry {k!r} is not on the device")
            try:
                g = self._capture(data)
            except Exception as e:                    # noqa: BLE001
                raise CaptureError(f"{type(e).__name__}: {e}") from e
            self._graphs[sig] = g
        _check_matcher(g)
        _stage_inputs(g, data)
        Fn.PREP.refresh()                             # one launch if an optimizer step happened since, nothing otherwise
        g["graph"].replay()
        self.replays += 1
        return g["outs"]
