"""Hungarian-matched set loss on the HIP kernels.

Drop-in for reference future_od/models/set_criterion.py (`SetCriterion`) plus the matcher the
reference pulls from the absent ConditionalDETR submodule (`HungarianMatcher` / `build_matcher`,
called at reference set_criterion.py:182,204).  One pass over ALL decoder levels:

    cost matrices (one kernel, all levels)  ->  ONE device-to-host copy  ->  host LAP on worker
    threads (libfod_hip.so, same algorithm as scipy's)  ->  matches to device  ->  one loss kernel
    (focal + L1 + GIoU + cardinality + class error per level)  ->  one gradient kernel in backward.

The reference does this six times with six host syncs.
"""
import torch
import torch.distributed as dist
import torch.nn as nn
from torch.autograd import Function

from future_od.native import ops


class HungarianMatcher(nn.Module):
    """Optimal bipartite matching between queries and targets; focal-style class cost + L1 + GIoU."""

    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"

    @torch.no_grad()
    def match_levels(self, logits, boxes, packed, threads=8):
        """logits f32 [L,B,M,C], boxes f32 [L,B,M,4] -> match int32 [L,B,M] on device: GLOBAL target
        index (into the concatenated target list) or -1."""
        Lv, B, M, _ = logits.shape
        sizes = packed["sizes"]
        ld = max(max(sizes), 1)
        cost = ops.match_cost(logits, boxes, packed["labels"], packed["boxes"], packed["offset"], ld,
                              self.cost_class, self.cost_bbox, self.cost_giou)
        if cost.device.type == "cuda":       # the one host sync of the step: D2H through pinned memory
            cost_h = torch.empty(cost.shape, dtype=cost.dtype, pin_memory=True)
            cost_h.copy_(cost, non_blocking=True)
            torch.cuda.current_stream(cost.device).synchronize()
        else:
            cost_h = cost
        local = ops.lap_solve_batch_host(cost_h.view(Lv * B, M, ld), sizes * Lv, threads).view(Lv, B, M)
        off = packed["offset_cpu"][:B].view(1, B, 1)
        glob = torch.where(local >= 0, local + off, local)
        return _to_device_async(glob, logits.device), local

    @torch.no_grad()
    def match_levels_async(self, logits, boxes, packed, threads=8):
        """match_levels() without the host stall: the stream parks on a host flag while a worker thread waits for the
        cost matrices, solves the assignment problems and writes the matches into pinned memory; this thread goes on
        queueing the loss and the backward pass while the GPU is still in the forward pass (include/fod.h,
        fod_match_after_event).  Returns the device match tensor only."""
        Lv, B, M, _ = logits.shape
        sizes = packed["sizes"]
        ld = max(max(sizes), 1)
        cost = ops.match_cost(logits, boxes, packed["labels"], packed["boxes"], packed["offset"], ld,
                              self.cost_class, self.cost_bbox, self.cost_giou)
        return _async_lap(cost.device).submit(cost, sizes * Lv, packed["offset_cpu"][:B].repeat(Lv), threads)

    @torch.no_grad()
    def match_levels_dev(self, logits, boxes, packed):
        """match_levels() with nothing leaving the device: the cost matrices are solved by one wavefront per problem
        (fod_lap_solve_batch_dev: the host solver's algorithm, arithmetic and tie rules, bit-identical assignments),
        the target counts are read from the device-side offset table.  No copy, no host thread, no parked stream:
        the step is a plain launch sequence (and can be captured as a hipGraph, future_od/graph.py).  Failures
        (non-finite costs) are reported one step late through `_DeviceLapStatus`."""
        cost = ops.match_cost(logits, boxes, packed["labels"], packed["boxes"], packed["offset"], packed["ld"],
                              self.cost_class, self.cost_bbox, self.cost_giou)
        st = _lap_status(cost.device)
        st.check()
        match = ops.lap_solve_batch_dev(cost, packed["offset"], st.flag)
        st.publish()
        return match

    @torch.no_grad()
    def forward(self, outputs, targets):
        """Reference-shaped API: list of (idx_pred int64 ascending, idx_tgt int64) per sample."""
        packed = pack_targets(targets, outputs["pred_logits"].device)
        _, local = self.match_levels(outputs["pred_logits"].detach().float()[None].contiguous(),
                                     outputs["pred_boxes"].detach().float()[None].contiguous(), packed)
        out = []
        for b in range(local.shape[1]):
            m = local[0, b]
            i = torch.nonzero(m >= 0).flatten()
            out.append((i.to(torch.int64), m[i].to(torch.int64)))
        return out


class _AsyncLap:
    """One worker thread + one host flag per device.  Tickets increase by one per submission; the stream waits for
    flag >= ticket.  The host may run at most one step ahead of the GPU: submit() first joins the previous job (which
    also surfaces its errors), so pinned buffers and the optimizer's pointer tables are never reused while in flight."""

    def __init__(self, device):
        import concurrent.futures
        import ctypes as C
        from future_od.native import lib as L
        self.L, self.C = L, C
        self.device = device
        flag = C.c_void_p()
        L._plain_call("fod_host_flag_create", C.addressof(flag))
        self.flag = flag.value
        self.ticket = 0
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="fod-matcher")
        self.prev = None
        self.slots = [(None, 0), (None, 0)]

    def _slot(self, words):
        """Pinned, device-mapped host block for the matches of this ticket (two blocks alternate: block t is rewritten
        by the worker of ticket t+2 only after that ticket's cost matrices arrived, i.e. after copy t ran)."""
        i = self.ticket & 1
        if self.slots[i][1] < words:
            L, C = self.L, self.C
            if self.slots[i][0] is not None:
                torch.cuda.current_stream(self.device).synchronize()        # growing is rare: wait out its readers
                L._plain_call("fod_host_free", self.slots[i][0])
            p = C.c_void_p()
            cap = max(words, 4096)
            L._plain_call("fod_host_alloc", C.addressof(p), cap * 4)
            self.slots[i] = (p.value, cap)
        return self.slots[i][0]

    def _job(self, event, cost_h, n_cols, col_off, match_host, ticket, threads):
        L, C = self.L, self.C
        P, M, ld = cost_h.shape
        # one foreign call (the GIL is released for all of it): event wait, assignment, offsets, flag
        L._plain_call("fod_match_after_event", self.device.index or 0,
                      event.cuda_event if event is not None else None,
                      cost_h.data_ptr(), P, M, ld, n_cols.data_ptr(), col_off.data_ptr(), match_host, self.flag,
                      ticket, threads)

    def join(self):
        prev, self.prev = self.prev, None
        if prev is not None:
            prev.result()                      # raises the worker's FodError, one step late

    def submit(self, cost, n_cols, col_off, threads):
        Lv, B, M, ld = cost.shape
        L, C = self.L, self.C
        self.join()
        cost_h = torch.empty((Lv * B, M, ld), dtype=cost.dtype, pin_memory=True)
        cost_h.copy_(cost.view(Lv * B, M, ld), non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        n_cols = torch.tensor(n_cols, dtype=torch.int32)
        col_off = col_off.to(torch.int32).contiguous()
        self.ticket += 1
        match_host = self._slot(Lv * B * M)
        match = torch.empty((Lv, B, M), dtype=torch.int32, device=cost.device)
        stream = torch.cuda.current_stream(cost.device).cuda_stream
        inline = False
        import os
        for _ in range(0 if os.environ.get("FOD_ASYNC_MATCH") == "2" else 8):      # "2": always park (tests)
            if event.query():
                # the GPU has already delivered the cost matrices: the step is host-bound here, nothing to overlap --
                # solve inline (measured: parking + worker cost a host-bound step ~8 %, and gain a GPU-bound one 15 %)
                inline = True
                break
        if inline:
            self._job(None, cost_h, n_cols, col_off, match_host, self.ticket, threads)
        else:
            try:
                self.prev = self.pool.submit(self._job, event, cost_h, n_cols, col_off, match_host, self.ticket, threads)
            except BaseException:
                L._plain_call("fod_host_flag_set", self.flag, self.ticket)         # nothing may stay parked
                raise
            for fn in take_unrun_while_matching():       # independent GPU work, queued ahead of the parked wait
                fn()
            try:
                L._plain_call("fod_stream_wait_flag", self.flag, self.ticket, stream)
            except L.FodError:
                # the stream could not be parked: wait for the worker on the host instead (correct, one stall), and
                # keep to the host-synchronous matcher from now on
                self.join()
                _WAIT_SUPPORT[self.device.index or 0] = False
        # the host block is read when this kernel RUNS, i.e. after the worker has released the stream
        L._plain_call("fod_copy_from_host_i32", match_host, match.data_ptr(), Lv * B * M, stream)
        return match


class _DeviceLapStatus:
    """Failure word of the device solver: a device int the kernel sets, mirrored into pinned host memory by an
    asynchronous copy after every solve and looked at (without waiting) before the next one -- scipy would have
    raised inside the step; here the error surfaces one step late, or at `check(wait=True)`."""

    def __init__(self, device):
        self.flag = torch.zeros(1, dtype=torch.int32, device=device)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()

    def publish(self):
        self.host.copy_(self.flag, non_blocking=True)

    def check(self, wait=False):
        if wait:
            torch.cuda.current_stream(self.flag.device).synchronize()
        code = int(self.host[0])
        if code:
            self.host.zero_()
            self.flag.zero_()
            from future_od.native.lib import FodError
            raise FodError({1: "matcher: non-finite cost matrix", 2: "matcher: more targets than the cost matrix holds",
                            3: "matcher: infeasible cost matrix"}.get(code, f"matcher: device solver status {code}"))


_LAP_STATUS = {}


def _lap_status(device):
    key = (device.type, device.index)
    if key not in _LAP_STATUS:
        _LAP_STATUS[key] = _DeviceLapStatus(device)
    return _LAP_STATUS[key]


def check_device_matcher(device):
    """Raise the device solver's failure of an EARLIER launch, if its status word has arrived (non-blocking).  Eager steps
    look at the word before every solve; a replayed graph runs no Python, so future_od/graph.py calls this before each
    replay: a non-finite cost matrix surfaces one step late there too instead of at the end of the epoch."""
    st = _LAP_STATUS.get((torch.device(device).type, torch.device(device).index))
    if st is not None:
        st.check()


def device_matching_enabled(device):
    """The matcher runs entirely on the GPU by default (FOD_DEVICE_MATCH=0: host solver, asynchronous or blocking)."""
    import os
    return torch.device(device).type == "cuda" and os.environ.get("FOD_DEVICE_MATCH", "1") != "0"


# Work that does not depend on the matches (post-processing, AP bookkeeping) can be queued between the cost matrices'
# host copy and the parked wait: the GPU then runs it while the worker solves the assignments instead of idling.
_WHILE_MATCHING = []


def run_while_matching(fn):
    """Register `fn` (no arguments) to be called by the next asynchronous submission right before the stream parks.
    If that submission does not park (host-sync path, inline solve, CPU), the caller runs it itself: see
    `take_unrun_while_matching`."""
    _WHILE_MATCHING.append(fn)


def take_unrun_while_matching():
    fns = list(_WHILE_MATCHING)
    _WHILE_MATCHING.clear()
    return fns


_ASYNC_LAP = {}


def _async_lap(device):
    key = (device.type, device.index)
    if key not in _ASYNC_LAP:
        _ASYNC_LAP[key] = _AsyncLap(device)
    return _ASYNC_LAP[key]


def join_matchers():
    """Surface a matcher failure NOW instead of at the next submission: joins every parked host job and reads the device
    solver's status word after a stream sync.  The reference (scipy) raises inside the step; here the error normally
    arrives one step late, so the LAST iteration of an epoch or an evaluation would otherwise never report it
    (Trainer._run_epoch calls this when its loop ends)."""
    for lap in list(_ASYNC_LAP.values()):
        lap.join()
    for st in list(_LAP_STATUS.values()):
        st.publish()
        st.check(wait=True)


_WAIT_SUPPORT = {}


def _stream_wait_supported(device):
    idx = torch.device(device).index or 0
    if idx not in _WAIT_SUPPORT:
        from future_od.native import lib as L
        _WAIT_SUPPORT[idx] = bool(L._ENTRY["fod_stream_wait_supported"](idx))      # a capability, not a status code
    return _WAIT_SUPPORT[idx]


def async_matching_enabled(device):
    """The asynchronous matcher needs a GPU stream to park; FOD_ASYNC_MATCH=0 restores the host sync, =2 parks even
    when the cost matrices have already arrived (so that tests reach the worker path on tiny inputs)."""
    import os
    return (torch.device(device).type == "cuda" and os.environ.get("FOD_ASYNC_MATCH", "1") != "0"
            and _stream_wait_supported(device))


def build_matcher(args):
    return HungarianMatcher(cost_class=args.set_cost_class, cost_bbox=args.set_cost_bbox,
                            cost_giou=args.set_cost_giou)


def _to_device_async(t, device):
    """Host tensor -> device through pinned memory, without blocking the host on the stream."""
    if t.device.type != "cpu" or device.type == "cpu":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def pack_targets(targets, device):
    """list of {"labels","boxes"} -> concatenated device tensors + per-sample offsets.  Host targets (the usual
    case, see st_detr.forward) are packed on the host and uploaded asynchronously; device targets cost one sync."""
    device = torch.device(device)
    sizes = [int(t["labels"].shape[0]) for t in targets]
    labels = torch.cat([t["labels"] for t in targets]).to(dtype=torch.int64).contiguous()
    boxes = torch.cat([t["boxes"] for t in targets]).to(dtype=torch.float32).contiguous()
    off = [0]
    for s in sizes:
        off.append(off[-1] + s)
    off_cpu = torch.tensor(off, dtype=torch.int32)
    if labels.numel() == 0:          # keep pointers valid for the kernels
        labels = torch.zeros(1, dtype=torch.int64, device=labels.device)
        boxes = torch.zeros((1, 4), dtype=torch.float32, device=boxes.device)
    return {"sizes": sizes, "labels": _to_device_async(labels, device), "boxes": _to_device_async(boxes, device),
            "offset": _to_device_async(off_cpu, device), "offset_cpu": off_cpu}


class _SetLossFn(Function):
    """(logits, boxes) [L,B,M,*] f32 -> table f32 [L,5] = ce, bbox, giou, cardinality_error, class_error."""

    @staticmethod
    def forward(ctx, logits, boxes, match, packed, num_boxes, alpha):
        table = ops.set_loss_fwd(logits, boxes, match, packed["labels"], packed["boxes"], packed["offset"],
                                 num_boxes, alpha)
        ctx.save_for_backward(logits, boxes, match)
        ctx.packed, ctx.num_boxes, ctx.alpha = packed, num_boxes, alpha
        return table

    @staticmethod
    def backward(ctx, dtable):
        logits, boxes, match = ctx.saved_tensors
        g = dtable[:, :3].contiguous()
        dl, db = ops.set_loss_bwd(logits, boxes, match, ctx.packed["labels"], ctx.packed["boxes"], g,
                                  ctx.num_boxes, ctx.alpha)
        return dl, db, None, None, None, None


_HOST_GROUP = []


def _host_group():
    """Process group for host-side scalar collectives: the default group when it is gloo (CPU tests, rehearsal),
    else a gloo group created once next to the RCCL one (every rank reaches this at its first forward)."""
    if not _HOST_GROUP:
        if dist.get_backend() == "gloo":
            _HOST_GROUP.append(None)
        else:
            group, err = "device", None
            try:
                group = dist.new_group(backend="gloo")
            except Exception as e:                       # no usable host transport on THIS rank
                err = e
            # new_group is a collective but fails rank by rank: a rank that fell back to the device group while the
            # others all-reduce on gloo would hang the first forward.  Agree on the outcome over the default group.
            ok = torch.tensor([0.0 if err is not None else 1.0], device=torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0:
                print(f"[fod] no gloo group for host-side scalars on every rank ({err}); "
                      "using a device all-reduce (one host stall per step)")
                group = "device"
            _HOST_GROUP.append(group)
    return _HOST_GROUP[0]


class LossDict(dict):
    """The reference's dict of named losses, plus the [levels, 5] table they are views of."""
    table = None
    level_of = None


class SetCriterion(nn.Module):
    """Reference set_criterion.py:11-217 (labels, boxes, cardinality; masks are not on the path)."""

    def __init__(self, num_classes, matcher, weight_dict, focal_alpha, losses, matching_mode):
        super().__init__()
        self.num_classes = num_classes
        self.matcher = matcher
        self.weight_dict = weight_dict
        self.losses = losses
        self.focal_alpha = focal_alpha
        assert matching_mode in ("per level", "last level")
        self._matching_mode = matching_mode
        for name in losses:
            assert name in ("labels", "boxes", "cardinality"), f"do you really want to compute {name} loss?"

    def global_num_boxes(self, targets, device, distributed, lazy=False):
        """set_criterion.py:185-193: mean number of boxes per rank, clamped to >= 1.

        The count is known on the host before the forward pass, so the all-reduce is a HOST collective (a gloo group
        beside the RCCL one): a device all-reduce would have to be read back, and that read-back is a host stall on
        the whole forward pass now that the matcher no longer stalls the host.  Costs ~0.1 ms of host time."""
        n = float(sum(int(t["labels"].shape[0]) for t in targets))
        if not distributed:
            return max(n, 1.0)
        group = _host_group()
        if group == "device":
            t = _to_device_async(torch.tensor([n], dtype=torch.float32), torch.device(device))
            dist.all_reduce(t)
        else:
            t = torch.tensor([n], dtype=torch.float64)
            dist.all_reduce(t, group=group)
        return max(float(t.item()) / dist.get_world_size(), 1.0)

    def device_num_boxes(self, count, distributed):
        """set_criterion.py:185-193 without the `.item()`: `count` is the device-side total of this rank's targets
        (ops.pack_targets_dev); the normaliser stays a device scalar that the loss kernels read when they run."""
        if distributed:
            count = count.clone()
            dist.all_reduce(count)
            count = count / dist.get_world_size()
        return torch.clamp(count, min=1.0)

    def forward(self, outputs, targets, distributed, packed=None, num_boxes=None):
        if "_stacked" in outputs:
            logits, boxes = outputs["_stacked"]
        else:
            levels = list(outputs.get("aux_outputs", [])) + [outputs]
            logits = torch.stack([o["pred_logits"].float() for o in levels])
            boxes = torch.stack([o["pred_boxes"].float() for o in levels])
        logits, boxes = logits.contiguous(), boxes.contiguous()
        Lv = logits.shape[0]
        if packed is None:
            packed = pack_targets(targets, logits.device)
        if num_boxes is None:
            num_boxes = self.global_num_boxes(targets, logits.device, distributed)
        levels = (lambda t: t.detach()) if self._matching_mode == "per level" else (lambda t: t.detach()[-1:].contiguous())
        if packed.get("device"):
            match = self.matcher.match_levels_dev(levels(logits), levels(boxes), packed)
        elif async_matching_enabled(logits.device):
            match = self.matcher.match_levels_async(levels(logits), levels(boxes), packed)
        elif self._matching_mode == "per level":
            match, _ = self.matcher.match_levels(logits.detach(), boxes.detach(), packed)
        else:
            match, _ = self.matcher.match_levels(logits.detach()[-1:], boxes.detach()[-1:], packed)
        if match.shape[0] != Lv:
            match = match.expand(Lv, -1, -1).contiguous()
        table = _SetLossFn.apply(logits, boxes, match, packed, num_boxes, self.focal_alpha)
        out = LossDict()
        out.table = table
        for lv in range(Lv):
            sfx = "" if lv == Lv - 1 else f"_{lv}"
            if "labels" in self.losses:
                out["loss_ce" + sfx] = table[lv, 0]
                if lv == Lv - 1:
                    out["class_error"] = table[lv, 4].detach()
            if "cardinality" in self.losses:
                out["cardinality_error" + sfx] = table[lv, 3].detach()
            if "boxes" in self.losses:
                out["loss_bbox" + sfx] = table[lv, 1]
                out["loss_giou" + sfx] = table[lv, 2]
        return out
