"""Autograd glue: each Function's forward and backward are calls into libfod_hip.so.

Parameters stay fp32 `nn.Parameter`s with the reference's names and shapes; kernels consume
prepared copies (cast to the compute dtype, transposed for input-gradients, BN scale folded for
convs) that are cached per parameter version, so a copy is rebuilt only after an optimizer step.
Activations and their gradients are in the compute dtype (torch.float32 = parity mode,
torch.bfloat16 = performance mode); parameter gradients are fp32.
"""
import os

import torch
from torch.autograd import Function

from . import lib as L
from . import ops

_VEC = {torch.float32: 4, torch.bfloat16: 8}


# ------------------------------------------------------------------------------------------------
# prepared-weight registry
#
# Every kernel operand derived from a parameter (compute-dtype copy, transposed copy for input gradients, BN
# scale folded in, several Linear layers stacked) is declared ONCE as a list of strided-copy jobs into persistent
# buffers.  After an optimizer step (parameter `_version` bumped) the first request for any of them refreshes ALL
# stale ones with a single `fod_multi_permute3` launch (one job table on the device, cached while the set is the
# same) instead of ~75 small launches scattered over the next forward.
# ------------------------------------------------------------------------------------------------
class _Job:
    __slots__ = ("src", "dst", "dims", "sstr", "dstr", "valid1", "valid2", "scale", "axis")

    def __init__(self, src, dst, dims, sstr, dstr=None, valid1=None, valid2=None, scale=None, axis=-1):
        d0, d1, d2 = dims
        self.src, self.dst, self.dims, self.sstr = src, dst, dims, sstr
        self.dstr = dstr if dstr is not None else (d1 * d2, d2)
        self.valid1 = d1 if valid1 is None else valid1
        self.valid2 = d2 if valid2 is None else valid2
        self.scale, self.axis = scale, axis


class _Entry:
    __slots__ = ("params", "vers", "jobs", "value", "epoch")


class _Prepared:
    def __init__(self):
        self._store = {}
        self._tables = {}
        self._chunk = None
        # Bumped by every writer that changes parameters WITHOUT going through torch (the fused AdamW kernel and a
        # replayed hipGraph write through raw pointers: `_version` does not move).  An operand is fresh only if both
        # its parameters' versions and this epoch are the ones it was built at.
        self.epoch = 0

    def _fresh(self, e):
        # (operands of frozen parameters -- stem, layer1, every frozen-BN fold -- are not touched by an optimizer)
        return (e.epoch == self.epoch or e.epoch == -2) and e.vers == tuple(p._version for p in e.params)

    def get(self, key, params, build):
        """`build()` -> (value, jobs): allocates the persistent buffers and declares how they are filled."""
        e = self._store.get(key)
        if e is None:
            e = _Entry()
            e.params = [p for p in params if p is not None]
            e.value, e.jobs = build()
            e.vers = None
            e.epoch = -1
            if len(self._store) > 4096:
                self.clear()
            self._store[key] = e
        if not self._fresh(e):
            self.refresh()
        return e.value

    def refresh(self):
        stale = [e for e in self._store.values() if not self._fresh(e)]
        if not stale:
            return
        key = tuple(id(e) for e in stale)
        tab = self._tables.get(key)
        if tab is None:
            tab = self._build_tables([j for e in stale for j in e.jobs])
            if len(self._tables) > 16:
                self._retired = getattr(self, "_retired", []) + [({}, dict(self._tables))]   # a graph may have baked one in
                self._tables.clear()
            self._tables[key] = tab
        jobs_dev, blk_job, blk_chunk, nblocks, _keep = tab
        L.call("fod_multi_permute3", ops.ptr(jobs_dev), ops.ptr(blk_job), ops.ptr(blk_chunk), nblocks, ops.stream())
        for e in stale:
            e.vers = tuple(p._version for p in e.params)
            e.epoch = self.epoch if any(p.requires_grad for p in e.params) else -2

    def _build_tables(self, jobs):
        import ctypes as C
        import numpy as np
        if self._chunk is None:
            self._chunk = int(L.LIB.fod_multi_permute_chunk())
        arr = (L.PermuteJob * len(jobs))()
        bj, bc = [], []
        for i, j in enumerate(jobs):
            d0, d1, d2 = j.dims
            arr[i] = L.PermuteJob(j.src.data_ptr(), j.dst.data_ptr(), 0 if j.scale is None else j.scale.data_ptr(),
                                  ops._DT[j.src.dtype], ops._DT[j.dst.dtype], d0, d1, d2, j.valid1, j.valid2, j.axis,
                                  j.sstr[0], j.sstr[1], j.sstr[2], j.dstr[0], j.dstr[1])
            nchunks = int(L.LIB.fod_multi_permute_tiles(d0, d1, d2, j.sstr[0], j.sstr[1], j.sstr[2]))
            assert nchunks >= 0, f"permute job {j.dims} too large"
            bj.extend([i] * nchunks)
            bc.extend(range(nchunks))
        dev = jobs[0].dst.device
        raw = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy())
        up = lambda t: (t.pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else t.to(dev))
        return (up(raw), up(torch.tensor(bj, dtype=torch.int32)), up(torch.tensor(bc, dtype=torch.int32)), len(bj),
                [(j.src, j.dst, j.scale) for j in jobs])       # keep the operands alive while the table exists

    def mark_stale(self):
        """Every prepared operand is out of date (parameters were changed behind autograd's back, e.g. by a replayed
        hipGraph whose AdamW kernel does not bump `_version`): the next request refreshes them all."""
        self.epoch += 1

    def clear(self):
        # the old buffers and job tables are retired, not freed: a captured graph of an earlier model replays kernels
        # that read them -- the refresh launch's job table holds raw POINTERS, and a recycled table is a wild access
        # (observed: a GPU memory fault when a graph of model 1 was replayed after model 2 had been built)
        self._retired = getattr(self, "_retired", []) + [(self._store, dict(self._tables))]
        self._store = {}                 # a NEW dict: per-parameter memos (prep_linear) compare its identity
        self._tables.clear()


PREP = _Prepared()


class _ZeroArena:
    """Zero-initialised f32 scratch for gradient accumulation targets (the `_acc` kernels add into
    their output).  One big buffer is cleared with ONE fill per step instead of one fill launch per
    gradient tensor (~1300 per step).  Slices stay valid until the next `recycle()`, which the
    optimizer's zero_grad() calls once the gradients have been consumed; until the first recycle()
    the arena is off and `zeros()` falls back to torch.zeros."""

    def __init__(self):
        self.buf = None
        self.off = 0
        self.high = 0
        self.active = False
        self.extra = []
        self.cap = 0
        self.retired = []

    def recycle(self, device):
        device = torch.device(device)
        need = max(self.high, 1 << 20)
        if self.buf is None or self.buf.device != device or self.buf.numel() < need:
            if self.buf is not None:
                self.retired.append(self.buf)      # a captured step may have baked slices of it in: never freed
            self.buf = torch.zeros(int(need * 1.25), dtype=torch.float32, device=device)
        elif self.off:
            self.buf[:self.off].zero_()
        self.off = 0
        self.high = 0
        self.extra = []
        self.active = True
        self.cap = self.buf.numel()

    def zeros(self, shape, device):
        # ~800 calls per step: one aten call (as_strided) instead of slice + view, no torch.device() construction
        nd = len(shape)
        if nd == 1:
            n = shape[0]
            stride = (1,)
        elif nd == 2:
            n = shape[0] * shape[1]
            stride = (shape[1], 1)
        else:
            n = 1
            stride = [1] * nd
            for i in range(nd - 1, -1, -1):
                stride[i] = n
                n *= shape[i]
        n_al = (n + 63) // 64 * 64           # keep every slice 256-byte aligned
        self.high += n_al
        buf = self.buf
        if not self.active or self.off + n_al > self.cap or (buf.device != device and buf.device != torch.device(device)):
            return torch.zeros(shape, dtype=torch.float32, device=device)
        out = buf.as_strided(shape, stride, self.off)
        self.off += n_al
        return out


ARENA = _ZeroArena()

# Data-parallel runs: the object that averages finished arena regions across ranks (parallel.GradientReducer),
# installed by FodDataParallel.forward for the coming backward pass; None otherwise.
GRAD_SYNC = None


def set_grad_sync(reducer):
    global GRAD_SYNC
    GRAD_SYNC = reducer


class BackboneCut:
    """Cuts the autograd graph at the backbone's output so that a backward pass can run in TWO pieces: `loss.backward()`
    stops at the returned leaf (every transformer gradient + d loss / d features), `finish()` then runs the backbone's
    backward from it.  A data-parallel captured step (future_od/graph.py) puts the two pieces into separate graphs and
    averages the transformer gradients across ranks WHILE the backbone piece runs."""

    def __init__(self):
        self.pairs = []

    def cut(self, feat):
        leaf = feat.detach().requires_grad_(True)
        self.pairs.append((feat, leaf))
        return leaf

    def finish(self):
        pairs, self.pairs = self.pairs, []
        if pairs:
            torch.autograd.backward([f for f, _ in pairs], [l.grad for _, l in pairs])


BACKBONE_CUT = None


def zeros_f32(shape, device):
    return ARENA.zeros(tuple(shape), device)


class _WgradQueue:
    """The SHORT weight gradients of a backward pass (dW = G^T X over at most 512 rows: every Linear on the decoder's
    query side, ~130 per step) launched together instead of one by one.  None of them is on the critical path of the
    backward pass -- their results are first read by the gradient norm -- but each is a launch (a ~5 us graph node for
    ~1 us of work).  A site hands its operands to `tn` / `grouped`; the autograd node returns the still all-zero
    destination as the gradient, and ONE fod_gemm_tn_multi launch fills every destination when the pass ends (engine
    callback), before the data-parallel reducer averages a region (parallel.GradientReducer.flush), or before a
    parameter that already has a pending gradient in this pass is used again (autograd sums the gradients of a shared
    parameter when the second one arrives: the first must be real by then).  A plain Linear used several times in a pass
    (the decoder's query_scale, once per layer) does better: its later uses join the FIRST use's job as further
    (g, x) segments (`chain`) -- the block that owns a tile sums them in queue order and stores once, the autograd
    node returns no gradient of its own, so there is neither a flush nor a gradient-sum kernel, and the parameter's
    gradient stays inside the arena (one flat all-reduce in data-parallel runs).

    Not deferred (the site launches at once, as without the queue): eagerly launched steps (see `eager` below),
    parameters that already hold a .grad (autograd adds
    the returned tensor to it on arrival), parameters with tensor hooks, operands outside the short kernel's domain,
    FOD_WGRAD_QUEUE=0, torch's own DistributedDataParallel reducer (it copies gradients into buckets on arrival;
    parallel.FodDataParallel switches the queue off for it).

    Inside a stream capture the job table is written into a pinned host buffer set aside BEFORE the capture (allocating
    pinned memory inside one hangs) and copied by a captured memcpy node; host and device side of such a table are
    never reused.  Without a spare the jobs are launched one by one."""

    SPARE_BYTES = 1 << 18

    def __init__(self):
        import os
        self.enabled = os.environ.get("FOD_WGRAD_QUEUE", "1") != "0" and os.environ.get("FOD_TN_SMALL", "1") != "0"
        if not hasattr(torch._C, "_current_graph_task_id"):      # (private API: how the end of a backward pass is found)
            self.enabled = False
        # the LONG weight gradients (nn.Linear layers applied to more than 512 rows: the encoder, the memory side of
        # the decoder) wait too and share one fod_gemm_tn_multi_long launch; rows per M-split of that launch
        # Only while a stream capture records the step (future_od/graph.py -- the product's launch mode): an eagerly
        # launched step is bound by the launching thread, not by the GPU, and the queue's bookkeeping (~30 us per site)
        # made it 5 ms slower (28.7 -> 34.0 ms).  `eager` = True (FOD_WGRAD_QUEUE_EAGER=1; tests, bench.py's profiling
        # leg) queues there too.
        self.eager = os.environ.get("FOD_WGRAD_QUEUE_EAGER", "0") == "1"
        self.long_enabled = os.environ.get("FOD_WGRAD_QUEUE_LONG", "1") != "0"
        self.long_rows = int(os.environ.get("FOD_WGRAD_LONG_ROWS", "2048"))
        self.long_jobs = []
        self._plans = {}
        self._maps = {}
        self.jobs = []
        self.members = {}            # index into jobs -> further (G, X, M) contributions to that job's outputs
        self.keep = []
        self.serial = 0              # flushes so far: a job index is only meaningful within one
        self.task = -1               # autograd graph task whose end-of-pass callback is installed
        self.epoch = 1
        self._tables = {}
        self._spares = []
        self._baked = []
        self.hold = False            # tests: collect jobs outside a backward pass until flush() is called
        self.launches = 0            # multi launches / jobs they carried (tests, bench diagnostics)
        self.carried = 0

    # -- sites
    def site(self, params):
        """A backward node about to produce the gradients of `params`: True if its short weight gradients may wait."""
        if not self.enabled or not (self.eager or torch.cuda.is_current_stream_capturing()):
            return False
        ok = True
        for p in params:
            if p is None:
                continue
            if not p.is_leaf:          # its gradient is read by further backward nodes as soon as it is returned
                ok = False
            elif p.__dict__.get("_fod_wq") == self.epoch:
                self.flush()           # used again in this pass: the gradient handed out earlier must be real now
                ok = False
            elif (p.grad is not None or p._backward_hooks
                  or getattr(p, "_post_accumulate_grad_hooks", None)):
                ok = False
        if ok:
            for p in params:
                if p is not None:
                    p._fod_wq = self.epoch
        return ok

    @staticmethod
    def _fits(g, x, dw, db, M, N1, K2, ldg, ldx):
        if g.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or not g.is_cuda or M > 512 or M < 1:
            return False
        if N1 % 8 or K2 % 8 or ldg % 8 or ldx % 8 or (g.data_ptr() | x.data_ptr() | dw.data_ptr()) % 16:
            return False
        if ((N1 + 63) // 64) * ((K2 + 63) // 64) > 256 or not dw.is_contiguous():
            return False
        return db is None or db.is_contiguous()

    @staticmethod
    def _fits_long(g, x, dw, db, M, N1, K2):
        if g.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or not g.is_cuda:
            return False
        if not (g.is_contiguous() and x.is_contiguous() and dw.is_contiguous() and (db is None or db.is_contiguous())):
            return False
        if N1 % 8 or K2 % 8 or (g.data_ptr() | x.data_ptr() | dw.data_ptr()) % 16:
            return False
        return 2 * M * max(N1, K2) < 0xFFFFFF00 and 4 * N1 * K2 < 0x7FFFFF00

    def tn(self, ok, g, x, dw, db, owner=None):
        """dw [N1, K2] = g [M, N1]^T x [M, K2], db [N1] = column sums of g (dw, db all-zero f32).  `owner`: the weight
        parameter, if further uses of it in this pass may add to this job (`chain`)."""
        N1, K2 = g.shape[-1], x.shape[-1]
        M = g.numel() // N1
        if ok and M > 512 and self.long_enabled and self._fits_long(g, x, dw, db, M, N1, K2):
            return self._push((g.data_ptr(), x.data_ptr(), dw.data_ptr(), 0 if db is None else db.data_ptr(), N1, K2, K2,
                               M, N1, K2, 0, 0, 0), g, x, dw, db, long=True)
        if not (ok and g.is_contiguous() and x.is_contiguous() and self._fits(g, x, dw, db, M, N1, K2, N1, K2)):
            return ops.gemm_tn_acc(g, x, dw, colsum=db, zeroed=True)
        self._push((g.data_ptr(), x.data_ptr(), dw.data_ptr(), 0 if db is None else db.data_ptr(), N1, K2, K2,
                    M, N1, K2, 0, 0, 0), g, x, dw, db)
        if owner is not None and self.jobs:
            owner._fod_wq_job = (self.epoch, self.serial, len(self.jobs) - 1)

    def chain(self, owner, want_db, g, x):
        """A FURTHER use of `owner` (a weight whose gradient job of this pass is still waiting): its g^T x -- and g's
        column sums -- are added inside that job (summed in queue order by the block that owns the tile, one store).
        True: done, the caller returns no gradient for the parameter (autograd would add a second tensor with a kernel
        of its own, and could not, the first one being unfinished).  False: not possible, proceed as usual."""
        rec = owner.__dict__.get("_fod_wq_job") if self.enabled else None
        if rec is None or rec[0] != self.epoch or rec[1] != self.serial or rec[2] >= len(self.jobs):
            return False
        head = self.jobs[rec[2]]
        N1, K2 = g.shape[-1], x.shape[-1]
        M = g.numel() // N1
        if (head[8], head[9]) != (N1, K2) or head[11] != 0 or (head[3] != 0) != bool(want_db):
            return False
        if not (g.is_contiguous() and x.is_contiguous() and g.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
                and 1 <= M <= 512 and (g.data_ptr() | x.data_ptr()) % 16 == 0):
            return False
        self.members.setdefault(rec[2], []).append((g.data_ptr(), x.data_ptr(), 0, 0, N1, K2, K2, M, N1, K2, 1, 0, 0))
        self.keep.append((g, x, None, None))
        return True

    def grouped(self, ok, g, x, dw, db):
        """g [P, rows, D] (P output gradients, each block contiguous), x [rows, K] -> dw [P*D, K], db [P*D]."""
        P, rows, D = g.shape
        K = x.shape[-1]
        if not (ok and D % 64 == 0 and g.is_contiguous() and x.is_contiguous()
                and self._fits(g, x, dw, db, rows, P * D, K, D, K)):
            return ops.group_linear_wgrad(g, x, dw, db, zeroed=True)
        self._push((g.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), D, K, K, rows, P * D, K, 0, D, rows * D),
                   g, x, dw, db)

    def _push(self, job, g, x, dw, db, long=False):
        task = torch._C._current_graph_task_id()
        if task != self.task and (self.jobs or self.long_jobs):    # left behind by a backward pass that raised
            self.jobs, self.long_jobs, self.keep, self.members = [], [], [], {}
            self.serial += 1
        (self.long_jobs if long else self.jobs).append(job)
        # detach(): a second handle on the same memory -- the gradient tensor itself must stay singly referenced, or
        # autograd copies it instead of adopting it as .grad
        self.keep.append((g, x, dw.detach(), None if db is None else db.detach()))
        if task < 0:                     # not inside a backward pass (a backward function called directly)
            self.task = -1
            if not self.hold:
                self.flush()
        elif task != self.task:
            self.task = task
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_pass)

    def _end_of_pass(self):
        self.task = -1
        self.epoch += 1
        self.flush()

    # -- launch
    # A table = the job array (pointers: rebuilt per flush, one numpy call) + the block maps, which depend on the jobs'
    # SHAPES only and are cached per shape list (building them is milliseconds of Python; eagerly launched steps, whose
    # activation addresses differ from step to step, would pay that every time).
    _JOB_DTYPE = None

    @classmethod
    def _job_array(cls, rows):
        import ctypes as C
        import numpy as np
        if cls._JOB_DTYPE is None:
            cls._JOB_DTYPE = np.dtype([(n, np.uint64 if t is C.c_void_p else (np.int64 if t is C.c_long else np.int32))
                                       for n, t in L.TnJob._fields_])
            assert cls._JOB_DTYPE.itemsize == C.sizeof(L.TnJob)
        head = np.array(rows, dtype=cls._JOB_DTYPE).view(np.uint8)
        pad = (-head.size) % 16
        return head if pad == 0 else np.concatenate([head, np.zeros(pad, np.uint8)])

    def _pack(self, jobs, members):
        import numpy as np
        key = tuple((j[7], j[8], j[9], tuple(m[7] for m in members.get(i, ()))) for i, j in enumerate(jobs))
        cached = self._maps.get(key)
        if cached is None:
            rows = lambda i: jobs[i][7] + sum(m[7] for m in members.get(i, ()))
            order = sorted(range(len(jobs)), key=lambda i: -rows(i))             # long reductions first
            bj, bt, slot = [], [], 0
            for i in order:
                j = jobs[i]
                tiles = ((j[8] + 63) // 64) * ((j[9] + 63) // 64)
                bj.extend([slot] * tiles)
                bt.extend(range(tiles))
                slot += 1 + len(members.get(i, ()))
            if len(self._maps) > 64:
                self._maps.clear()
            cached = self._maps[key] = (order, np.asarray(bj + bt, dtype=np.int32).view(np.uint8), len(bj))
        order, maps, nblocks = cached
        table = []
        for i in order:
            more = members.get(i, ())
            for e, n in [(jobs[i], len(more))] + [(m, 0) for m in more]:
                table.append(e[:12] + (n, e[12], 0, 0))
        head = self._job_array(table)
        return np.concatenate([head, maps]), head.size, nblocks

    def _pack_long(self, jobs):
        """Table + block maps of a fod_gemm_tn_multi_long launch.  The blocks of one M-split of a job re-read the same
        rows of G and X: they go to ONE XCD (block ids congruent mod 8 share an L2), splits dealt round-robin; idle
        blocks (job -1) pad the shorter XCD queues."""
        import ctypes as C
        import numpy as np
        key = ("long",) + tuple((j[7], j[8], j[9]) for j in jobs)
        cached = self._maps.get(key)
        if cached is None:
            order = sorted(range(len(jobs)), key=lambda i: -jobs[i][7] * jobs[i][8] * jobs[i][9])
            plans = []
            queues = [[] for _ in range(8)]
            turn = 0
            for slot, i in enumerate(order):
                j = jobs[i]
                plan = self._plans.get(j[7])
                if plan is None:
                    mps, ns = C.c_int(), C.c_int()
                    L.call("fod_tn_plan_long", j[7], self.long_rows, C.addressof(mps), C.addressof(ns))
                    plan = self._plans[j[7]] = (mps.value, ns.value)
                plans.append(plan)
                ntile = ((j[8] + 127) // 128) * ((j[9] + 127) // 128)
                for sp in range(plan[1]):
                    queues[turn % 8].append((slot, sp * ntile, ntile))
                    turn += 1
            depth = max(sum(r[2] for r in q) for q in queues)
            bj = np.full((depth, 8), -1, dtype=np.int32)
            bl = np.zeros((depth, 8), dtype=np.int32)
            for xcd, q in enumerate(queues):
                at = 0
                for slot, first, ntile in q:
                    bj[at:at + ntile, xcd] = slot
                    bl[at:at + ntile, xcd] = np.arange(first, first + ntile, dtype=np.int32)
                    at += ntile
            maps = np.concatenate([bj.reshape(-1), bl.reshape(-1)]).view(np.uint8)
            if len(self._maps) > 64:
                self._maps.clear()
            cached = self._maps[key] = (order, plans, maps, depth * 8)
        order, plans, maps, nblocks = cached
        head = self._job_array([jobs[i][:12] + (0, jobs[i][12]) + plans[k] for k, i in enumerate(order)])
        return np.concatenate([head, maps]), head.size, nblocks

    def _top_up(self, device):
        while len(self._spares) < 8:
            self._spares.append((torch.empty(self.SPARE_BYTES, dtype=torch.uint8).pin_memory(),
                                 torch.empty(self.SPARE_BYTES, dtype=torch.uint8, device=device)))

    def prepare(self, device):
        """Set aside the capture-time tables (call outside a capture; future_od/graph.py does before it captures)."""
        if self.enabled and torch.device(device).type == "cuda":
            self._top_up(device)

    def flush(self):
        jobs, long_jobs = self.jobs, self.long_jobs
        if not jobs and not long_jobs:
            return
        keep, members = self.keep, self.members
        self.jobs, self.long_jobs, self.keep, self.members = [], [], [], {}
        self.serial += 1
        dev = keep[0][0].device
        if jobs:
            self._launch("fod_gemm_tn_multi", jobs, members, dev, keep)
        if long_jobs:
            self._launch("fod_gemm_tn_multi_long", long_jobs, {}, dev, keep)

    def _launch(self, entry, jobs, members, dev, keep):
        long = entry == "fod_gemm_tn_multi_long"
        sig = (entry, tuple(jobs), tuple((i, tuple(m)) for i, m in sorted(members.items())))
        tab = self._tables.get(sig)
        if tab is None:
            raw, off, nblocks = self._pack_long(jobs) if long else self._pack(jobs, members)
            if torch.cuda.is_current_stream_capturing():
                if not self._spares or raw.size > self.SPARE_BYTES or self._spares[-1][1].device != dev:
                    return self._one_by_one(jobs, members, long)
                pin, table = self._spares.pop()
                self._baked.append((pin, table))         # the graph reads both at every replay: never reused
                pin[:raw.size].copy_(torch.from_numpy(raw))
                table[:raw.size].copy_(pin[:raw.size], non_blocking=True)
                tab = (table, off, nblocks)
            else:
                table = torch.from_numpy(raw).pin_memory().to(dev, non_blocking=True)
                tab = (table, off, nblocks)
                if len(self._tables) >= 32:
                    self._tables.clear()
                self._tables[sig] = tab
                self._top_up(dev)
        table, off, nblocks = tab
        base = table.data_ptr()
        L.call(entry, base, base + off, base + off + 4 * nblocks, nblocks, ops.stream(),
               work=sum(2.0 * j[7] * j[8] * j[9] for j in jobs)
               + sum(2.0 * m[7] * m[8] * m[9] for ms in members.values() for m in ms), tag="fod_gemm_tn_acc")
        self.launches += 1
        self.carried += len(jobs) + sum(len(m) for m in members.values())

    def _one_by_one(self, jobs, members, long=False):
        if long:
            for G, X, dW, cs, ldg, ldx, ldw, M, N1, K2, acc, _sc, _ss in jobs:
                ws, ws_bytes = ops.tn_workspace(torch.device("cuda", torch.cuda.current_device())) if M >= 8192 else (None, 0)
                L.call("fod_gemm_tn_acc", L.BF16, G, ldg, X, ldx, dW, ldw, M, N1, K2, 0, cs, 1, ws, ws_bytes, ops.stream(),
                       work=2.0 * M * N1 * K2)
            return
        for i, (G, X, dW, cs, ldg, ldx, ldw, M, N1, K2, acc, seg_cols, seg_stride) in enumerate(jobs):
            L.call("fod_gemm_tn_grouped", L.BF16, G, ldg, seg_cols, seg_stride, X, ldx, dW, ldw, M, N1, K2, cs,
                   acc, ops.stream(), work=2.0 * M * N1 * K2, tag="fod_gemm_tn_acc")
            for m in members.get(i, ()):             # stream-ordered after the head: plain read-modify-write is safe
                L.call("fod_gemm_tn_grouped", L.BF16, m[0], m[4], 0, 0, m[1], m[5], dW, ldw, m[7], N1, K2, cs, 1,
                       ops.stream(), work=2.0 * m[7] * N1 * K2, tag="fod_gemm_tn_acc")


WGRADS = _WgradQueue()


def _pad_to(n, v):
    return (n + v - 1) // v * v


def _wkey(p, kind, dtype, extra=()):
    return (p.data_ptr(), tuple(p.shape), tuple(p.stride()), kind, dtype) + tuple(extra)


def prep_linear(weight, dtype, transposed):
    """weight [N,K] f32 -> [Np,K] (rows zero-padded to the vector width) or its transpose [K,Np]."""
    # fast path (~270 calls per step): the registry entry is remembered on the parameter object itself
    memo = weight.__dict__.get("_fod_prep")
    if memo is not None:
        e = memo.get((transposed, dtype))
        if (e is not None and e[0] is PREP._store and e[2] == weight.data_ptr()
                and (e[1].epoch == PREP.epoch or e[1].epoch == -2) and e[1].vers == (weight._version,)):
            return e[1].value
    N, K = weight.shape
    v = _VEC[dtype]
    assert K % v == 0, f"Linear in_features {K} must be a multiple of {v} (pad the input)"
    Np = _pad_to(N, v)
    w = weight.detach()
    sn, sk = w.stride()

    def build():
        if not transposed:   # dst[1][n][k], rows n >= N zero
            out = torch.empty((Np, K), dtype=dtype, device=w.device)
            return out, [_Job(w, out, (1, Np, K), (0, sn, sk), valid1=N)]
        out = torch.empty((K, Np), dtype=dtype, device=w.device)      # dst[1][k][n] = w[n][k], columns n >= N zero
        return out, [_Job(w, out, (1, K, Np), (0, sk, sn), valid2=N)]

    key = _wkey(weight, "lin_t" if transposed else "lin", dtype)
    value = PREP.get(key, [weight], build)
    if weight.__dict__.get("_fod_prep") is None:
        try:
            weight._fod_prep = {}
        except Exception:            # not an attribute-capable tensor (should not happen for Parameters)
            return value
    weight._fod_prep[(transposed, dtype)] = (PREP._store, PREP._store[key], weight.data_ptr())
    return value


def prep_conv(weight, dtype, scale, transposed, cin_pad=None):
    """OIHW f32 (any strides) -> [Cout][kh*kw][Cin_p] * scale[co]  or  [Cin][kh*kw][Cout] * scale[co]."""
    co, ci, kh, kw = weight.shape
    s_co, s_ci, s_kh, s_kw = weight.stride()
    assert s_kh == kw * s_kw, "conv weight must have a contiguous tap plane"
    cp = ci if cin_pad is None else cin_pad
    w = weight.detach()

    def build():
        if not transposed:
            out = torch.empty((co, kh * kw, cp), dtype=dtype, device=w.device)
            return out, [_Job(w, out, (co, kh * kw, cp), (s_co, s_kw, s_ci), valid2=ci, scale=scale,
                              axis=0 if scale is not None else -1)]
        out = torch.empty((ci, kh * kw, co), dtype=dtype, device=w.device)
        return out, [_Job(w, out, (ci, kh * kw, co), (s_ci, s_kw, s_co), scale=scale,
                          axis=2 if scale is not None else -1)]

    tag = ("conv_t" if transposed else "conv") + ("_s" if scale is not None else "")
    return PREP.get(_wkey(weight, tag, dtype, (cp, 0 if scale is None else scale.data_ptr())), [weight], build)


def prep_stem(weight, dtype, scale7):
    """Stem weight OIHW [Cout,3,7,7] f32 -> [Cout][7 tap rows][8 pixels][4 channels] * scale (zeros for pixel 7 /
    channel 3): the k order of fod_conv_stem_fwd.  `scale7` = the frozen-BN scale repeated per tap row [Cout*7]."""
    co, ci, kh, kw = weight.shape
    s_co, s_ci, s_kh, s_kw = weight.stride()
    assert (ci, kh, kw) == (3, 7, 7) and s_co == kh * s_kh, "stem weight: expected [Cout,3,7,7] with dense tap rows"
    w = weight.detach()

    def build():
        out = torch.empty((co, 7, 8, 4), dtype=dtype, device=w.device)
        return out, [_Job(w, out, (co * 7, 8, 4), (s_kh, s_kw, s_ci), valid1=7, valid2=3, scale=scale7,
                          axis=0 if scale7 is not None else -1)]

    return PREP.get(_wkey(weight, "stem", dtype, (0 if scale7 is None else scale7.data_ptr(),)), [weight], build)


def cast(t, dtype, pad_cols=None):
    """[rows, cols] tensor -> dtype, optionally zero-padding the last dim."""
    cols = t.shape[-1]
    rows = t.numel() // cols
    t = t.contiguous()
    pc = cols if pad_cols is None else pad_cols
    return ops.permute3_cast(t, dtype, (1, rows, pc), (0, cols, 1), valid2=cols).view(*t.shape[:-1], pc)


# ------------------------------------------------------------------------------------------------
class LinearFn(Function):
    """y = act(x W^T + b) (+ residual)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, out_f32):
        dtype = x.dtype
        w = prep_linear(weight, dtype, False)
        N = weight.shape[0]
        y = ops.gemm_nt(x, w[:N] if w.shape[0] != N else w, shift=bias, relu=relu, out_f32=out_f32)
        y = y.view(*x.shape[:-1], N)
        ctx.relu, ctx.out_f32 = relu, out_f32
        ctx.weight, ctx.bias = weight, bias
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        weight = ctx.weight
        dtype = x.dtype
        N, K = weight.shape
        v = _VEC[dtype]
        Np = _pad_to(N, v)
        dy = dy.contiguous()
        if dy.dtype != dtype or Np != N:
            g = cast(dy.view(-1, N), dtype, pad_cols=Np)
        else:
            g = dy.view(-1, N)
        if ctx.relu:
            assert Np == N
            g = ops.eltwise(L.EW_RELU_MASK, g, y.view(-1, N))
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(g, prep_linear(weight, dtype, True)).view(x.shape)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] and WGRADS.chain(weight, want_db, g, x.view(-1, K)):
            return dx, None, None, None, None        # a further use of a shared layer: summed inside the first use's job
        dbp = zeros_f32((Np,), x.device) if want_db else None
        if ctx.needs_input_grad[1]:
            dwp = zeros_f32((Np, K), x.device)
            # bias gradient from the same pass
            WGRADS.tn(WGRADS.site((weight, ctx.bias)), g, x.view(-1, K), dwp, dbp, owner=weight)
            dw = dwp[:N] if Np != N else dwp
        elif want_db:
            ops.colsum_acc(g, dbp)
        if want_db:
            db = dbp[:N] if Np != N else dbp
        return dx, dw, db, None, None


def linear(x, weight, bias=None, relu=False, out_f32=False):
    return LinearFn.apply(x.contiguous(), weight, bias, relu, out_f32)


class LinearKeepFn(Function):
    """(x, act(x W^T + b)): the projection of a tensor that is ALSO the residual of the sub-layer it opens.  As two
    consumers of x, autograd would sum their gradients with a kernel of its own (a 5 us graph node per sub-layer, ~45 per
    step); here the residual path's gradient enters the projection's input-gradient GEMM as its epilogue operand."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, grad_masked=False, precomputed=None):
        """`grad_masked` (with relu): the consumer of y applies the (y > 0) mask to the gradient it sends back
        (LinearAddNormFn(a_relu=True)); this node then skips its own masking pass.  `precomputed`: act(x W^T + b) as the
        launch that produced x already computed it (LinearAddNormFn(then=...)): no launch here, the same backward."""
        dtype = x.dtype
        N = weight.shape[0]
        if precomputed is not None:
            y = precomputed.view(*x.shape[:-1], N)
        else:
            w = prep_linear(weight, dtype, False)
            y = ops.gemm_nt(x, w[:N] if w.shape[0] != N else w, shift=bias, relu=relu).view(*x.shape[:-1], N)
        ctx.relu, ctx.weight, ctx.bias, ctx.has_bias = relu, weight, bias, bias is not None
        ctx.mask_here = relu and not grad_masked
        ctx.save_for_backward(x, y if ctx.mask_here else None)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dkeep, dy):
        x, y = ctx.saved_tensors
        weight = ctx.weight
        dtype = x.dtype
        N, K = weight.shape
        assert N % _VEC[dtype] == 0
        g = dy.contiguous().view(-1, N)
        if g.dtype != dtype:
            g = cast(g, dtype)
        if ctx.mask_here:
            g = ops.eltwise(L.EW_RELU_MASK, g, y.view(-1, N))
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            res = None if dkeep is None else dkeep.contiguous().view(-1, K)
            if res is not None and res.dtype != dtype:
                res = cast(res, dtype)
            dx = ops.gemm_nt(g, prep_linear(weight, dtype, True), residual=res).view(x.shape)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        dbp = zeros_f32((N,), x.device) if want_db else None
        if ctx.needs_input_grad[1]:
            dw = zeros_f32((N, K), x.device)
            WGRADS.tn(WGRADS.site((weight, ctx.bias)), g, x.view(-1, K), dw, dbp)
        elif want_db:
            ops.colsum_acc(g, dbp)
        return dx, dw, dbp, None, None, None


_LINEAR_KEEP = __import__("os").environ.get("FOD_LINEAR_KEEP", "1") != "0"       # "0": two autograd consumers (experiments)


def linear_keep(x, weight, bias=None, relu=False, grad_masked=False, precomputed=None):
    """-> (x for the residual path, act(x W^T + b)); see LinearKeepFn.  Falls back to two consumers when the output
    width does not fit the vector epilogue (that path masks its own gradient: a second mask upstream is harmless)."""
    if weight.shape[0] % _VEC[x.dtype] != 0 or not _LINEAR_KEEP:
        assert precomputed is None
        return x, linear(x, weight, bias, relu=relu)
    return LinearKeepFn.apply(x.contiguous(), weight, bias, relu, grad_masked, precomputed)


class AddFn(Function):
    """a + b[row(m)], b broadcast over row groups (div) or periodically (mod)."""

    @staticmethod
    def forward(ctx, a, b, b_row_div, b_row_mod):
        ctx.cfg = (b_row_div, b_row_mod, b.shape, a.shape)
        return ops.eltwise(L.EW_ADD, a, b, b_row_div=b_row_div, b_row_mod=b_row_mod)

    @staticmethod
    def backward(ctx, g):
        div, mod, bshape, ashape = ctx.cfg
        g = g.contiguous()
        db = None
        if ctx.needs_input_grad[1]:
            cols = ashape[-1]
            rows = g.numel() // cols
            if not div and not mod:
                db = g.view(bshape)
            elif mod and not div:
                db = _sum_periodic(g.view(rows, cols), mod).view(bshape)
            else:
                assert not mod
                out = zeros_f32((rows // div, cols), g.device)
                ops.colsum_acc(g.view(rows, cols), out, group_rows=div)
                db = cast(out, g.dtype).view(bshape)
        return (g.view(ashape) if ctx.needs_input_grad[0] else None), db, None, None


def add(a, b, b_row_div=0, b_row_mod=0):
    return AddFn.apply(a.contiguous(), b.contiguous(), b_row_div, b_row_mod)


class MulFn(Function):
    """a * b[m % b_row_mod] (b broadcast periodically over the rows of a when b_row_mod > 0)."""

    @staticmethod
    def forward(ctx, a, b, b_row_mod):
        ctx.save_for_backward(a, b)
        ctx.mod = b_row_mod
        return ops.eltwise(L.EW_MUL, a, b, b_row_mod=b_row_mod)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da = ops.eltwise(L.EW_MUL, g, b, b_row_mod=ctx.mod)
        db = ops.eltwise(L.EW_MUL, g, a)
        if ctx.mod:
            db = _sum_periodic(db, ctx.mod).view(b.shape)
        return da, db, None


def _queued_linear_wgrad(weight, bias, g, x, need_w, need_b):
    """(dw, db) of a Linear layer as LinearFn.backward produces them: through the weight-gradient queue, a further use of a
    shared layer inside the first use's job ((None, None) then)."""
    N, K = weight.shape
    want_db = bias is not None and need_b
    if need_w and WGRADS.chain(weight, want_db, g, x):
        return None, None
    db = zeros_f32((N,), x.device) if want_db else None
    dw = None
    if need_w:
        dw = zeros_f32((N, K), x.device)
        WGRADS.tn(WGRADS.site((weight, bias)), g, x, dw, db, owner=weight)
    elif want_db:
        ops.colsum_acc(g, db)
    return dw, db


class TableGradAcc:
    """One f32 accumulation buffer for the gradient of a table that several Mlp2MulFn nodes multiply by (the decoder
    layers' query_scale(x) * sine embedding): every node's backward launch adds into it with atomics; the node that runs
    LAST in the backward pass (the first in the forward pass: later layers depend on its output) hands the total on --
    one gradient instead of one per layer that autograd would sum with kernels of its own."""

    def __init__(self):
        self.buf = None

    def get(self, shape, device):
        if self.buf is None:
            self.buf = zeros_f32(shape, device)
        return self.buf

    def take(self):
        buf, self.buf = self.buf, None
        return buf


# "0": the decoder's query_scale MLP and the product with the sine embedding as separate launches (2 GEMMs + multiply)
FUSED_MLP2 = os.environ.get("FOD_FUSED_MLP2", "1") != "0"


class Mlp2MulFn(Function):
    """((relu(x W1^T + b1)) W2^T + b2) * table[m % rows(table)] as ONE launch forward and ONE backward
    (fod_mlp2_mul_fwd / _bwd, csrc/linear_norm.hip): reference transformer.py:384-386, once per decoder layer.  As separate
    nodes that was 2 GEMMs + a multiply forward and 2 GEMMs + a ReLU gate + 2 multiplies + a sum over the batch backward."""

    @staticmethod
    def forward(ctx, x, table, w1, b1, w2, b2, acc, hands_on):
        D = x.shape[-1]
        x2 = x.view(-1, D)
        out, h, q = ops.mlp2_mul_fwd(x2, prep_linear(w1, x.dtype, False), b1, prep_linear(w2, x.dtype, False), b2, table)
        ctx.save_for_backward(x2, h, q, table)
        ctx.params, ctx.acc, ctx.hands_on = (w1, b1, w2, b2), acc, hands_on
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        x2, h, q, table = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        D = x2.shape[-1]
        g = g.contiguous().view(-1, D)
        if g.dtype != x2.dtype:
            g = cast(g, x2.dtype)
        acc = ctx.acc if ctx.acc is not None else TableGradAcc()
        buf = acc.get(tuple(table.shape), x2.device)
        ds, dh, dx = ops.mlp2_mul_bwd(g, table, q, h, prep_linear(w2, x2.dtype, True), prep_linear(w1, x2.dtype, True), buf)
        dw2, db2 = _queued_linear_wgrad(w2, b2, ds, h, ctx.needs_input_grad[4], ctx.needs_input_grad[5])
        dw1, db1 = _queued_linear_wgrad(w1, b1, dh, x2, ctx.needs_input_grad[2], ctx.needs_input_grad[3])
        dtable = None
        if ctx.needs_input_grad[1] and (ctx.acc is None or ctx.hands_on):
            dtable = cast(acc.take(), table.dtype).view(table.shape)
        return (dx.view(-1, D).view(x2.shape) if ctx.needs_input_grad[0] else None), dtable, dw1, db1, dw2, db2, None, None


def mlp2_mul_fits(x, mlp, table):
    """Whether Mlp2MulFn applies: bf16, a two-layer 256 -> 256 -> 256 MLP with biases, a [rows, 256] table dividing x's rows."""
    if not FUSED_MLP2 or x.dtype != torch.bfloat16 or getattr(mlp, "num_layers", 0) != 2:
        return False
    l0, l1 = mlp.layers
    D = x.shape[-1]
    return (D == 256 and tuple(l0.weight.shape) == (D, D) and tuple(l1.weight.shape) == (D, D) and l0.bias is not None
            and l1.bias is not None and table.dtype == x.dtype and table.dim() == 2 and table.shape[1] == D
            and (x.numel() // D) % table.shape[0] == 0)


def mlp2_mul(x, mlp, table, acc=None, hands_on=True):
    """mlp(x) * table[m % rows(table)] (see Mlp2MulFn).  `acc` (TableGradAcc) shared by several calls on one table: only the
    call with hands_on=True -- the FIRST of them in the forward pass -- returns the table's (total) gradient."""
    l0, l1 = mlp.layers
    return Mlp2MulFn.apply(x.contiguous(), table.contiguous(), l0.weight, l0.bias, l1.weight, l1.bias, acc, hands_on)


def _sum_periodic(g, mod):
    """[rows, cols] -> [mod, cols]: sum of the rows congruent modulo `mod` (rows/mod is tiny)."""
    cols = g.shape[-1]
    parts = g.view(-1, mod, cols)
    acc = parts[0]
    for i in range(1, parts.shape[0]):
        acc = ops.eltwise(L.EW_ADD, acc.contiguous(), parts[i].contiguous())
    return acc.contiguous()


def mul(a, b, b_row_mod=0):
    return MulFn.apply(a.contiguous(), b.contiguous(), b_row_mod)


class ExpandRowsFn(Function):
    """[rows, cols] -> [reps*rows, cols] (periodic copy); backward sums the copies."""

    @staticmethod
    def forward(ctx, x, reps):
        rows, cols = x.shape
        ctx.rows = rows
        out = torch.empty((reps * rows, cols), dtype=x.dtype, device=x.device)
        return ops.eltwise(L.EW_COPY_B, out, x, b_row_mod=rows, out=out)

    @staticmethod
    def backward(ctx, g):
        return _sum_periodic(g.contiguous(), ctx.rows), None


def expand_rows(x, reps):
    return ExpandRowsFn.apply(x.contiguous(), reps)


# "0": every encoder layer sums the gradient of its per-frame IMU rows itself (fod_colsum_acc + a cast per layer)
RES_GRAD_QUEUE = os.environ.get("FOD_RES_GRAD_QUEUE", "1") != "0"


class ResGradQueue:
    """The gradients of P broadcast residual rows ([groups, D] each: the per-frame IMU rows that P encoder layers add to
    their tokens) as slots of ONE [P, groups, D] buffer, filled by ONE launch (fod_colsum_groups_multi) when their reader
    -- ImuBranchFn.backward -- is reached: a LayerNormFn whose residual is such a row hands its gradient's tokens over
    and returns the (still unfilled) slot."""

    def __init__(self, P):
        self.P, self.buf, self.jobs, self.geom = P, None, [], None

    def push(self, slot, g, groups, group_rows, D):
        if self.buf is None:
            self.buf = torch.empty((self.P, groups, D), dtype=g.dtype, device=g.device)
        geom = (groups, group_rows, D)
        assert self.geom in (None, geom) and tuple(self.buf.shape[1:]) == (groups, D)
        self.geom = geom
        self.jobs.append((g, self.buf[slot]))
        return self.buf[slot]

    def flush(self):
        if self.jobs:
            jobs, self.jobs = self.jobs, []
            for i in range(0, len(jobs), 16):
                ops.colsum_groups_multi(jobs[i:i + 16], *self.geom)


class LayerNormFn(Function):
    """LayerNorm(x + residual[row(m)]) over the last dim."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, res_row_div, res_queue=None, res_slot=0):
        y, s, mean, rstd = ops.layernorm_fwd(x, gamma, beta, residual=residual, res_row_div=res_row_div)
        ctx.save_for_backward(s, mean, rstd, gamma)
        ctx.res_row_div = res_row_div
        ctx.res_shape = None if residual is None else residual.shape
        ctx.res_queue, ctx.res_slot = res_queue, res_slot
        return y

    @staticmethod
    def backward(ctx, dy):
        s, mean, rstd, gamma = ctx.saved_tensors
        D = gamma.numel()
        dg = zeros_f32((D,), dy.device)
        db = zeros_f32((D,), dy.device)
        dx = ops.layernorm_bwd(dy.contiguous(), s, mean, rstd, gamma, dg, db)
        dres = None
        if ctx.res_shape is not None and ctx.needs_input_grad[1]:
            rows = dx.numel() // D
            if (ctx.res_row_div and ctx.res_queue is not None and dx.dtype == torch.bfloat16 and D <= 256 and D % 4 == 0
                    and rows % ctx.res_row_div == 0):
                # the sum over each frame's tokens waits for the other layers' (ResGradQueue): the slot is filled later
                dres = ctx.res_queue.push(ctx.res_slot, dx.view(rows, D), rows // ctx.res_row_div, ctx.res_row_div,
                                          D).view(ctx.res_shape)
            elif ctx.res_row_div:
                out = zeros_f32((rows // ctx.res_row_div, D), dy.device)
                ops.colsum_acc(dx.view(rows, D), out, group_rows=ctx.res_row_div)
                dres = cast(out, dx.dtype).view(ctx.res_shape)
            else:
                dres = dx.view(ctx.res_shape)
        return dx, dres, dg, db, None, None, None


def layer_norm(x, gamma, beta, residual=None, res_row_div=0, res_queue=None, res_slot=0):
    """res_queue / res_slot (with res_row_div): the residual's gradient is summed later, with the other slots of the queue."""
    return LayerNormFn.apply(x.contiguous(), None if residual is None else residual.contiguous(), gamma, beta,
                             res_row_div, res_queue, res_slot)


# "0": the output projection and the post-norm of the decoder's attention blocks as two launches (fod_gemm_nt +
# fod_layernorm_fwd) instead of fod_linear_add_norm_fwd
FUSED_LINEAR_NORM = os.environ.get("FOD_FUSED_LINEAR_NORM", "1") != "0"
# the next cross-attention block's query-content projection from the same launch (fod_linear_add_norm_fwd then_*); "0": its own GEMM
FUSED_LINEAR_THEN = os.environ.get("FOD_FUSED_LINEAR_THEN", "1") != "0"
# ... and its backward: the input gradient of that projection (dq . then_w) is formed inside the fused backward launch
# (fod_linear_add_norm_bwd pre_*) instead of by a GEMM of LinearKeepFn; "0": LinearKeepFn(precomputed=...) as before
FUSED_LINEAR_PRE = os.environ.get("FOD_FUSED_LINEAR_PRE", "1") != "0"
# rows up to which the fused launches are used.  The kernels walk 16-row tiles with the weights stationary, so they take any
# row count, but at the encoder's 14 500 rows they do not pay (one wave per SIMD, every tile's loads queue behind the
# previous tile's stores): forward 19.5 us against 21.3 for GEMM + norm, backward 26.3 against 20.9, whole step +0.1 ms
# (profiles/r03l_fused_linear_norm.txt)
FUSED_LINEAR_NORM_ROWS = int(os.environ.get("FOD_FUSED_LINEAR_NORM_ROWS", "1024"))


class LinearAddNormFn(Function):
    """y = LayerNorm(x + (a W^T + b)): a sub-layer's output projection, the residual add and the post-norm as ONE
    autograd node -- the same two kernels forward and four backward as LinearFn + LayerNormFn, but one Function
    application instead of two per site (~55 sites per step; the launching thread is the bottleneck on slow hosts).
    Used where no dropout sits between the projection and the add (eval mode or p = 0)."""

    @staticmethod
    def forward(ctx, a, x, weight, bias, gamma, beta, a_relu=False, then_weight=None, then_bias=None):
        """`a_relu`: `a` is the output of a ReLU whose producer leaves the masking of its incoming gradient to THIS
        node (LinearKeepFn(grad_masked=True)): the (a > 0) mask is applied in the epilogue of the GEMM that forms da --
        one pass over a [rows, 2048] gradient less per feed-forward block."""
        N = weight.shape[0]
        w = prep_linear(weight, a.dtype, False)
        assert w.shape[0] == N, "linear_add_norm: out_features must be a multiple of the vector width"
        rows = a.numel() // a.shape[-1]
        if (FUSED_LINEAR_NORM and a.dtype == torch.bfloat16 and N == 256 and weight.shape[1] == 256 and w.shape[1] == 256
                and rows <= FUSED_LINEAR_NORM_ROWS):
            # the decoder's query side: projection + residual add + norm as ONE launch (csrc/linear_norm.hip)
            if then_weight is not None:
                # ... and the next block's 256 -> 256 projection of y from the same launch: a CONSTANT second output
                # that LinearKeepFn(precomputed=...) adopts as its result (its backward is the usual one)
                y, s, mean, rstd, nxt = ops.linear_add_norm_fwd(a, w, bias, x, gamma, beta,
                                                                then_w=prep_linear(then_weight, a.dtype, False),
                                                                then_bias=then_bias)
            else:
                y, s, mean, rstd = ops.linear_add_norm_fwd(a, w, bias, x, gamma, beta)
        else:
            assert then_weight is None, "linear_add_norm(then=...) outside the fused launch's domain (the caller checks)"
            o = ops.gemm_nt(a, w, shift=bias).view(x.shape)
            y, s, mean, rstd = ops.layernorm_fwd(x, gamma, beta, residual=o)
        ctx.then = None
        ctx.weight, ctx.bias, ctx.has_bias, ctx.a_relu = weight, bias, bias is not None, bool(a_relu)
        if then_weight is not None and FUSED_LINEAR_PRE:
            # the second result is differentiable: its gradient comes back to THIS node (backward's pre form)
            ctx.then = (then_weight, then_bias)
            ctx.save_for_backward(a, s, mean, rstd, gamma, y)
            return y, nxt
        ctx.save_for_backward(a, s, mean, rstd, gamma)
        if then_weight is not None:
            ctx.mark_non_differentiable(nxt)
            return y, nxt
        return y

    @staticmethod
    def backward(ctx, dy, dnxt=None):
        a, s, mean, rstd, gamma = ctx.saved_tensors[:5]
        weight = ctx.weight
        N, K = weight.shape
        dev = s.device
        dg, dbeta = zeros_f32((N,), dev), zeros_f32((N,), dev)
        da = None
        rows = s.numel() // N
        dw2 = db2 = None
        pre_g = pre_wt = None
        if ctx.then is not None and dnxt is not None:
            # the THEN projection's own gradients: its weight / bias through the queue (dq^T y), its input gradient
            # dq . then_w joins dy inside the fused launch below (or through a GEMM epilogue on the fallback path)
            then_w, then_b = ctx.then
            y = ctx.saved_tensors[5]
            pre_g = dnxt.contiguous().view(-1, N)
            if pre_g.dtype != a.dtype:
                pre_g = cast(pre_g, a.dtype)
            pre_wt = prep_linear(then_w, a.dtype, True)
            want_db2 = then_b is not None and ctx.needs_input_grad[8]
            if want_db2:
                db2 = zeros_f32((N,), dev)
            if ctx.needs_input_grad[7]:
                dw2 = zeros_f32((N, N), dev)
                WGRADS.tn(WGRADS.site((then_w, then_b)), pre_g, y.view(-1, N), dw2, db2)
            elif want_db2:
                ops.colsum_acc(pre_g, db2)
        if dy is not None:
            dy = dy.contiguous()
        if (FUSED_LINEAR_NORM and s.dtype == torch.bfloat16 and N == 256 and K == 256 and rows <= FUSED_LINEAR_NORM_ROWS
                and ctx.needs_input_grad[0] and not ctx.a_relu):
            # the decoder's query side: layer-norm gradient and the projection's input gradient as ONE launch
            wt = prep_linear(weight, a.dtype, True)
            assert tuple(wt.shape) == (K, N)
            dsum, da = ops.linear_add_norm_bwd(dy, s, mean, rstd, gamma, wt, dg, dbeta, pre_g=pre_g, pre_w_t=pre_wt)
            dsum, da = dsum.view(s.shape), da.view(a.shape)
            g = dsum.view(-1, N)
        else:
            if pre_g is not None:
                dy = ops.gemm_nt(pre_g, pre_wt, residual=None if dy is None else dy.view(-1, N)).view(s.shape)
            dsum = ops.layernorm_bwd(dy.contiguous(), s, mean, rstd, gamma, dg, dbeta)     # d(x + o): feeds both branches
            g = dsum.view(-1, N)
            if ctx.needs_input_grad[0]:
                da = ops.gemm_nt(g, prep_linear(weight, a.dtype, True),
                                 relu_mask=a.view(-1, K) if ctx.a_relu else None).view(a.shape)
        dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[3]
        if want_db:
            db = zeros_f32((N,), dev)
        if ctx.needs_input_grad[2]:
            dw = zeros_f32((N, K), dev)
            WGRADS.tn(WGRADS.site((weight, ctx.bias)), g, a.view(-1, K), dw, db)
        elif want_db:
            ops.colsum_acc(g, db)
        return da, (dsum if ctx.needs_input_grad[1] else None), dw, db, dg, dbeta, None, dw2, db2


def linear_add_norm(a, x, weight, bias, gamma, beta, a_relu=False):
    return LinearAddNormFn.apply(a.contiguous(), x.contiguous(), weight, bias, gamma, beta, a_relu, None, None)


def linear_add_norm_then_fits(a, weight, then_weight):
    """Whether linear_add_norm_then may be used: the fused launch's domain (bf16, 256-wide projections, few rows)."""
    rows = a.numel() // a.shape[-1]
    return (FUSED_LINEAR_NORM and FUSED_LINEAR_THEN and a.dtype == torch.bfloat16 and tuple(weight.shape) == (256, 256)
            and tuple(then_weight.shape) == (256, 256) and rows <= FUSED_LINEAR_NORM_ROWS and _LINEAR_KEEP)


def linear_add_norm_then(a, x, weight, bias, gamma, beta, then_weight, then_bias):
    """(y, y then_weight^T + then_bias) with y = LayerNorm(x + a W^T + b): ONE launch.  With FUSED_LINEAR_PRE the second
    result is a differentiable output of this node (use it as it is: its gradient returns here and the backward launch
    forms dq . then_weight itself); without, it is a constant for autograd -- hand it to
    linear_keep(y, then_weight, then_bias, precomputed=...), whose backward needs nothing else."""
    return LinearAddNormFn.apply(a.contiguous(), x.contiguous(), weight, bias, gamma, beta, False, then_weight, then_bias)


# ------------------------------------------------------------------------------------------------
class ConvFn(Function):
    """y = act(conv2d(x, W) + b) on NHWC activations: a stand-alone trainable convolution (the backbone has its own
    whole-network node, native/backbone.py).  W is an OIHW parameter (any strides with a dense tap plane), stride 1,
    padding k // 2.  Backward: ReLU mask, input gradient, weight gradient (f32 atomics into a [Cout][kh][kw][Cin]
    buffer returned as an OIHW view), bias gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        co, ci, kh, kw = weight.shape
        geom = ops.conv_geom(x.shape, co, kh, 1, kh // 2)
        y = ops.conv2d_fwd(x, prep_conv(weight, x.dtype, None, False), geom, shift=bias, relu=relu)
        ctx.save_for_backward(x, y if relu else None)
        ctx.weight, ctx.geom, ctx.relu, ctx.has_bias = weight, geom, relu, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        weight, geom = ctx.weight, ctx.geom
        co, ci, kh, kw = weight.shape
        g = dy.contiguous()
        if ctx.relu:
            g = ops.eltwise(L.EW_RELU_MASK, g.view(-1, co), y.view(-1, co)).view(dy.shape)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_dgrad(g, prep_conv(weight, x.dtype, None, True), geom)
        if ctx.needs_input_grad[1]:
            buf = zeros_f32((co, kh, kw, ci), x.device)
            ops.conv2d_wgrad_acc(g, x, buf, geom, zeroed=True)
            dw = buf.view(co, ci, 1, 1) if kh == 1 and kw == 1 else buf.permute(0, 3, 1, 2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = zeros_f32((co,), x.device)
            ops.colsum_acc(g.view(-1, co), db)
        return dx, dw, db, None


def conv2d_same(x, weight, bias=None, relu=False, dilation=1):
    """NHWC convolution with padding="same", odd kernel, stride 1 and an optional dilation d.  The conv kernels have
    no dilation argument; a d-dilated convolution IS d*d ordinary convolutions on the pixel-parity sub-grids
    x[:, i::d, j::d] -> y[:, i::d, j::d] (each sub-grid sees its neighbours at distance d as adjacent; zero padding
    k // 2 on the sub-grid = d * (k // 2) on the image), so that is what runs -- exact, launch-heavy, and only used
    by the F2F baseline encoder (reference paper.py:245-261)."""
    x = x.contiguous()
    if dilation == 1:
        return ConvFn.apply(x, weight, bias, relu)
    B, H, W, _ = x.shape
    y = torch.empty((B, H, W, weight.shape[0]), dtype=x.dtype, device=x.device)
    parts = []
    for i in range(min(dilation, H)):
        for j in range(min(dilation, W)):
            parts.append((i, j, ConvFn.apply(x[:, i::dilation, j::dilation].contiguous(), weight, bias, relu)))
    return _ScatterGridFn.apply(y, dilation, *[t for _, _, t in parts])


class _ScatterGridFn(Function):
    """Interleave the d*d sub-grid results back into one image (parts ordered (i, j) row-major); the gradient of a
    part is the matching strided slice."""

    @staticmethod
    def forward(ctx, y, d, *parts):
        ctx.d = d
        k = 0
        H, W = y.shape[1], y.shape[2]
        for i in range(min(d, H)):
            for j in range(min(d, W)):
                y[:, i::d, j::d] = parts[k]
                k += 1
        return y

    @staticmethod
    def backward(ctx, g):
        d = ctx.d
        H, W = g.shape[1], g.shape[2]
        out = [g[:, i::d, j::d].contiguous() for i in range(min(d, H)) for j in range(min(d, W))]
        return (None, None) + tuple(out)


# ------------------------------------------------------------------------------------------------
# Dropout (train mode only; eval and p = 0 never reach these).  No mask tensor: the kernel hashes (seed, index),
# and the backward is the same call on the gradient.  Seeds advance with every call; data-parallel ranks are offset.
class _DropSeeds:
    def __init__(self):
        self.count = 0

    def next(self):
        self.count += 1
        rank = 0
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank = dist.get_rank()
        except Exception:
            rank = 0
        return mix64(torch.initial_seed() * 0x9E3779B97F4A7C15 + (rank << 40) + self.count)


def mix64(x):
    """splitmix64 finaliser.  The kernels hash (index ^ seed_lo) * C + seed_hi: seeds that differ only in their low
    word (a plain call counter) give masks that are index-XOR permutations of one another -- adjacent dropout sites
    and consecutive steps would share one mask up to a neighbour swap.  Every bit of the mixed seed depends on every
    bit of the counter, so consecutive calls differ in both words."""
    x &= 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


DROP_SEEDS = _DropSeeds()


class DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return ops.dropout(x, p, seed)

    @staticmethod
    def backward(ctx, g):
        return ops.dropout(g.contiguous(), ctx.p, ctx.seed), None, None


def dropout(x, p, training):
    """nn.Dropout(p)(x) semantics: identity unless `training` and p > 0."""
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x.contiguous(), float(p), DROP_SEEDS.next())


# fp8 attention (BASELINE.json configs[4], csrc/attention_fp8.hip).  "off": bf16 / f32 kernels everywhere (default);
# "long": the launches the block-shared LDS kernels take in bf16 (long query sequences: the encoder's self-attention,
# the joint encoders) run the MX-fp8 forward; "all": every bf16 attention without active dropout does (the decoder's
# few-query launches then lose their key split across blocks: slower, kept for tests and the record).
# Set by runs/_model.build_model(args.attn_dtype) / bench.py --attn-dtype; FOD_ATTN_FP8 overrides.
ATTN_FP8 = {"mode": os.environ.get("FOD_ATTN_FP8", "off")}


def _fp8_takes(q1, k1, drop_p):
    mode = ATTN_FP8["mode"]
    if mode == "off" or q1.dtype != torch.bfloat16 or drop_p > 0.0:
        return False
    few_queries = q1.shape[1] <= 512 and k1.shape[1] >= 128       # launch_all()'s split predicate (attention.hip)
    return mode == "all" or not few_queries


class AttentionFn(Function):
    @staticmethod
    def forward(ctx, q1, k1, v, q2, k2, scale, drop_p=0.0, drop_seed=0):
        ctx.drop = (drop_p, drop_seed)
        ctx.scale = scale
        ctx.two = q2 is not None
        ctx.fp8 = _fp8_takes(q1, k1, drop_p)
        if ctx.fp8:
            need_bwd = any(t is not None and t.requires_grad for t in (q1, k1, v, q2, k2))
            o, lse2, deq = ops.attn_fwd_fp8(q1, k1, v, scale, q2, k2, want_backward=need_bwd)
            if need_bwd:                 # the backward pass runs on the dequantised operands (same quantised scores)
                ctx.save_for_backward(deq[0], deq[1], deq[2], deq[3], deq[4], o, lse2)
            return o
        o, lse2 = ops.attn_fwd(q1, k1, v, scale, q2, k2, drop_p=drop_p, drop_seed=drop_seed)
        ctx.save_for_backward(q1, k1, v, q2, k2, o, lse2)
        return o

    @staticmethod
    def backward(ctx, do):
        q1, k1, v, q2, k2, o, lse2 = ctx.saved_tensors
        slots = {}
        qk = _column_siblings(q1, k1)
        if qk is not None:
            # q | k are the column halves of one buffer (InProjFn): dq | dk go out the same way
            D = q1.shape[-1]
            dqk = torch.empty_strided(qk.shape, qk.stride(), dtype=qk.dtype, device=qk.device)
            slots = dict(dq1_out=dqk[..., :D], dk1_out=dqk[..., D:])
        elif q1.shape == k1.shape == v.shape and q1.is_contiguous() and k1.is_contiguous() and v.is_contiguous():
            # self-attention: dq | dk | dv as consecutive blocks of one buffer, the layout a GroupLinearFn that
            # produced q, k, v takes as it is (no gather copies)
            buf = torch.empty((3,) + tuple(q1.shape), dtype=q1.dtype, device=q1.device)
            slots = dict(dq1_out=buf[0], dk1_out=buf[1], dv_out=buf[2])
        if ctx.fp8:
            # q was multiplied by scale * log2(e) before quantisation: scores = (1 / log2 e) q'.k, dq = scale log2(e) dq'
            dq1, dk1, dq2, dk2, dv = ops.attn_bwd(q1, k1, v, o, do.contiguous(), lse2, 1.0 / ops.LOG2E, q2, k2,
                                                  dq_scale=ctx.scale * ops.LOG2E, **slots)
            return dq1, dk1, dv, dq2, dk2, None, None, None
        dq1, dk1, dq2, dk2, dv = ops.attn_bwd(q1, k1, v, o, do.contiguous(), lse2, ctx.scale, q2, k2,
                                              drop_p=ctx.drop[0], drop_seed=ctx.drop[1], **slots)
        return dq1, dk1, dv, dq2, dk2, None, None, None


def _strided_ok(t):
    """A [B, T, E] operand the attention kernels take as it is: channel-contiguous, 16-byte strides / pointer."""
    v = _VEC[t.dtype]
    return (t.dim() == 3 and t.stride(-1) == 1 and t.stride(0) % v == 0 and t.stride(1) % v == 0
            and t.data_ptr() % 16 == 0)


def attention(q1, k1, v, scale, q2=None, k2=None, drop_p=0.0, training=False):
    """`drop_p` (with `training`): dropout on the attention probabilities, as MultiheadAttention(dropout=p)."""
    c = lambda t: None if t is None else (t if _strided_ok(t) else t.contiguous())
    if training and drop_p > 0.0:
        return AttentionFn.apply(c(q1), c(k1), c(v), c(q2), c(k2), scale, float(drop_p), DROP_SEEDS.next())
    return AttentionFn.apply(c(q1), c(k1), c(v), c(q2), c(k2), scale)


def _column_siblings(a, b):
    """a = buf[..., :D], b = buf[..., D:2D] of one [..., 2D] buffer?  Returns that buffer (a view) or None."""
    D = a.shape[-1]
    if (b is None or a.shape != b.shape or a.stride() != b.stride() or a.stride(-1) != 1 or a.stride(-2) != 2 * D
            or a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr()
            or b.storage_offset() != a.storage_offset() + D):
        return None
    outer = 2 * D
    for n, st in zip(reversed(a.shape[:-1]), reversed(a.stride()[:-1])):      # dense apart from the column split
        if st != outer:
            return None
        outer *= n
    return torch.as_strided(a, tuple(a.shape[:-1]) + (2 * D,), a.stride())


class InProjFn(Function):
    """torch.nn.MultiheadAttention's packed input projection as the encoder uses it (reference
    transformer.py:401-411: q = k-input = src + pos, v-input = src): q | k from ONE GEMM over xp (N = 2D), v from
    one over src; the packed weight / bias get ONE gradient tensor each (no slice-and-accumulate kernels, and the
    same address every step).  q and k come back as the two column halves of one [.., 2D] buffer; the attention
    kernels read them strided and write dq | dk the same way."""

    @staticmethod
    def forward(ctx, xp, src, w, b):
        D = w.shape[1]
        dtype = xp.dtype
        wqk, wv = prep_linear(w[:2 * D], dtype, False), prep_linear(w[2 * D:], dtype, False)
        qk = ops.gemm_nt(xp, wqk, shift=b[:2 * D]).view(*xp.shape[:-1], 2 * D)
        v = ops.gemm_nt(src, wv, shift=b[2 * D:]).view(src.shape)
        ctx.save_for_backward(xp, src)
        ctx.w, ctx.b = w, b
        return qk[..., :D], qk[..., D:], v

    @staticmethod
    def backward(ctx, dq, dk, dv):
        xp, src = ctx.saved_tensors
        w, b = ctx.w, ctx.b
        D = w.shape[1]
        dtype = xp.dtype
        rows = xp.numel() // D
        dqk = _column_siblings(dq, dk)
        if dqk is None:
            dqk = torch.cat([dq, dk], dim=-1)
        dqk = dqk.reshape(rows, 2 * D)
        dv = dv.contiguous().view(rows, D)
        dxp = ops.gemm_nt(dqk, prep_linear(w[:2 * D], dtype, True)).view(xp.shape) if ctx.needs_input_grad[0] else None
        dsrc = ops.gemm_nt(dv, prep_linear(w[2 * D:], dtype, True)).view(src.shape) if ctx.needs_input_grad[1] else None
        dw = zeros_f32((3 * D, D), xp.device)
        db = zeros_f32((3 * D,), xp.device)
        wq = WGRADS.site((w, b))
        WGRADS.tn(wq, dqk, xp.view(rows, D), dw[:2 * D], db[:2 * D])
        WGRADS.tn(wq, dv, src.view(rows, D), dw[2 * D:], db[2 * D:])
        return dxp, dsrc, dw, db


def in_proj(xp, src, weight, bias):
    return InProjFn.apply(xp.contiguous(), src.contiguous(), weight, bias)


class InProjSelfFn(Function):
    """The encoder's self-attention input side as ONE autograd consumer of src: xp = src + pos, q | k = xp W_qk^T, v =
    src W_v^T, and src handed back for the residual step.  Separately, src has three consumers (the positional add,
    the v projection, the residual) and autograd sums their gradients with two kernels of its own per layer (14500 x 256
    elements each); here the residual gradient rides in the q | k input-gradient GEMM's epilogue and that result in the
    v one's.  `pos` (a table or per-frame tensor without gradient) is broadcast by the add kernel."""

    @staticmethod
    def forward(ctx, src, pos, w, b, row_mod):
        D = w.shape[1]
        dtype = src.dtype
        xp = ops.eltwise(L.EW_ADD, src, pos, b_row_mod=row_mod)
        wqk, wv = prep_linear(w[:2 * D], dtype, False), prep_linear(w[2 * D:], dtype, False)
        qk = ops.gemm_nt(xp, wqk, shift=b[:2 * D]).view(*xp.shape[:-1], 2 * D)
        v = ops.gemm_nt(src, wv, shift=b[2 * D:]).view(src.shape)
        ctx.save_for_backward(xp, src)
        ctx.w, ctx.b = w, b
        return src.view_as(src), qk[..., :D], qk[..., D:], v

    @staticmethod
    def backward(ctx, dkeep, dq, dk, dv):
        xp, src = ctx.saved_tensors
        w, b = ctx.w, ctx.b
        D = w.shape[1]
        dtype = xp.dtype
        rows = xp.numel() // D
        dqk = _column_siblings(dq, dk)
        if dqk is None:
            dqk = torch.cat([dq, dk], dim=-1)
        dqk = dqk.reshape(rows, 2 * D)
        dv = dv.contiguous().view(rows, D)
        dsrc = None
        if ctx.needs_input_grad[0]:
            res = None if dkeep is None else dkeep.contiguous().view(rows, D)
            if res is not None and res.dtype != dtype:
                res = cast(res, dtype)
            dxp = ops.gemm_nt(dqk, prep_linear(w[:2 * D], dtype, True), residual=res)
            dsrc = ops.gemm_nt(dv, prep_linear(w[2 * D:], dtype, True), residual=dxp).view(src.shape)
        dw = zeros_f32((3 * D, D), xp.device)
        db = zeros_f32((3 * D,), xp.device)
        wq = WGRADS.site((w, b))
        WGRADS.tn(wq, dqk, xp.view(rows, D), dw[:2 * D], db[:2 * D])
        WGRADS.tn(wq, dv, src.view(rows, D), dw[2 * D:], db[2 * D:])
        return dsrc, None, dw, db, None


def in_proj_self(src, pos, weight, bias):
    """-> (src for the residual step, q, k, v) with q = k-input = src + pos (see InProjSelfFn); None if `pos` carries a
    gradient or the fused form is switched off (the caller then takes the separate add + in_proj)."""
    if pos.requires_grad or not _LINEAR_KEEP:
        return None
    row_mod = pos.shape[-2] if pos.dim() == 2 else 0
    return InProjSelfFn.apply(src.contiguous(), pos.contiguous(), weight, bias, row_mod)


class InProjCrossFn(Function):
    """The packed input projection with three different inputs (encoder cross-attention onto the previous output /
    a previous frame, reference transformer.py:464-478: q-input = x + pos, k-input = other + pos, v-input = other).
    Three GEMMs; the packed weight / bias still get ONE gradient tensor each."""

    @staticmethod
    def forward(ctx, xq, xk, xv, w, b):
        D = w.shape[1]
        dtype = xq.dtype
        q = ops.gemm_nt(xq, prep_linear(w[:D], dtype, False), shift=b[:D]).view(xq.shape)
        k = ops.gemm_nt(xk, prep_linear(w[D:2 * D], dtype, False), shift=b[D:2 * D]).view(xk.shape)
        v = ops.gemm_nt(xv, prep_linear(w[2 * D:], dtype, False), shift=b[2 * D:]).view(xv.shape)
        ctx.save_for_backward(xq, xk, xv)
        ctx.w, ctx.b = w, b
        return q, k, v

    @staticmethod
    def backward(ctx, dq, dk, dv):
        xs = ctx.saved_tensors
        w, b = ctx.w, ctx.b
        D = w.shape[1]
        dtype = xs[0].dtype
        dw = zeros_f32((3 * D, D), xs[0].device)
        db = zeros_f32((3 * D,), xs[0].device)
        dxs = []
        wq = WGRADS.site((w, b))
        for i, (x, g) in enumerate(zip(xs, (dq, dk, dv))):
            rows = x.numel() // D
            g = g.contiguous().view(rows, D)
            sl = slice(i * D, (i + 1) * D)
            dxs.append(ops.gemm_nt(g, prep_linear(w[sl], dtype, True)).view(x.shape) if ctx.needs_input_grad[i] else None)
            WGRADS.tn(wq, g, x.view(rows, D), dw[sl], db[sl])
        return dxs[0], dxs[1], dxs[2], dw, db


def in_proj_cross(xq, xk, xv, weight, bias):
    return InProjCrossFn.apply(xq.contiguous(), xk.contiguous(), xv.contiguous(), weight, bias)


class RefPointSineFn(Function):
    """ref_logit [R,2] -> (ref f32 [R,2] = sigmoid, sine [R,D] ordered (y | x))."""

    @staticmethod
    def forward(ctx, ref_logit, D):
        ref, sine = ops.refpoint_sine_fwd(ref_logit, D)
        ctx.save_for_backward(ref)
        ctx.D, ctx.act_dtype = D, ref_logit.dtype
        return ref, sine

    @staticmethod
    def backward(ctx, dref, dsine):
        (ref,) = ctx.saved_tensors
        if dsine is None:
            dsine = torch.zeros((ref.shape[0], ctx.D), dtype=ctx.act_dtype, device=ref.device)
        return ops.refpoint_sine_bwd(dsine.contiguous(), ref, None if dref is None else dref.contiguous()), None


class BoxFinishFn(Function):
    """boxes f32 [L,R,4] = sigmoid(t + [inverse_sigmoid(ref[r % ref_rows]), 0, 0])."""

    @staticmethod
    def forward(ctx, t, ref, levels):
        boxes = ops.box_finish_fwd(t, ref, levels)
        ctx.save_for_backward(boxes, ref)
        ctx.act_dtype = t.dtype
        ctx.t_shape = t.shape
        return boxes

    @staticmethod
    def backward(ctx, dboxes):
        boxes, ref = ctx.saved_tensors
        dt_, dref = ops.box_finish_bwd(dboxes.contiguous(), boxes, ref, ctx.act_dtype)
        return dt_.view(ctx.t_shape), dref, None


class ZeroGradAnchor(Function):
    """Identity on `x`; gives the listed parameters an explicit all-zero gradient.

    Used for parameters whose gradient is identically zero by construction (projections that only
    feed the logits of a softmax over ONE key): the reference materialises zeros for them, and an
    optimizer with decoupled weight decay treats zero and None differently."""

    @staticmethod
    def forward(ctx, x, *params):
        ctx.shapes = [(p.shape, p.device) for p in params]
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (g,) + tuple(zeros_f32(s, d) for s, d in ctx.shapes)


class CastFn(Function):
    """dtype conversion of a [*, cols] tensor with optional zero padding of the last dim; the gradient
    is converted back (and un-padded)."""

    @staticmethod
    def forward(ctx, x, dtype, pad_cols):
        ctx.src_dtype, ctx.cols = x.dtype, x.shape[-1]
        return cast(x, dtype, pad_cols)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        cols = g.shape[-1]
        rows = g.numel() // cols
        out = ops.permute3_cast(g, ctx.src_dtype, (1, rows, ctx.cols), (0, cols, 1))
        return out.view(*g.shape[:-1], ctx.cols), None, None


def cast_ad(x, dtype, pad_cols=None):
    if x.dtype == dtype and pad_cols in (None, x.shape[-1]):
        return x
    return CastFn.apply(x.contiguous(), dtype, pad_cols)


# ------------------------------------------------------------------------------------------------
# Memory side of the decoder, hoisted out of the layer loop.
#
# The value / key_content projections of an image memory and the key_pos projection of the positional
# table do not depend on the decoder state, only on (memory, weights).  Instead of 2*L small GEMMs per
# image inside the layer loop (plus their 4*L backward GEMMs), every image gets ONE GEMM against the
# row-concatenated weights of all layers, the positional table gets ONE GEMM for all (layer, image)
# pairs, and each cross-attention reads / writes its [*, D] column slot of those wide buffers through
# the attention kernels' stride arguments.
# ------------------------------------------------------------------------------------------------
class _CatCache:
    """P same-shaped Linear layers stacked: (wcat [P*D_out, D_in], bcat f32 [P*D_out], wcat_t [D_in, P*D_out])."""

    def get(self, weights, biases, dtype):
        D_out, D_in = weights[0].shape
        P = len(weights)

        def build():
            dev = weights[0].device
            wcat = torch.empty((P * D_out, D_in), dtype=dtype, device=dev)
            bcat = torch.empty((P * D_out,), dtype=torch.float32, device=dev)
            wcat_t = torch.empty((D_in, P * D_out), dtype=dtype, device=dev)
            jobs = []
            for i, (w, b) in enumerate(zip(weights, biases)):
                wd, bd = w.detach(), b.detach()
                sn, sk = wd.stride()
                jobs.append(_Job(wd, wcat[i * D_out:(i + 1) * D_out], (1, D_out, D_in), (0, sn, sk)))
                jobs.append(_Job(bd, bcat[i * D_out:(i + 1) * D_out], (1, 1, D_out), (0, 0, bd.stride(0))))
                # column block i of the transposed stack: dst[k][i*D_out + n] = w[n][k]
                jobs.append(_Job(wd, wcat_t[:, i * D_out:(i + 1) * D_out], (1, D_in, D_out), (0, sk, sn),
                                 dstr=(0, P * D_out)))
            return (wcat, bcat, wcat_t), jobs

        key = (tuple(w.data_ptr() for w in weights), dtype, "cat")
        return PREP.get(key, list(weights) + list(biases), build)


CAT = _CatCache()


class WideLinearFn(Function):
    """y[*, P*D] = x[*, Din] @ cat_p(W_p)^T + cat_p(b_p): P Linear layers sharing one input, one GEMM.
    `x_grad=False` for an input without gradient (the positional table).  When `batch_sum` > 1 the
    incoming gradient is [batch_sum, rows, P*D] although x is [rows, Din] (x was shared by the batch)."""

    @staticmethod
    def forward(ctx, x, batch_sum, *params):
        P = len(params) // 2
        weights, biases = params[:P], params[P:]
        wcat, bcat, _ = CAT.get(weights, biases, x.dtype)
        y = ops.gemm_nt(x, wcat, shift=bcat).view(*x.shape[:-1], wcat.shape[0])
        ctx.save_for_backward(x)
        ctx.params, ctx.batch_sum = params, batch_sum
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        P = len(ctx.params) // 2
        weights, biases = ctx.params[:P], ctx.params[P:]
        D_out, D_in = weights[0].shape
        _, _, wcat_t = CAT.get(weights, biases, x.dtype)
        dy = dy.contiguous()
        if ctx.batch_sum > 1:
            dy = _sum_leading(dy, ctx.batch_sum)
        g = dy.view(-1, P * D_out)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(g, wcat_t).view(x.shape)
        dw = zeros_f32((P * D_out, D_in), x.device)
        db = zeros_f32((P * D_out,), x.device)
        WGRADS.tn(WGRADS.site(ctx.params), g, x.view(-1, D_in), dw, db)
        gw = [dw[i * D_out:(i + 1) * D_out] for i in range(P)]
        gb = [db[i * D_out:(i + 1) * D_out] for i in range(P)]
        return (dx, None) + tuple(gw) + tuple(gb)


class GroupLinearFn(Function):
    """P same-shaped Linear layers applied to ONE input, returned as P separate contiguous tensors (blocks of one
    [P, rows, D] buffer): one launch forward, one for the input gradient (the P output gradients are the k-segments
    of its A operand), one for all weight / bias gradients.  The decoder's query side is made of these: the
    self-attention's query_content / key_content / value, the per-image query_sine projections of a layer, and
    the per-layer projections of the learned query positions (reference transformer.py:66-70,139-141)."""

    @staticmethod
    def forward(ctx, x, nres, res_mod, *tensors):
        """`nres` > 0: the first nres of `tensors` are [R, D] tables (consecutive blocks of ONE buffer, as an earlier
        GroupLinearFn returns them) added to the first nres outputs in the launch's epilogue, row m + table[m % res_mod]
        -- the self-attention's q = q_content + q_pos, k = k_content + k_pos without their two element-wise launches."""
        res, params = tensors[:nres], tensors[nres:]
        P = len(params) // 2
        weights, biases = params[:P], params[P:]
        wcat, bcat, _ = CAT.get(weights, biases, x.dtype)
        D = weights[0].shape[0]
        rblock = None
        if nres:
            rblock = _as_segments(res, res[0].numel() // D, D)
            assert rblock is not None, "group_linear(residual=...): the tables must be consecutive blocks (the caller checks)"
        y = ops.group_linear_fwd(x.view(-1, x.shape[-1]), wcat, bcat, P, residual=rblock, res_row_mod=res_mod)
        ctx.save_for_backward(x)
        ctx.params, ctx.nres, ctx.res_mod, ctx.res_shape = params, nres, res_mod, (res[0].shape if nres else None)
        return tuple(y[p].view(*x.shape[:-1], D) for p in range(P))

    @staticmethod
    def backward(ctx, *gs):
        (x,) = ctx.saved_tensors
        P = len(ctx.params) // 2
        weights, biases = ctx.params[:P], ctx.params[P:]
        D, K = weights[0].shape
        _, _, wcat_t = CAT.get(weights, biases, x.dtype)
        rows = x.numel() // K
        g = _as_segments(gs, rows, D)
        if g is None:                                # gather the P gradients into the segment layout: ONE launch
            g = torch.empty((P, rows, D), dtype=x.dtype, device=x.device)
            dsts, srcs = [], []
            for p_, gp in enumerate(gs):
                if gp is None:
                    g[p_].zero_()
                else:
                    dsts.append(g[p_])
                    srcs.append(gp.reshape(rows, D))
            if all(t.dtype == x.dtype and t.is_contiguous() for t in srcs):
                torch._foreach_copy_(dsts, srcs)     # (17 separate copies of 64 KB were 17 graph nodes of ~5 us each)
            else:
                for d_, s_ in zip(dsts, srcs):
                    d_.copy_(s_)
        dx = ops.group_linear_dgrad(g, wcat_t, P).view(x.shape) if ctx.needs_input_grad[0] else None
        dw = zeros_f32((P * D, K), x.device)
        db = zeros_f32((P * D,), x.device)
        WGRADS.grouped(WGRADS.site(ctx.params), g, x.view(rows, K), dw, db)
        dres = ()
        if ctx.nres:
            # the tables' gradients: the first nres output gradients summed over the rows that shared a table row
            dres = tuple((_sum_periodic(g[i], ctx.res_mod).view(ctx.res_shape) if ctx.res_mod and rows > ctx.res_mod
                          else g[i].view(ctx.res_shape)) if ctx.needs_input_grad[3 + i] else None for i in range(ctx.nres))
        return ((dx, None, None) + dres + tuple(dw[i * D:(i + 1) * D] for i in range(P))
                + tuple(db[i * D:(i + 1) * D] for i in range(P)))


def _as_segments(gs, rows, D):
    """The tensors of `gs` as one [P, rows, D] tensor if they already are consecutive contiguous blocks of one
    allocation (producers that know about the grouping write their gradients that way), else None."""
    g0 = gs[0]
    if g0 is None or not g0.is_contiguous() or g0.numel() != rows * D:
        return None
    st, step = g0.untyped_storage(), rows * D
    for p_, gp in enumerate(gs):
        if (gp is None or gp.dtype != g0.dtype or not gp.is_contiguous() or gp.numel() != step
                or gp.untyped_storage().data_ptr() != st.data_ptr()
                or gp.storage_offset() != g0.storage_offset() + p_ * step):
            return None
    return torch.as_strided(g0, (len(gs), rows, D), (step, D, 1))


# "0": the self-attention's position adds as their own element-wise launches (not in the grouped projection's epilogue)
GROUP_RESIDUAL = os.environ.get("FOD_GROUP_RESIDUAL", "1") != "0"


def group_linear(x, linears, residual=None, res_row_mod=0):
    """[m(x) for m in linears] for nn.Linear-like holders of equal shape; grouped launches in bf16, plain
    per-layer launches otherwise (fp32 parity mode, odd widths).  `residual`: a list of [R, D] tables added to the first
    len(residual) results, row m + table[m % res_row_mod] -- inside the launch when the tables are consecutive blocks of
    one buffer (what an earlier group_linear returns), as separate additions otherwise."""
    D, K = linears[0].weight.shape
    same = all(tuple(m.weight.shape) == (D, K) and m.bias is not None for m in linears)
    res = list(residual) if residual is not None else []
    if x.dtype != torch.bfloat16 or not same or D % 64 or K % 32 or len(linears) < 2:
        out = [linear(x, m.weight, m.bias) for m in linears]
        fused = False
    else:
        fused = bool(res) and GROUP_RESIDUAL and all(r.dim() == 2 and r.shape[1] == D and r.dtype == x.dtype for r in res) \
            and _as_segments(res, res[0].shape[0], D) is not None and 0 < res_row_mod <= res[0].shape[0]
        out = list(GroupLinearFn.apply(x.contiguous(), len(res) if fused else 0, res_row_mod if fused else 0,
                                       *(res if fused else []), *[m.weight for m in linears], *[m.bias for m in linears]))
    if res and not fused:
        for i, r in enumerate(res):
            out[i] = add(out[i], r, b_row_mod=res_row_mod)
    return out


class _StackCache:
    """P same-length f32 vectors (the LayerNorm weights / biases of P layers) as one [P, D] table, refreshed with the
    other prepared operands."""

    def get(self, vecs):
        def build():
            D = vecs[0].numel()
            table = torch.empty((len(vecs), D), dtype=torch.float32, device=vecs[0].device)
            jobs = []
            for i, v in enumerate(vecs):
                vd = v.detach()
                jobs.append(_Job(vd, table[i], (1, 1, D), (0, 0, vd.stride(0))))
            return (table,), jobs

        key = (tuple(v.data_ptr() for v in vecs), torch.float32, "stack")
        return PREP.get(key, list(vecs), build)[0]


STACK = _StackCache()

# "0": every encoder layer computes its IMU-token block itself (4 GEMMs + 2 norms on [frames, D] rows per layer)
IMU_BATCHED = os.environ.get("FOD_IMU_BATCHED", "1") != "0"


class ImuBranchFn(Function):
    """The one-key IMU-token blocks of P transformer layers at once (EgodeepAttention.forward_single_key of every layer;
    reference transformer.py:108-119 with one key per frame, where softmax = 1 and the block reduces to
    value -> out_proj [-> norm1(o + o) -> MLP -> norm2]).  The blocks read only the IMU embedding `ego` [frames, D] -- not
    the layer's tokens -- so the P of them are independent: every step of the chain runs for all layers in ONE launch
    (fod_gemm_nt_grouped for the shared input, fod_gemm_nt_batched after it, the norms with per-layer parameter tables).
    Per layer that was 4 GEMMs + 2 norms forward and 4 + 2 + 2 backward on ~10 rows, each a ~5-9 us graph node.

    apply(ego, P, use_mlp, *params) with params = P value weights, P value biases, P out_proj weights, P biases and, with
    use_mlp, norm1 weights, biases, mlp[0] weights, biases, mlp[3] weights, biases, norm2 weights, biases (P each)
    -> P tensors [frames, D] (blocks of one buffer)."""

    @staticmethod
    def forward(ctx, ego, P, use_mlp, queue, *params):
        """`queue` (ResGradQueue or None): where the consumers of the P results leave their gradients' sums pending."""
        ctx.queue = queue
        groups = [params[i * P:(i + 1) * P] for i in range(len(params) // P)]
        wv, bv, wo, bo = groups[:4]
        M, D = ego.shape
        dtype = ego.dtype
        t = ops.group_linear_fwd(ego, *CAT.get(wv, bv, dtype)[:2], P)                       # [P, M, D]
        wo_cat, bo_cat, _ = CAT.get(wo, bo, dtype)
        o = ops.gemm_nt_batched(t, wo_cat, shift=bo_cat)
        saved = [ego, t]
        if use_mlp:
            g1, b1, w0, b0, w3, b3, g2, b2 = groups[4:]
            o2 = o.view(P * M, D)
            n1, s1, mean1, rstd1 = ops.layernorm_fwd(o2, STACK.get(g1), STACK.get(b1), residual=o2, group_rows=M)
            w0_cat, b0_cat, _ = CAT.get(w0, b0, dtype)
            h = ops.gemm_nt_batched(n1.view(P, M, D), w0_cat, shift=b0_cat, relu=True)      # [P, M, Dff]
            w3_cat, b3_cat, _ = CAT.get(w3, b3, dtype)
            y = ops.gemm_nt_batched(h, w3_cat, shift=b3_cat)
            e, s2, mean2, rstd2 = ops.layernorm_fwd(n1, STACK.get(g2), STACK.get(b2), residual=y.view(P * M, D),
                                                    group_rows=M)
            saved += [s1, mean1, rstd1, n1, h, s2, mean2, rstd2]
            e = e.view(P, M, D)
        else:
            e = o
        ctx.save_for_backward(*saved)
        ctx.P, ctx.use_mlp, ctx.params = P, use_mlp, params
        return tuple(e[p] for p in range(P))

    @staticmethod
    def backward(ctx, *gs):
        P, use_mlp, params = ctx.P, ctx.use_mlp, ctx.params
        groups = [params[i * P:(i + 1) * P] for i in range(len(params) // P)]
        wv, bv, wo, bo = groups[:4]
        saved = ctx.saved_tensors
        ego, t = saved[:2]
        M, D = ego.shape
        dtype, dev = ego.dtype, ego.device
        if ctx.queue is not None:
            ctx.queue.flush()                                # the slots handed out as gradients become real
        de = _as_segments(gs, M, D)
        if de is None or de.dtype != dtype:
            de = torch.empty((P, M, D), dtype=dtype, device=dev)
            dsts, srcs = [], []
            for p_, gp in enumerate(gs):
                if gp is None:
                    de[p_].zero_()
                else:
                    dsts.append(de[p_])
                    srcs.append(gp.reshape(M, D))
            if all(s_.dtype == dtype and s_.is_contiguous() for s_ in srcs):
                torch._foreach_copy_(dsts, srcs)
            else:
                for d_, s_ in zip(dsts, srcs):
                    d_.copy_(s_)

        def wgrads(ws, bs, g, x):
            """queued weight / bias gradients of P layers with their own inputs: g [P, M, N], x [P, M, K]"""
            N, K = ws[0].shape
            dw, db = zeros_f32((P * N, K), dev), zeros_f32((P * N,), dev)
            for p_ in range(P):
                WGRADS.tn(WGRADS.site((ws[p_], bs[p_])), g[p_], x[p_], dw[p_ * N:(p_ + 1) * N], db[p_ * N:(p_ + 1) * N])
            return [dw[p_ * N:(p_ + 1) * N] for p_ in range(P)], [db[p_ * N:(p_ + 1) * N] for p_ in range(P)]

        extra = []
        if use_mlp:
            g1, b1, w0, b0, w3, b3, g2, b2 = groups[4:]
            s1, mean1, rstd1, n1, h, s2, mean2, rstd2 = saved[2:]
            Dff = w0[0].shape[0]
            dg2, db2 = zeros_f32((P * D,), dev), zeros_f32((P * D,), dev)
            ds2 = ops.layernorm_bwd(de.view(P * M, D), s2, mean2, rstd2, STACK.get(g2), dg2, db2, group_rows=M)
            ds2 = ds2.view(P, M, D)                          # gradient of n1 + mlp(n1): the MLP output's and n1's residual share
            # the ReLU's gate rides in the epilogue of the GEMM that forms the hidden gradient
            dh = ops.gemm_nt_batched(ds2, CAT.get(w3, b3, dtype)[2], b_cols=True, relu_mask=h)          # [P, M, Dff]
            dw3, db3 = wgrads(w3, b3, ds2, h)
            dn1 = ops.gemm_nt_batched(dh, CAT.get(w0, b0, dtype)[2], b_cols=True, residual=ds2)         # + the residual path
            dw0, db0 = wgrads(w0, b0, dh, n1.view(P, M, D))
            dg1, db1 = zeros_f32((P * D,), dev), zeros_f32((P * D,), dev)
            ds1 = ops.layernorm_bwd(dn1.view(P * M, D), s1, mean1, rstd1, STACK.get(g1), dg1, db1, group_rows=M)
            do = ops.eltwise(L.EW_SCALE, ds1, alpha=2.0).view(P, M, D)       # norm1(o + o): o enters twice
            sl = lambda v: [v[p_ * D:(p_ + 1) * D] for p_ in range(P)]
            extra = sl(dg1) + sl(db1) + dw0 + db0 + dw3 + db3 + sl(dg2) + sl(db2)
        else:
            do = de
        dt_ = ops.gemm_nt_batched(do, CAT.get(wo, bo, dtype)[2], b_cols=True)                           # [P, M, D]
        dwo, dbo = wgrads(wo, bo, do, t)
        dego = ops.group_linear_dgrad(dt_, CAT.get(wv, bv, dtype)[2], P) if ctx.needs_input_grad[0] else None
        dwv, dbv = zeros_f32((P * D, D), dev), zeros_f32((P * D,), dev)
        WGRADS.grouped(WGRADS.site(tuple(wv) + tuple(bv)), dt_, ego, dwv, dbv)
        sl = lambda v: [v[p_ * D:(p_ + 1) * D] for p_ in range(P)]
        return (dego, None, None, None) + tuple(sl(dwv) + sl(dbv) + dwo + dbo + extra)


def imu_branch_fits(ego, blocks):
    """Whether the one-key IMU blocks `blocks` (EgodeepAttention-like holders) can run as ImuBranchFn: bf16, >= 2 same-shaped
    blocks with biases, widths inside the short-launch kernel's vector epilogue."""
    if not IMU_BATCHED or len(blocks) < 2 or ego.dim() != 2 or ego.dtype != torch.bfloat16:
        return False
    D = ego.shape[1]
    b0 = blocks[0]
    for b in blocks:
        if b.use_mlp != b0.use_mlp:
            return False
        lins = [b.value, b.fun.out_proj] + ([b.mlp[0], b.mlp[3]] if b.use_mlp else [])
        if any(m.bias is None for m in lins):
            return False
        if tuple(b.value.weight.shape) != (D, D) or tuple(b.fun.out_proj.weight.shape) != (D, D):
            return False
        if b.use_mlp and (tuple(b.mlp[0].weight.shape) != tuple(b0.mlp[0].weight.shape)
                          or tuple(b.mlp[3].weight.shape) != (D, b0.mlp[0].weight.shape[0])):
            return False
    return D % 64 == 0 and (not b0.use_mlp or b0.mlp[0].weight.shape[0] % 64 == 0)


def imu_branch(ego, blocks, queue=None):
    """[block.forward_single_key(ego) without dropout for block in blocks] in a handful of launches (ImuBranchFn)."""
    P = len(blocks)
    use_mlp = blocks[0].use_mlp
    params = ([b.value.weight for b in blocks] + [b.value.bias for b in blocks]
              + [b.fun.out_proj.weight for b in blocks] + [b.fun.out_proj.bias for b in blocks])
    if use_mlp:
        params += ([b.norm1.weight for b in blocks] + [b.norm1.bias for b in blocks]
                   + [b.mlp[0].weight for b in blocks] + [b.mlp[0].bias for b in blocks]
                   + [b.mlp[3].weight for b in blocks] + [b.mlp[3].bias for b in blocks]
                   + [b.norm2.weight for b in blocks] + [b.norm2.bias for b in blocks])
    return list(ImuBranchFn.apply(ego.contiguous(), P, use_mlp, queue, *params))


def _sum_leading(t, n):
    """[n, ...] -> [...] : sum over the (tiny) leading dim."""
    acc = t[0]
    for i in range(1, n):
        acc = ops.eltwise(L.EW_ADD, acc.contiguous(), t[i].contiguous())
    return acc.contiguous()


def wide_linear(x, linears, batch_sum=1):
    ws = [m.weight for m in linears]
    bs = [m.bias for m in linears]
    return WideLinearFn.apply(x.contiguous(), batch_sum, *ws, *bs)


class MemorySide:
    """Per-forward holder of the hoisted projections and of the gradient buffers their slots are written to."""

    def __init__(self, big, ks_all, n_layers, n_images, D):
        self.big = big              # list over images: [B, N, 2*L*D]  (per layer: [value | key_content])
        self.ks_all = ks_all        # [N, L*K*D]                         (per (layer, image): key_pos(pos))
        self.L, self.K, self.D = n_layers, n_images, D
        self.dbig = [None] * n_images
        self.dks = None
        self.dq2 = {}

    def slots(self, layer, image):
        D = self.D
        b = self.big[image]
        v = b[..., (2 * layer) * D:(2 * layer + 1) * D]
        kc = b[..., (2 * layer + 1) * D:(2 * layer + 2) * D]
        q = layer * self.K + image
        ks = self.ks_all[..., q * D:(q + 1) * D]      # [N, D] table shared by the batch, or [B, N, D] (temporal term)
        return kc, ks, v


# "0": every cross-attention block launches its own dk / dv pass (fod_attn_bwd) instead of queueing it
DKV_QUEUE = os.environ.get("FOD_DKV_QUEUE", "1") != "0"


class _DkvQueue:
    """The dk / dv passes of the decoder's cross-attention blocks (one per layer and image: 30 per step) wait here during the
    backward sweep and run as ONE launch (fod_attn_bwd_dkv_multi; 32 jobs at most per launch) when the first consumer of
    their results is reached: each writes its own (layer, image) slot of MemorySide.dbig / .dks, which nothing reads
    before the call that hands the buffers to autograd (layer 0)."""

    def __init__(self):
        self.jobs, self.key, self.shp = [], None, None

    def push(self, key, shp, job):
        if self.jobs and (key != self.key or len(self.jobs) == 32):
            self.flush()
        self.key, self.shp = key, shp
        self.jobs.append(job)

    def flush(self):
        if self.jobs:
            jobs, self.jobs = self.jobs, []
            ops.attn_bwd_dkv_multi(jobs, self.shp)


class HoistedCrossAttnFn(Function):
    """softmax((q1.kc + q2.ks) * scale) v with kc / v / ks taken as column slots of the hoisted buffers.
    Gradients of the slots are written in place into MemorySide.dbig / .dks; the buffers are handed to
    autograd once, by the call that runs LAST in the backward sweep (layer 0; image 0 for ks), when every
    slot has been filled -- all earlier (higher-layer) calls return None for them."""

    @staticmethod
    def forward(ctx, q1, q2, big_j, ks_all, side, layer, image, scale, drop_p=0.0, drop_seed=0):
        kc, ks, v = side.slots(layer, image)
        o, lse2 = ops.attn_fwd(q1, kc, v, scale, q2, ks, drop_p=drop_p, drop_seed=drop_seed)
        ctx.save_for_backward(q1, q2, o, lse2)
        ctx.side, ctx.layer, ctx.image, ctx.scale = side, layer, image, scale
        ctx.drop = (drop_p, drop_seed)
        return o

    @staticmethod
    def backward(ctx, do):
        q1, q2, o, lse2 = ctx.saved_tensors
        side, layer, image, D = ctx.side, ctx.layer, ctx.image, ctx.side.D
        kc, ks, v = side.slots(layer, image)
        if side.dbig[image] is None:
            side.dbig[image] = torch.empty_like(side.big[image])
        per_batch_pos = side.ks_all.dim() == 3
        if side.dks is None:
            B = q1.shape[0]
            shape = tuple(side.ks_all.shape) if per_batch_pos else (B,) + tuple(side.ks_all.shape)
            side.dks = torch.empty(shape, dtype=side.ks_all.dtype, device=q1.device)
        db = side.dbig[image]
        dv = db[..., (2 * layer) * D:(2 * layer + 1) * D]
        dkc = db[..., (2 * layer + 1) * D:(2 * layer + 2) * D]
        qi = layer * side.K + image
        dks = side.dks[..., qi * D:(qi + 1) * D]
        # dq2 of the K images of a layer as consecutive blocks (their query_sine projections are one GroupLinearFn)
        key = (layer, tuple(q2.shape))
        if key not in side.dq2:
            side.dq2[key] = torch.empty((side.K,) + tuple(q2.shape), dtype=q2.dtype, device=q2.device)
        B_, Tq_, S_ = q1.shape[0], q1.shape[1], kc.shape[1]
        if (DKV_QUEUE and q1.dtype == torch.bfloat16 and ctx.drop[0] == 0.0 and Tq_ <= 512 and S_ >= 256
                and dks.dim() == 3):
            # (the few-query shapes: what fod_attn_bwd would run as dq + attn_bwd_dkv_pf_kernel; the same kernel body runs
            # the queued jobs, so the results are bit-equal)
            do_c = do.contiguous()
            dq1, dq2, delta, shp = ops.attn_bwd_dq(q1, kc, v, o, do_c, lse2, ctx.scale, q2, ks, dk2_like=dks,
                                                   dq2_out=side.dq2[key][image])
            if getattr(side, "dkv_queue", None) is None:
                side.dkv_queue = _DkvQueue()
            qkey = (tuple(q1.shape), q1.stride(), kc.stride(), v.stride(), ks.stride(), dks.stride(), tuple(kc.shape),
                    float(ctx.scale))
            side.dkv_queue.push(qkey, shp, (q1, q2, kc, ks, v, do_c, lse2, delta, dkc, dks, dv))
            if layer == 0:
                side.dkv_queue.flush()                  # this call hands dbig[image] (and, at image 0, dks) on
        else:
            if getattr(side, "dkv_queue", None) is not None:
                side.dkv_queue.flush()
            dq1, _, dq2, _, _ = ops.attn_bwd(q1, kc, v, o, do.contiguous(), lse2, ctx.scale, q2, ks,
                                             dk1_out=dkc, dv_out=dv, dk2_out=dks, dq2_out=side.dq2[key][image],
                                             drop_p=ctx.drop[0], drop_seed=ctx.drop[1])
        g_big = side.dbig[image] if layer == 0 else None
        # a batch-shared ks_all gets the sum of the per-batch slots
        g_ks = None
        if layer == 0 and image == 0:
            g_ks = side.dks if per_batch_pos else _sum_leading(side.dks, side.dks.shape[0])
        return dq1, dq2, g_big, g_ks, None, None, None, None, None, None


def hoisted_cross_attention(q1, q2, side, layer, image, scale, drop_p=0.0, training=False):
    if training and drop_p > 0.0:
        return HoistedCrossAttnFn.apply(q1.contiguous(), q2.contiguous(), side.big[image], side.ks_all, side, layer,
                                        image, scale, float(drop_p), DROP_SEEDS.next())
    return HoistedCrossAttnFn.apply(q1.contiguous(), q2.contiguous(), side.big[image], side.ks_all, side, layer,
                                    image, scale)
