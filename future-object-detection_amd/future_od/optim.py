"""AdamW with global-norm gradient clipping as two HIP launches per step (all parameters at once).

Same update rule and state layout as torch.optim.AdamW (the reference's optimizer, reference
runs/_helper.py:105) and torch.nn.utils.clip_grad_norm_ (reference future_od/trainer.py:186-187);
`state_dict()` / `load_state_dict()` interchange with torch's, so reference checkpoints resume.
"""
import ctypes as C
import math

import torch

from future_od.native import lib as L
from future_od.native.ops import ptr, stream


def _same_layout(a, b):
    sa = [s for s, n in zip(a.stride(), a.shape) if n > 1]
    sb = [s for s, n in zip(b.stride(), b.shape) if n > 1]
    return sa == sb


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self.max_norm = float(max_norm)
        self._chunk = L.LIB.fod_multi_chunk()
        self._tables = {}
        self._sq = None
        self._bias_dev = None        # device-resident bias corrections (captured steps, see future_od/graph.py)
        self._dev_step = self._dev_betas = None
        self.last_grad_norm = None

    def zero_grad(self, set_to_none=True):
        """Drops the gradients (set_to_none) and recycles the zero arena they were accumulated in."""
        for grp in self.param_groups:            # (torch's generic zero_grad costs 3x this loop in bookkeeping)
            for p in grp["params"]:
                if p.grad is not None:
                    p.grad = None
        from future_od.native import functional as Fn
        for grp in self.param_groups:
            if grp["params"]:
                Fn.ARENA.recycle(grp["params"][0].device)
                break

    def _state_for(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @staticmethod
    def _upload(host, dev):
        """Host table -> device without blocking the host on the stream (a pageable copy synchronises it, i.e.
        waits for the whole queued backward; the optimizer is the last thing queued in a step, so that wait
        would also keep the host from queueing the next forward while the GPU drains)."""
        pinned = host.pin_memory()
        return pinned.to(dev, non_blocking=True), pinned

    def _build_plan(self):
        """(Re)build the device tables for the current set of (param, grad, moment) pointers."""
        entries, params = [], []
        for grp in self.param_groups:
            for p in grp["params"]:
                g = p.grad
                if g is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda:
                    raise L.FodError("FusedAdamW handles float32 device parameters only")
                if g.dtype != torch.float32 or not _same_layout(g, p):
                    return None            # needs per-step copies: handled by the general path
                st = self._state_for(p)
                entries.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                p.numel(), grp["lr"], grp["weight_decay"]))
                params.append(p)
        if not entries:
            return None
        dev = params[0].device
        tab, ring = self._tables_for(entries, dev)
        return {"params": params, "gptrs": [e[1] for e in entries], "tab": tab, "ring": ring, "turn": 0,
                "spares": [ring[0].clone().pin_memory() for _ in range(4)],
                "lrs": [(g["lr"], g["weight_decay"]) for g in self.param_groups], "dev": dev}

    def _tables_for(self, entries, dev):
        ptrs = torch.tensor([e[:4] for e in entries], dtype=torch.int64)
        numel = torch.tensor([e[4] for e in entries], dtype=torch.int64)
        lr_wd = torch.tensor([e[5:7] for e in entries], dtype=torch.float32)
        bt, bc = [], []
        for t, e in enumerate(entries):
            for c in range((e[4] + self._chunk - 1) // self._chunk):
                bt.append(t)
                bc.append(c)
        up = [self._upload(x, dev) for x in (ptrs, numel, lr_wd, torch.tensor(bt, dtype=torch.int32),
                                             torch.tensor(bc, dtype=torch.int32))]
        # pinned copies of the pointer table, used in turn when only gradient addresses move between steps; each slot
        # carries an event recorded after its upload (see _refresh_grad_pointers), so a slot is never rewritten while
        # a copy that reads it is still queued
        ring = [up[0][1], up[0][1].clone().pin_memory(), up[0][1].clone().pin_memory()]
        self._keep = [u[1] for u in up]
        return tuple(u[0] for u in up), ring

    def _refresh_grad_pointers(self, plan):
        """Same parameters (with a gradient) as last step, in one pass over all parameters; returns False if the plan
        no longer applies."""
        gptrs = []
        planned, olds = plan["params"], plan["gptrs"]
        n, i = len(planned), 0
        for grp in self.param_groups:
            for p in grp["params"]:
                g = p.grad
                if g is None:
                    continue
                if i >= n or p is not planned[i]:
                    return False                     # a parameter gained (or another lost) its gradient
                gp = g.data_ptr()
                if gp != olds[i] and (g.dtype != torch.float32 or not _same_layout(g, p)):
                    return False
                gptrs.append(gp)
                i += 1
        if i != n:
            return False
        if gptrs != plan["gptrs"]:
            plan["turn"] = (plan["turn"] + 1) % len(plan["ring"])
            turn = plan["turn"]
            # the ring is safe by itself: a slot is rewritten only after the upload that last read it has executed (the
            # event is normally long complete; a loop without the matcher's one-step bound simply waits here)
            done = plan.setdefault("ring_events", [None] * len(plan["ring"]))
            capturing = torch.cuda.is_current_stream_capturing()
            if done[turn] is not None and not capturing:
                done[turn].synchronize()
            host = plan["ring"][turn]
            host[:, 1] = torch.tensor(gptrs, dtype=torch.int64)
            plan["tab"][0].copy_(host, non_blocking=True)
            if capturing:
                # the copy becomes a graph node that re-reads this slot at every replay: the slot is retired from the
                # ring's rotation for good (a captured step never comes back here; a later capture builds a new plan)
                # (its replacement was pinned when the plan was built: no host allocation may happen inside a capture)
                done[turn] = None
                spares = plan.setdefault("spares", [])
                if spares:
                    plan["ring"][turn] = spares.pop()
                    plan.setdefault("captured_slots", []).append(host)
                else:
                    # out of spare slots: the next eager step rebuilds the plan (and its ring); this one's tables are
                    # baked into the capture in progress and stay alive
                    self._retired_plans = getattr(self, "_retired_plans", []) + [plan]
                    self._plan = None
            else:
                ev = torch.cuda.Event()
                ev.record()
                done[turn] = ev
            plan["gptrs"] = gptrs
        return True

    def enable_device_step(self, device):
        """Keep the step count and the bias corrections in device memory, advanced by (capturable) device ops inside
        step(): a captured step must not bake `1 - beta**t` into a kernel argument."""
        betas = self.param_groups[0]["betas"]
        self._dev_step = torch.tensor([float(getattr(self, "_step_no", 0))], dtype=torch.float64, device=device)
        self._dev_betas = torch.tensor(list(betas), dtype=torch.float64, device=device)
        self._bias_dev = torch.ones(2, dtype=torch.float32, device=device)

    def disable_device_step(self):
        self._dev_step = self._dev_betas = self._bias_dev = None

    def sync_hyperparams(self, plan=None):
        """Write changed learning rates / weight decays into a plan's device table IN PLACE (a captured step reads the
        table of the plan that was current when it was CAPTURED -- `plan`, kept in the graph's record by
        future_od/graph.py -- which need not be the optimizer's current one any more: an eager step in between may have
        rebuilt it).  Default: the current plan.  Returns True if something was written."""
        if plan is None:
            plan = getattr(self, "_plan", None)
        if plan is None:
            return False
        lrs = [(g["lr"], g["weight_decay"]) for g in self.param_groups]
        if lrs == plan["lrs"]:
            return False
        group_of = {id(p): grp for grp in self.param_groups for p in grp["params"]}
        rows = [(group_of[id(p)]["lr"], group_of[id(p)]["weight_decay"]) for p in plan["params"]]
        assert len(rows) == plan["tab"][2].shape[0], "parameter set changed under a captured step"
        host = torch.tensor(rows, dtype=torch.float32).pin_memory()
        plan["tab"][2].copy_(host, non_blocking=True)
        plan["keep_lr"] = host                   # the copy reads it when it runs
        plan["lrs"] = lrs
        return True

    def _launch(self, tab, dev, betas, eps):
        ptrs, numel, lr_wd, bt, bc = tab
        if getattr(self, "_dev_step", None) is not None:
            self._dev_step.add_(1.0)
            self._bias_dev.copy_(1.0 - torch.pow(self._dev_betas, self._dev_step))
        if self._sq is None or self._sq.device != dev:
            self._sq = torch.zeros(1, dtype=torch.float32, device=dev)
        nblocks = bt.numel()
        sq = None
        if self.max_norm > 0:
            # fixed summation order: data-parallel replicas with equal gradients get bit-equal clip factors
            scr = getattr(self, "_sq_scratch", None)
            if scr is None or scr.device != dev or scr.numel() < nblocks + 1:
                if scr is not None:
                    self._retired = getattr(self, "_retired", []) + [scr]     # a captured step may still hold its address
                scr = self._sq_scratch = torch.zeros(nblocks + 1, dtype=torch.float32, device=dev)
            L.call("fod_multi_sqnorm_det", ptr(ptrs), ptr(numel), ptr(bt), ptr(bc), nblocks, ptr(self._sq), ptr(scr), stream())
            sq = self._sq
            self.last_grad_norm = self._sq      # device scalar holding the squared norm
        bc1 = 1.0 - betas[0] ** self._step_no
        bc2 = 1.0 - betas[1] ** self._step_no
        L.call("fod_multi_adamw", ptr(ptrs), ptr(numel), ptr(lr_wd), ptr(bt), ptr(bc), nblocks, betas[0], betas[1],
               eps, bc1, bc2, ptr(self._bias_dev), ptr(sq), self.max_norm, stream())
        # the kernel wrote the parameters through raw pointers (their `_version` did not move): every prepared
        # operand derived from a parameter (compute-dtype / transposed / BN-folded copies) is now out of date
        from future_od.native import functional as Fn
        Fn.PREP.mark_stale()

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        betas, eps = self.param_groups[0]["betas"], self.param_groups[0]["eps"]
        for grp in self.param_groups:
            assert grp["betas"] == betas and grp["eps"] == eps    # lr / weight decay are per tensor, these are shared
        self._step_no = getattr(self, "_step_no", 0) + 1
        # fast path: the same parameters as last step.  The gradient arena hands out the same slots every step, so
        # usually nothing moved; gradients that autograd allocates itself (sums of slices) get fresh addresses,
        # then only the pointer column is refreshed -- asynchronously, never a blocking copy
        plan = getattr(self, "_plan", None)
        if plan is not None:
            ok = (plan["lrs"] == [(g["lr"], g["weight_decay"]) for g in self.param_groups]
                  and self._refresh_grad_pointers(plan))
            if ok:
                self._launched_plan = plan
                self._launch(plan["tab"], plan["dev"], betas, eps)
                return None
        old_plan = getattr(self, "_plan", None)
        if old_plan is not None:
            # its device tables may be baked into a captured step (future_od/graph.py): kept alive, never freed
            self._retired_plans = getattr(self, "_retired_plans", []) + [old_plan]
        plan = self._build_plan()
        self._plan = plan
        self._launched_plan = plan               # None: the general path below (its tables are not reusable)
        if plan is not None:
            self._launch(plan["tab"], plan["dev"], betas, eps)
            return None
        # general path: some gradient needs a layout/dtype copy first
        entries, keepalive = [], []
        for grp in self.param_groups:
            for p in grp["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.dtype != torch.float32 or not _same_layout(g, p):
                    g2 = torch.empty_like(p, memory_format=torch.preserve_format)
                    g2.copy_(g)
                    g = g2
                    keepalive.append(g2)
                st = self._state_for(p)
                entries.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                p.numel(), grp["lr"], grp["weight_decay"]))
        if not entries:
            return None
        dev = self.param_groups[0]["params"][0].device
        tab, _ring = self._tables_for(entries, dev)
        self._launch(tab, dev, betas, eps)
        del keepalive
        return None

    def state_dict(self):
        """torch.optim.AdamW layout; the shared step counter is materialised per parameter."""
        step_no = float(getattr(self, "_step_no", 0))
        for st in self.state.values():
            if "exp_avg" in st:
                st["step"] = torch.tensor(step_no)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = [float(st["step"]) for st in self.state.values() if "step" in st]
        self._step_no = int(max(steps)) if steps else 0
        self._state_token = getattr(self, "_state_token", 0) + 1      # the moment tensors are new objects: captured steps
        if getattr(self, "_dev_step", None) is not None:              # that baked the old ones in must be re-captured
            self._dev_step.fill_(float(self._step_no))
        if getattr(self, "_plan", None) is not None:
            self._retired_plans = getattr(self, "_retired_plans", []) + [self._plan]
        self._plan = None
