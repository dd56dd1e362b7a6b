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
        self.last_grad_norm = None

    def zero_grad(self, set_to_none=True):
        """Drops the gradients (set_to_none) and recycles the zero arena they were accumulated in."""
        super().zero_grad(set_to_none=True)
        from future_od.native import functional as Fn
        for grp in self.param_groups:
            if grp["params"]:
                Fn.ARENA.recycle(grp["params"][0].device)
                break

    def _state_for(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        # all groups share betas/eps (as the reference's two groups do); lr / weight decay are per tensor
        entries, keepalive = [], []
        betas, eps = self.param_groups[0]["betas"], self.param_groups[0]["eps"]
        step_no = None
        for grp in self.param_groups:
            assert grp["betas"] == betas and grp["eps"] == eps
            for p in grp["params"]:
                g = p.grad
                if g is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda:
                    raise L.FodError("FusedAdamW handles float32 device parameters only")
                if g.dtype != torch.float32 or not _same_layout(g, p):
                    g2 = torch.empty_like(p, memory_format=torch.preserve_format)
                    g2.copy_(g)
                    g = g2
                    keepalive.append(g2)
                st = self._state_for(p)
                st["step"] += 1
                step_no = float(st["step"]) if step_no is None else step_no
                entries.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                p.numel(), grp["lr"], grp["weight_decay"]))
        if not entries:
            return None
        dev = self.param_groups[0]["params"][0].device
        key = tuple(entries)
        tab = self._tables.get(key)
        if tab is None:
            if len(self._tables) > 8:
                self._tables.clear()
            ptrs = torch.tensor([e[:4] for e in entries], dtype=torch.int64)
            numel = torch.tensor([e[4] for e in entries], dtype=torch.int64)
            lr_wd = torch.tensor([e[5:7] for e in entries], dtype=torch.float32)
            bt, bc = [], []
            for t, e in enumerate(entries):
                for c in range((e[4] + self._chunk - 1) // self._chunk):
                    bt.append(t)
                    bc.append(c)
            tab = tuple(x.to(dev) for x in (ptrs, numel, lr_wd, torch.tensor(bt, dtype=torch.int32),
                                            torch.tensor(bc, dtype=torch.int32)))
            self._tables[key] = tab
        ptrs, numel, lr_wd, bt, bc = tab
        if self._sq is None or self._sq.device != dev:
            self._sq = torch.zeros(1, dtype=torch.float32, device=dev)
        nblocks = bt.numel()
        sq = None
        if self.max_norm > 0:
            self._sq.zero_()
            L.call("fod_multi_sqnorm_acc", ptr(ptrs), ptr(numel), ptr(bt), ptr(bc), nblocks, ptr(self._sq), stream())
            sq = self._sq
            self.last_grad_norm = self._sq      # device scalar holding the squared norm
        bc1 = 1.0 - betas[0] ** step_no
        bc2 = 1.0 - betas[1] ** step_no
        L.call("fod_multi_adamw", ptr(ptrs), ptr(numel), ptr(lr_wd), ptr(bt), ptr(bc), nblocks, betas[0], betas[1],
               eps, bc1, bc2, ptr(sq), self.max_norm, stream())
        del keepalive
        return None
