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

Restrictions (checked): one device, no data-parallel gradient reducer (its side-stream collectives are not captured),
dropout inactive (eval-mode math or p = 0: a captured kernel would replay the same mask), static shapes -- a batch
with other shapes is re-captured.
"""
import torch

from future_od.native import functional as Fn


class GraphedStep:
    """step = GraphedStep(model, optimizer);  post, loss, stats, od = step(data)

    The first call with a given batch signature runs `warmup` eager steps (they are real optimizer steps: nothing is
    rolled back) and captures one more; later calls copy the batch into the captured input buffers and replay.
    Returned tensors are the graph's static outputs: valid until the next call."""

    def __init__(self, model, optimizer, warmup=2):
        self.model, self.opt, self.warmup = model, optimizer, max(int(warmup), 2)
        self._graphs = {}
        self.replays = 0

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _signature(data):
        return tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in data.items() if isinstance(v, torch.Tensor)))

    def _check(self, data):
        core = getattr(self.model, "module", self.model)
        if core is not self.model or Fn.GRAD_SYNC is not None:
            raise RuntimeError("GraphedStep: data-parallel wrappers are not captured; use the eager step")
        if self.model.training:
            drops = [m.p for m in self.model.modules() if isinstance(m, torch.nn.Dropout)]
            rates = [float(getattr(m, "droprate", 0.0) or 0.0) for m in self.model.modules()]
            if any(p > 0 for p in drops + rates):
                raise RuntimeError("GraphedStep: dropout is active (train mode, p > 0): a captured step would replay "
                                   "the same masks; use model.eval() (autograd stays on) or p = 0")
        for k, v in data.items():
            if isinstance(v, torch.Tensor) and not v.is_cuda:
                raise RuntimeError(f"GraphedStep: batch entry {k!r} is not on the device")

    def _eager(self, data):
        self.opt.zero_grad()
        post, _state, loss, stats, od = self.model(data=data, distributed=False)
        loss.backward()
        self.opt.step()
        return post, loss, stats, od

    def _capture(self, data):
        self._check(data)
        dev = next(v for v in data.values() if isinstance(v, torch.Tensor)).device
        static = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in data.items()
                  if k != "_host_annotations"}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(self.warmup):          # populates every cache (prepared-weight tables, optimizer plan, ...)
                self._eager(static)
            if getattr(self.opt, "_dev_step", None) is None:
                self.opt.enable_device_step(dev)
                self._eager(static)               # one eager step through the device-side bias corrections
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs = self._eager(static)
        torch.cuda.synchronize(dev)
        # the capture itself executed nothing: parameters, moments and the device step count are those after the
        # warm-up steps; python's step counter ran one ahead during capture
        self.opt._step_no -= 1
        return {"graph": graph, "static": static, "outs": outs}

    # ------------------------------------------------------------------------------------------
    def __call__(self, data):
        sig = self._signature(data)
        g = self._graphs.get(sig)
        if g is None:
            g = self._graphs[sig] = self._capture(data)
        for k, v in g["static"].items():
            if isinstance(v, torch.Tensor):
                src = data[k]
                if src.data_ptr() != v.data_ptr():
                    v.copy_(src, non_blocking=True)
        self.opt.sync_hyperparams()
        g["graph"].replay()
        self.opt._step_no += 1
        self.replays += 1
        Fn.PREP.mark_stale()          # the replay changed the parameters without bumping their python-side versions
        return g["outs"]
