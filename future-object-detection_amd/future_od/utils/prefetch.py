"""Host -> device staging of loader batches, one batch ahead.

The reference's loop moves each batch with blocking `.to(device)` calls after the previous step has been queued
(reference future_od/trainer.py:173-176, runs/_loader.py:107-115: no pinned memory, no prefetch).  Here the NEXT
batch is copied from pinned host memory on a side stream while the current step runs, so its PCIe time (3.3 ms per
900x1600 float clip of six frames, a quarter of that for uint8 frames) is off the step:

    for data in DevicePrefetcher(loader, device):
        ...                      # `data` is on the device; its host annotation copies ride along

The batch dict keeps `_host_annotations` (see `recursive_to`), which the model uses to build the targets without
reading anything back from the device.
"""
import torch

from future_od.utils.recursive_functions import _walk, recursive_to, HOST_ANNOTATIONS, _ANNOTATION_KEYS


class DevicePrefetcher:
    def __init__(self, loader, device):
        self.loader = loader
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def __len__(self):
        return len(self.loader)

    @property
    def batch_size(self):
        return getattr(self.loader, "batch_size", None)

    def _stage(self, batch):
        if self.stream is None:
            return recursive_to(batch, self.device), None
        host = {}
        if isinstance(batch, dict) and all(k in batch for k in _ANNOTATION_KEYS):
            host = {k: batch[k] for k in _ANNOTATION_KEYS
                    if isinstance(batch[k], torch.Tensor) and batch[k].device.type == "cpu"}
        with torch.cuda.stream(self.stream):
            moved = _walk(batch, lambda t: (t if t.is_pinned() else t.pin_memory()).to(self.device, non_blocking=True)
                          if t.device.type == "cpu" else t)
            ready = self.stream.record_event()
        if len(host) == len(_ANNOTATION_KEYS) and HOST_ANNOTATIONS not in moved:
            moved[HOST_ANNOTATIONS] = host
        return moved, ready

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            batch, ready = nxt
            if ready is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ready)
                _walk(batch, lambda t: t.record_stream(cur) if t.is_cuda else None)
            try:
                nxt = self._stage(next(it))      # queued now, travels while the caller runs the step on `batch`
            except StopIteration:
                nxt = None
            yield batch
