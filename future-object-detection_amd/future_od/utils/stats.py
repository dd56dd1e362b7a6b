"""Epoch statistics for the Trainer's log lines.

`AverageMeter` keeps the sample-weighted mean of a scalar over the current epoch and the list of past epoch means.
Public surface as the Trainer uses it: `.update(value, n)`, `.avg`, `.val`, `.count`, `.new_epoch()`, `.history`,
`.reset()`, `.clear()`; the mean after updates of zero total weight is the string "nan".

Checkpoints pickle these objects (`Trainer.save_checkpoint` stores the stats dict, as the reference's
future_od/trainer.py:282-300 does), so the pickled state keeps the reference's attribute names -- history, val, sum,
avg, count -- in both directions: a checkpoint written by the reference loads here and vice versa."""


class AverageMeter:
    def __init__(self):
        self.history = []            # one mean per finished epoch
        self.reset()

    # ---- current epoch
    def reset(self):
        self.val, self.sum, self.count, self._touched = 0, 0, 0, False

    def update(self, val, n=1):
        self.val = val               # the most recent sample
        self.sum = self.sum + val * n
        self.count = self.count + n
        self._touched = True

    @property
    def avg(self):
        if self.count > 0:
            return self.sum / self.count
        return "nan" if self._touched else 0      # updated with zero weight only: undefined mean

    # ---- epoch boundary
    def new_epoch(self):
        self.history.append(self.avg)
        self.reset()

    def clear(self):
        del self.history[:]
        self.reset()

    # ---- pickling (checkpoints)
    def __getstate__(self):
        return {"history": list(self.history), "val": self.val, "sum": self.sum, "avg": self.avg, "count": self.count}

    def __setstate__(self, state):
        self.history = list(state.get("history", []))
        self.val, self.sum, self.count = state.get("val", 0), state.get("sum", 0), state.get("count", 0)
        self._touched = self.count > 0 or state.get("avg", 0) == "nan"
