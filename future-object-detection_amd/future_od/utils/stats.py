class AverageMeter:
    """Running mean of a scalar within an epoch plus the history of epoch means (reference utils/stats.py)."""

    def __init__(self):
        self.history = []
        self.reset()

    def reset(self):
        self.val = self.sum = self.avg = 0
        self.count = 0

    def clear(self):
        self.history = []
        self.reset()

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count if self.count > 0 else "nan"

    def new_epoch(self):
        self.history.append(self.avg)
        self.reset()
