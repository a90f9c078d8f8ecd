"""Wall-clock Timer with the attributes the reference's callers read (`.diff`, `.avg`;
scripts/test_kitti.py:83,92-93).  Semantics follow util/timer.py:12-54 (time.time, no device sync)."""
import time


class Timer:
    def __init__(self):
        self.reset()

    def reset(self):
        self.diff = 0.0
        self.sum = 0.0
        self.count = 0
        self.avg = 0.0
        self.val = 0.0

    def tic(self):
        self.start_time = time.time()

    def toc(self, average=True):
        self.diff = time.time() - self.start_time
        self.val = self.diff
        self.sum += self.diff
        self.count += 1
        self.avg = self.sum / self.count
        return self.avg if average else self.diff
