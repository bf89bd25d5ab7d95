"""Span accounting for the driver's progress line (detect-time / misc-time, lib/test.py:253-261).

The driver needs three things per span kind: how many spans were closed, their summed
wall time, and the mean (``average_time`` -- the attribute name the progress line reads).
The fused path keeps two images in flight, so spans may overlap: every ``tic`` pushes an
opening stamp and every ``toc`` closes the OLDEST open one (FIFO, the order images are
collected in)."""
import time
from collections import deque


class Timer(object):
    __slots__ = ("_open", "total_time", "calls", "diff")

    def __init__(self):
        self._open = deque()
        self.total_time = 0.0
        self.calls = 0
        self.diff = 0.0

    def tic(self):
        self._open.append(time.perf_counter())

    def toc(self, average=True):
        t0 = self._open.popleft() if self._open else time.perf_counter()
        self.diff = time.perf_counter() - t0
        self.total_time += self.diff
        self.calls += 1
        return self.average_time if average else self.diff

    @property
    def average_time(self):
        return self.total_time / self.calls if self.calls else 0.0
