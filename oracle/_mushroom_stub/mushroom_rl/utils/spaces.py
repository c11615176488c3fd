import numpy as np


class Box:
    def __init__(self, low, high, shape=None):
        self._low = np.array(low, dtype=float)
        self._high = np.array(high, dtype=float)

    @property
    def low(self):
        return self._low

    @property
    def high(self):
        return self._high

    @property
    def shape(self):
        return self._low.shape
