import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    def sample(self):
        return np.zeros(self.shape, self.dtype)


class Dict(dict):
    def __init__(self, spaces=None, **kw):
        super().__init__(spaces or {}, **kw)
        self.spaces = self
