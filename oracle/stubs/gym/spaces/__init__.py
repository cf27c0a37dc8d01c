"""Stand-in for gym.spaces: shape/dtype containers only."""
import numpy as np


class Space(object):
    pass


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low = np.asarray(low)
        self.high = np.asarray(high)
        self.shape = shape if shape is not None else self.low.shape
        self.dtype = dtype


class Discrete(Space):
    def __init__(self, n, dtype=None):
        self.n = n


class Dict(Space):
    def __init__(self, spaces=None):
        self.spaces = dict(spaces or {})

    def __getitem__(self, k):
        return self.spaces[k]

    def __setitem__(self, k, v):
        self.spaces[k] = v
