"""Minimal gym-style containers (the `gym` package is not a dependency of the simulator): an Env base class and
shape/dtype spaces with the attribute names RL code reads (low, high, shape, dtype, spaces).  If a real `gym` /
`gymnasium` is installed, `gym_collision_avoidance_amd.register_with_gym()` (called on import of the package root)
exposes the env under the reference's id `CollisionAvoidance-v0`."""
import numpy as np


class Env(object):
    metadata = {}

    def reset(self):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError


class Space(object):
    pass


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high = np.asarray(low), np.asarray(high)
        self.shape = tuple(shape) if shape is not None else self.low.shape
        self.dtype = np.dtype(dtype)

    def __repr__(self):
        return "Box(%s, %s)" % (self.shape, self.dtype)


class Dict(Space):
    def __init__(self, spaces=None):
        self.spaces = dict(spaces or {})

    def __getitem__(self, k):
        return self.spaces[k]

    def __setitem__(self, k, v):
        self.spaces[k] = v
