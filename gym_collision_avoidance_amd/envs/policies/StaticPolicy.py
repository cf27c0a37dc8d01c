import numpy as np

from gym_collision_avoidance_amd import _native as nat
from .InternalPolicy import InternalPolicy


class StaticPolicy(InternalPolicy):
    """Never moves; its goal becomes its position (reference policies/StaticPolicy.py:21-23)."""
    kernel_id = nat.POL_STATIC

    def __init__(self):
        InternalPolicy.__init__(self, str="Static")

    def find_next_action(self, obs, agents, i):
        return np.array([0.0, 0.0])
