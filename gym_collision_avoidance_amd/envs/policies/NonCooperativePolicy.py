import numpy as np

from gym_collision_avoidance_amd import _native as nat
from .InternalPolicy import InternalPolicy


class NonCooperativePolicy(InternalPolicy):
    """Drive at pref_speed straight toward the goal (reference policies/NonCooperativePolicy.py:21)."""
    kernel_id = nat.POL_NONCOOP

    def __init__(self):
        InternalPolicy.__init__(self, str="NonCooperativePolicy")

    def find_next_action(self, obs, agents, i):  # host restatement, used only when stepping outside the kernel
        return np.array([agents[i].pref_speed, -agents[i].heading_ego_frame])
