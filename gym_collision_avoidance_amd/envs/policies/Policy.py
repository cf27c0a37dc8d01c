import numpy as np

from gym_collision_avoidance_amd import _native as nat  # noqa: F401  (ids used by subclasses)


class Policy(object):
    """Base class (reference policies/Policy.py:11-14): `str`, `is_still_learning`, `is_external`.
    `kernel_id` (None here) selects the in-kernel implementation; see policies/__init__.py."""
    kernel_id = None

    def __init__(self, str="NoPolicy"):
        self.str = str
        self.is_still_learning = False
        self.is_external = False

    def near_goal_smoother(self, dist_to_goal, pref_speed, heading, raw_action):
        # reference Policy.py:16-35 always returns one of these two (its ramp-down result is overwritten)
        return np.array([0., 0.]) if dist_to_goal < 0.3 else raw_action
