import numpy as np

from gym_collision_avoidance_amd import _native as nat
from .ExternalPolicy import ExternalPolicy


class LearningPolicy(ExternalPolicy):
    """External actions in [0,1]^2 scaled by the agent (reference policies/LearningPolicy.py:29-33):
    speed = pref_speed * a[0], delta_heading = max_heading_change * (2 a[1] - 1)."""
    kernel_id = nat.POL_LEARNING

    def __init__(self):
        ExternalPolicy.__init__(self, str="learning")
        self.is_still_learning = True
        self.ppo_or_learning_policy = True

    def external_action_to_action(self, agent, external_action):
        return np.array([agent.pref_speed * external_action[0],
                         agent.max_heading_change * (2. * external_action[1] - 1.)])
