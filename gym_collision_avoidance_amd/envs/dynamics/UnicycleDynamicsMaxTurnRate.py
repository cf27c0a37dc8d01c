from gym_collision_avoidance_amd import _native as nat
from .Dynamics import Dynamics


class UnicycleDynamicsMaxTurnRate(Dynamics):
    """Unicycle with the turn rate clipped to +-3 rad/s (reference dynamics/UnicycleDynamicsMaxTurnRate.py:17-43)."""
    kernel_id = nat.DYN_MAX_TURN_RATE

    def __init__(self, agent):
        Dynamics.__init__(self, agent)
        self.max_turn_rate = 3.0

    def step(self, action, dt):
        """Host-callable like the reference's (UnicycleDynamicsMaxTurnRate.py:17-43; no turning_dir update there)."""
        import numpy as np
        from gym_collision_avoidance_amd.envs.agent import wrap
        rate = np.clip(action[1] / dt, -self.max_turn_rate, self.max_turn_rate)
        self._host_unicycle(action[0], wrap(rate * dt + self.agent.heading_global_frame), dt, False)
