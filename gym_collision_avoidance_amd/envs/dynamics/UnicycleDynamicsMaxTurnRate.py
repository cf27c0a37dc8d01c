from gym_collision_avoidance_amd import _native as nat
from .Dynamics import Dynamics


class UnicycleDynamicsMaxTurnRate(Dynamics):
    """Unicycle with the turn rate clipped to +-3 rad/s (reference dynamics/UnicycleDynamicsMaxTurnRate.py:17-43)."""
    kernel_id = nat.DYN_MAX_TURN_RATE

    def __init__(self, agent):
        Dynamics.__init__(self, agent)
        self.max_turn_rate = 3.0
