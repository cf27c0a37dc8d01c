from gym_collision_avoidance_amd import _native as nat
from .Dynamics import Dynamics


class ExternalDynamics(Dynamics):
    """State is set from outside (Agent.set_state); the step leaves pos / heading alone (reference
    dynamics/ExternalDynamics.py)."""
    kernel_id = nat.DYN_EXTERNAL

    def step(self, action, dt):
        """Nothing to integrate: the state comes from outside (dynamics/ExternalDynamics.py)"""
        return
