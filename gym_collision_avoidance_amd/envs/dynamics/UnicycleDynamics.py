from gym_collision_avoidance_amd import _native as nat
from .Dynamics import Dynamics


class UnicycleDynamics(Dynamics):
    """heading' = wrap(heading + a[1]); pos += a[0] * (cos, sin)(heading') * dt (reference
    dynamics/UnicycleDynamics.py:14-47)."""
    kernel_id = nat.DYN_UNICYCLE
