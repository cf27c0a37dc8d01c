"""gym_collision_avoidance_amd -- MI355X-native batched collision-avoidance simulator.

Drop-in for the hot path of mit-acl/gym-collision-avoidance (`CollisionAvoidanceEnv.step` and the
Agent / Policy / Dynamics / Sensor plugin surface), with the per-step work in hand-written HIP kernels
(csrc/cagpu.hip) behind the C ABI of include/cagpu.h.  (The directory is spelled with underscores because a
Python package name cannot contain '-'.)
"""
__version__ = "0.1.0"

ENV_ID = "CollisionAvoidance-v0"
ENTRY_POINT = "gym_collision_avoidance_amd.envs.collision_avoidance_env:CollisionAvoidanceEnv"


def register_with_gym():
    """Register the env under the reference's id (gym_collision_avoidance/__init__.py:6-9) with whichever of `gym` /
    `gymnasium` is importable, so that `gym.make("CollisionAvoidance-v0")` returns this implementation.  The simulator
    itself does not depend on either package (envs/spaces.py holds the minimal containers); returns the list of
    module names the id was registered with (empty when neither is installed)."""
    done = []
    for name in ("gym", "gymnasium"):
        try:
            mod = __import__(name + ".envs.registration", fromlist=["register"])
        except ImportError:
            continue
        registry = getattr(mod, "registry", None)
        known = False
        try:  # gym < 0.26: EnvRegistry with .env_specs; newer gym / gymnasium: a plain dict
            known = ENV_ID in (registry.env_specs if hasattr(registry, "env_specs") else registry)
        except TypeError:
            known = False
        if not known:
            mod.register(id=ENV_ID, entry_point=ENTRY_POINT)
        done.append(name)
    return done


register_with_gym()
