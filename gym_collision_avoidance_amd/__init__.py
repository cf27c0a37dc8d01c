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
        except Exception:  # noqa: BLE001 -- absent, or an installed gym that fails to import (numpy 2 vs old gym): never
            continue       # a reason for `import gym_collision_avoidance_amd` to fail
        registry = getattr(mod, "registry", None)
        known = False
        try:  # gym < 0.26: EnvRegistry with .env_specs; newer gym / gymnasium: a plain dict
            known = ENV_ID in (registry.env_specs if hasattr(registry, "env_specs") else registry)
        except TypeError:
            known = False
        try:
            if not known:
                # (the env derives from the package's own minimal Env / spaces: gym >= 0.26 and gymnasium check the class
                # and the spaces in make(), so their checker is switched off for this id)
                try:
                    mod.register(id=ENV_ID, entry_point=ENTRY_POINT, disable_env_checker=True)
                except TypeError:
                    mod.register(id=ENV_ID, entry_point=ENTRY_POINT)
        except Exception:  # noqa: BLE001
            continue
        done.append(name)
    return done


register_with_gym()
