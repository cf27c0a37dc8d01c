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


def install_as(alias="gym_collision_avoidance", provide_gym=False):
    """Opt-in import alias: after `gym_collision_avoidance_amd.install_as()` caller code written against the reference --

        from gym_collision_avoidance.envs import Config, test_cases as tc
        from gym_collision_avoidance.experiments.src.env_utils import run_episode, create_env

    -- imports THIS package under the reference's name (gym_collision_avoidance/__init__.py:6-9 and the module paths of
    its tree; `experiments.src.*` maps to `experiments.*`): every aliased name is the same module object as its
    `gym_collision_avoidance_amd` original, so the Config singleton and the registries are shared.  Refuses to shadow an
    importable package of that name (the reference itself) unless it is this alias already.

    provide_gym: the reference's callers also `import gym` (`gym.make("CollisionAvoidance-v0")`, `gym.logger.set_level`);
    where neither gym nor gymnasium is installed -- this image -- True installs a minimal stand-in module `gym` (make /
    logger / Env / spaces from envs/spaces.py) that knows this one env id.  Never replaces a real gym."""
    import importlib
    import importlib.abc
    import importlib.util
    import sys
    real = __name__
    for f in sys.meta_path:
        if getattr(f, "_cagpu_alias", None) == alias:
            break
    else:
        if alias in sys.modules or importlib.util.find_spec(alias) is not None:
            raise ImportError("install_as(%r): a package of that name is already importable; not shadowing it" % alias)

        def target(fullname):
            rest = fullname[len(alias):]
            if rest.startswith(".experiments.src"):       # the reference keeps its experiment code one level deeper
                rest = ".experiments" + rest[len(".experiments.src"):]
            return real + rest

        class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
            _cagpu_alias = alias

            def find_spec(self, fullname, path=None, target_=None):
                if fullname == alias or fullname.startswith(alias + "."):
                    try:
                        if importlib.util.find_spec(target(fullname)) is None:
                            return None
                    except ModuleNotFoundError:
                        return None
                    return importlib.util.spec_from_loader(fullname, self)
                return None

            def create_module(self, spec):
                return importlib.import_module(target(spec.name))   # the SAME module object under a second name

            def exec_module(self, module):
                pass

        sys.meta_path.insert(0, _AliasFinder())
    if provide_gym and "gym" not in sys.modules and importlib.util.find_spec("gym") is None:
        import types

        class _Gym(types.ModuleType):
            def __getattr__(self, name):   # (lazily: importing envs.* instantiates the Config singleton, which the caller
                if name in ("spaces", "Env"):   # selects through GYM_CONFIG_CLASS before ITS first import of the package)
                    sp = importlib.import_module(real + ".envs.spaces")
                    return sp if name == "spaces" else sp.Env
                raise AttributeError(name)

        g = _Gym("gym")
        g.__doc__ = "minimal stand-in installed by gym_collision_avoidance_amd.install_as(provide_gym=True): no gym in this image"

        def make(env_id, **kwargs):
            if env_id != ENV_ID:
                raise KeyError("the gym stand-in of gym_collision_avoidance_amd only knows %r" % ENV_ID)
            mod, cls = ENTRY_POINT.split(":")
            return getattr(importlib.import_module(mod), cls)(**kwargs)
        g.make = make
        g.logger = types.SimpleNamespace(set_level=lambda level: None)
        sys.modules["gym"] = g
    return alias


register_with_gym()
