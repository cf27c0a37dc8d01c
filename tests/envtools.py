"""Re-import the env package under a chosen Config class (the Config object is an import-time singleton, like the
reference's; its own test file purges sys.modules the same way, tests/test_collision_avoidance.py:10-19)."""
import importlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def fresh(config_class, config_path=os.path.join(HERE, "env_configs.py")):
    os.environ["GYM_CONFIG_PATH"] = config_path
    os.environ["GYM_CONFIG_CLASS"] = config_class
    for m in [m for m in sys.modules if m.startswith("gym_collision_avoidance_amd.envs")
              or m.startswith("gym_collision_avoidance_amd.experiments")]:
        del sys.modules[m]
    envs = importlib.import_module("gym_collision_avoidance_amd.envs")
    tc = importlib.import_module("gym_collision_avoidance_amd.envs.test_cases")
    cae = importlib.import_module("gym_collision_avoidance_amd.envs.collision_avoidance_env")
    return envs.Config, tc, cae.CollisionAvoidanceEnv


def default():
    os.environ.pop("GYM_CONFIG_PATH", None)
    os.environ.pop("GYM_CONFIG_CLASS", None)
    for m in [m for m in sys.modules if m.startswith("gym_collision_avoidance_amd.envs")]:
        del sys.modules[m]
