"""Config singleton, selected exactly like the reference does it (gym_collision_avoidance/envs/__init__.py:4-18):
env var GYM_CONFIG_PATH (default: this package's config.py) + GYM_CONFIG_CLASS (default "Config"), instantiated once
at import time; every module then does `from gym_collision_avoidance_amd.envs import Config`."""
import importlib.util
import os

_here = os.path.dirname(os.path.realpath(__file__))
gym_config_path = os.environ.get("GYM_CONFIG_PATH", os.path.join(_here, "config.py"))
gym_config_class = os.environ.get("GYM_CONFIG_CLASS", "Config")

_spec = importlib.util.spec_from_file_location(gym_config_class, gym_config_path)
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
_cls = getattr(_mod, gym_config_class, None)
assert callable(_cls), "config class %r not found in %s" % (gym_config_class, gym_config_path)
Config = _cls()
