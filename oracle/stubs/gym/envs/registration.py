"""Stand-in for gym.envs.registration: id -> entry_point registry + make()."""
import importlib

_REGISTRY = {}


def register(id, entry_point, **kwargs):
    _REGISTRY[id] = entry_point


def make(id, **kwargs):
    mod, cls = _REGISTRY[id].split(":")
    return getattr(importlib.import_module(mod), cls)(**kwargs)
