"""Minimal stand-in for the `gym` package (absent in this image): only what the reference
imports at module scope.  Test infrastructure for oracle/gen_golden.py -- no simulator logic."""
from . import spaces, envs  # noqa: F401
from .envs.registration import register, make  # noqa: F401


class Env(object):
    metadata = {}

    def reset(self):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.observation_space = getattr(env, "observation_space", None)
        self.action_space = getattr(env, "action_space", None)

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, action):
        return self.env.step(action)


class ObservationWrapper(Wrapper):
    def reset(self, **kw):
        out = self.env.reset(**kw)
        if isinstance(out, tuple):
            return (self.observation(out[0]),) + tuple(out[1:])
        return self.observation(out)

    def step(self, action):
        out = self.env.step(action)
        return (self.observation(out[0]),) + tuple(out[1:])

    def observation(self, observation):
        raise NotImplementedError


class _Logger(object):
    def set_level(self, level):
        pass


logger = _Logger()
