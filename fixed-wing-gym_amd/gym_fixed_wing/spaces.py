"""Minimal stand-ins for gym.spaces.Box / gym.Env used when `gym` is not installed (the reference only touches
.shape/.low/.high of its spaces, fixed_wing.py:176-191)."""
import numpy as np

try:  # pragma: no cover - depends on the environment
    import gym as _gym
    Env = _gym.Env
    GoalEnv = getattr(_gym, "GoalEnv", _gym.Env)
    Box = _gym.spaces.Box
    DictSpace = _gym.spaces.Dict
    HAVE_GYM = True
except Exception:  # gym is optional
    HAVE_GYM = False

    class DictSpace(dict):
        """gym.spaces.Dict stand-in: a dict of spaces (fixed_wing.py:1183-1187 only builds it)."""

        def __init__(self, spaces):
            super().__init__(spaces)
            self.spaces = spaces

    class Env(object):
        metadata = {}
        reward_range = (-float("inf"), float("inf"))

    class Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:
                low, high = np.asarray(low), np.asarray(high)
                shape = low.shape
            else:
                low, high = np.full(shape, low), np.full(shape, high)
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

        def sample(self):
            lo = np.clip(self.low, -1e6, 1e6)
            hi = np.clip(self.high, -1e6, 1e6)
            return np.random.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return "Box{}".format(self.shape)

    class GoalEnv(Env):
        pass
