"""Build-container-only harness: imports the VERBATIM reference gym module from /root/reference over stub ``gym`` and
``pyfly`` packages.  Nothing from the reference is copied; when /root/reference is absent (GPU box)
``load_reference()`` returns None and callers skip.  The stub ``pyfly`` is the oracle's restated simulator."""
import importlib.util
import os
import sys
import types

import numpy as np

REFERENCE_FILE = "/root/reference/gym_fixed_wing/fixed_wing.py"
REFERENCE_DIR = "/root/reference/gym_fixed_wing"

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _install_stubs():
    if "gym" not in sys.modules:
        gym = types.ModuleType("gym")

        class Env(object):
            pass

        class GoalEnv(Env):
            pass

        class Box(object):
            def __init__(self, low, high, shape=None, dtype=np.float32):
                if shape is None:
                    self.low, self.high = np.asarray(low), np.asarray(high)
                    self.shape = self.low.shape
                else:
                    self.low, self.high = np.full(shape, low), np.full(shape, high)
                    self.shape = tuple(shape)
                self.dtype = dtype

        class Dict(dict):
            def __init__(self, spaces):
                super().__init__(spaces)
                self.spaces = spaces

        spaces = types.ModuleType("gym.spaces")
        spaces.Box, spaces.Dict = Box, Dict
        utils = types.ModuleType("gym.utils")
        seeding = types.ModuleType("gym.utils.seeding")

        def np_random(seed=None):
            seed = 0 if seed is None else int(seed)
            return np.random.RandomState(seed % (2 ** 32)), seed

        seeding.np_random = np_random
        utils.seeding = seeding
        gym.Env, gym.GoalEnv, gym.spaces, gym.utils = Env, GoalEnv, spaces, utils
        sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.utils": utils, "gym.utils.seeding": seeding})
    if "pyfly" not in sys.modules:
        from oracle import pyfly_restated
        pyfly = types.ModuleType("pyfly")
        pyfly_pyfly = types.ModuleType("pyfly.pyfly")
        pyfly_pyfly.PyFly = pyfly_restated.PyFly
        pid = types.ModuleType("pyfly.pid_controller")
        pid.PIDController = pyfly_restated.PIDController
        pyfly.pyfly, pyfly.pid_controller = pyfly_pyfly, pid
        sys.modules.update({"pyfly": pyfly, "pyfly.pyfly": pyfly_pyfly, "pyfly.pid_controller": pid})


_cached = None


def load_reference():
    """Returns the verbatim reference module (class FixedWingAircraft) or None when the reference is not mounted."""
    global _cached
    if _cached is not None:
        return _cached
    if not os.path.exists(REFERENCE_FILE):
        return None
    import matplotlib
    matplotlib.use("Agg")
    _install_stubs()
    spec = importlib.util.spec_from_file_location("_reference_fixed_wing", REFERENCE_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _cached = mod
    return mod
