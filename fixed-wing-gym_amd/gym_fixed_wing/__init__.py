"""gym_fixed_wing -- MI355X-native vectorised fixed-wing gym (drop-in package name of eivindeb/fixed-wing-gym).

    from gym_fixed_wing.fixed_wing import FixedWingAircraft       # reference-compatible single env
    from gym_fixed_wing.vec_env import FixedWingVecEnv            # N envs, one fused HIP launch per step

The compute path is libfwgym.so (hand-written HIP for gfx950, see csrc/); there is no CPU fallback.
"""
__all__ = ["FixedWingAircraft", "FixedWingVecEnv"]


def __getattr__(name):
    if name == "FixedWingAircraft":
        from .fixed_wing import FixedWingAircraft
        return FixedWingAircraft
    if name == "FixedWingVecEnv":
        from .vec_env import FixedWingVecEnv
        return FixedWingVecEnv
    raise AttributeError(name)
