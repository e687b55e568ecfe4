import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")]
import torch
import configs
from gym_fixed_wing.vec_env import FixedWingVecEnv
from gym_fixed_wing.actor import DeviceActor
from gym_fixed_wing.rollout import FusedRollout, MlpPolicy
cfg = configs.reference_like("examples")
n = 65536
vec = FixedWingVecEnv(cfg, num_envs=n, device=0, derived_views=False)
vec.reset()
actor = DeviceActor.for_env(vec, seed=1)
actor.load_policy(MlpPolicy(12))
ro = FusedRollout(vec, actor, 128, graph=False, fused="auto")
for _ in range(3): ro.run()
torch.cuda.synchronize()
