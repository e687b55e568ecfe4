#!/usr/bin/env python3
"""Times the fused C5 rollout (env step + rollout head, hipGraph replay) for every libfwgym variant under
gym_fixed_wing/_abl/ (built by tools/ablate.py NAME="-DFLAG ...") plus the product library."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fixed-wing-gym_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import configs
from gym_fixed_wing import _native as nat
from gym_fixed_wing.vec_env import FixedWingVecEnv
from gym_fixed_wing.actor import DeviceActor
from gym_fixed_wing.rollout import FusedRollout, MlpPolicy
OUT = os.path.join(PKG, "gym_fixed_wing", "_abl")
libs = {"product": None}
if os.path.isdir(OUT):
    libs.update({f[9:-3]: os.path.join(OUT, f) for f in sorted(os.listdir(OUT)) if f.endswith(".so")})
cfg = configs.reference_like("examples")
n = 65536
for name, path in libs.items():
    vec = FixedWingVecEnv(cfg, num_envs=n, device=0, derived_views=False, _lib_path=path)
    vec.reset()
    actor = DeviceActor.for_env(vec, seed=1)
    actor.load_policy(MlpPolicy(12))
    ro = FusedRollout(vec, actor, 128, graph=True, fused="auto")
    ro.run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); ro.run(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 128 * 1e6)
    print("%-16s %7.2f us/step  (spec %d)" % (name, best, vec.spec_index), flush=True)
    vec.close()
