#!/usr/bin/env python3
"""Per-step kernel time of the first steps of a fresh C3 VecEnv (all envs in step): python3 tools/first_steps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from gym_fixed_wing import presets
from gym_fixed_wing.vec_env import FixedWingVecEnv
cfg, ckw, skw, n, desc = presets.workload("c3")
vec = FixedWingVecEnv(cfg, num_envs=n, device=0, config_kw=ckw, sim_config_kw=skw, derived_views=False, seed=1)
acts = [torch.rand((n, 3), device="cuda") * 2 - 1 for _ in range(8)]
res = []
for rep in range(6):
    vec.reset()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for t, (a, b) in enumerate(ev):
        a.record(); vec.step_device(acts[t % 8], want_obs=False); b.record()
    torch.cuda.synchronize()
    res.append([a.elapsed_time(b) * 1e3 for a, b in ev])
med = np.median(np.array(res[1:]), axis=0)
print("step: us  " + "  ".join("{}:{:.1f}".format(t + 1, x) for t, x in enumerate(med)))
