#!/usr/bin/env python3
"""C5 rollout pieces timed separately with HIP events (eager launches): fwg_step (with the attached head's moments) and
fwg_actor_act, for synchronised and staggered episode ages.  python3 tools/prof_c5.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from gym_fixed_wing import presets
from gym_fixed_wing.vec_env import FixedWingVecEnv
from gym_fixed_wing.actor import DeviceActor
from gym_fixed_wing.rollout import MlpPolicy


def timed(fn, reps=200):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2], t[int(len(t) * 0.9)]


for stagger in (0, 32):
    cfg, ckw, skw, n, desc = presets.workload("c5")
    vec = FixedWingVecEnv(cfg, num_envs=n, device=0, config_kw=ckw, sim_config_kw=skw, derived_views=False)
    vec.reset()
    a0 = torch.rand((n, 3), device="cuda") * 2 - 1
    if stagger:
        per = int(vec.cfg["steps_max"]) // stagger
        for k in range(stagger):
            vec.reset(indices=np.arange(k, n, stagger))
            for _ in range(per):
                vec.step_device(a0, want_obs=False)
    torch.manual_seed(0)
    actor = DeviceActor.for_env(vec, seed=7)
    actor.load_policy(MlpPolicy(vec.obs_dim))
    for _ in range(50):
        vec.step_device(a0)
    torch.cuda.synchronize()
    plain = timed(lambda: vec.step_device(a0))
    dones = float(vec._done.float().mean().item()) * n
    terms = np.bincount(vec._term.cpu().numpy()[vec._done.cpu().numpy() > 0], minlength=4)[:8]
    actor.attach(vec)
    attached = timed(lambda: vec.step_device(a0))
    out = {k: torch.zeros((n,) + s, device="cuda") for k, s in (("o", (12,)), ("a", (3,)), ("v", ()), ("l", ()))}
    act = timed(lambda: actor.act(vec._obs, reward=vec._rew, done=vec._done, norm_obs=out["o"], action=out["a"], value=out["v"], logp=out["l"]))
    print("stagger {:2d}: fwg_step {:.2f} us (p90 {:.2f}), with attached head {:.2f} us (p90 {:.2f}), fwg_actor_act {:.2f} us (p90 {:.2f}); "
          "~{:.0f} envs done per step, termination codes {}".format(stagger, plain[0], plain[1], attached[0], attached[1], act[0], act[1], dones, terms.tolist()))
    actor.close(); vec.close()
