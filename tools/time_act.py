#!/usr/bin/env python3
"""Times fwg_actor_act alone (hipGraph replay of 200 launches, 65 536 envs, obs 12) for the product library and every
variant under gym_fixed_wing/_abl/."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fixed-wing-gym_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from gym_fixed_wing import _native as nat
from gym_fixed_wing.actor import DeviceActor
from gym_fixed_wing.rollout import MlpPolicy
OUT = os.path.join(PKG, "gym_fixed_wing", "_abl")
libs = {"product": None}
if os.path.isdir(OUT):
    libs.update({f[9:-3]: os.path.join(OUT, f) for f in sorted(os.listdir(OUT)) if f.endswith(".so")})
n, d = 65536, 12
obs = torch.randn((n, d), device="cuda")
rew = torch.randn(n, device="cuda")
for name, path in libs.items():
    for precise in (True, False):
        a = DeviceActor(n, d, precise=precise, _lib=nat.load_library(path))
        a.load_policy(MlpPolicy(d))
        outs = [torch.zeros((n, d), device="cuda"), torch.zeros((n, 3), device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
        def call():
            a.act(obs, reward=rew, norm_obs=outs[0], action=outs[1], value=outs[2], logp=outs[3], norm_reward=outs[4])
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            call(); call()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(200): call()
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
        print("%-14s precise=%d  %7.2f us/launch" % (name, precise, best), flush=True)
        a.close()
