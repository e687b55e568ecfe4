#!/usr/bin/env python3
"""Phase stamps of k_actor_act (measurement build of tools/timeline.py).  Stamps: 0 entry | 1 weight DMA + observation loads
issued | 2 statistics folded | 3 observations normalised and split | 4 weights landed | 5 both MLPs done | 6 outputs stored."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from gym_fixed_wing import presets, _native as nat
from gym_fixed_wing.vec_env import FixedWingVecEnv
from gym_fixed_wing.actor import DeviceActor
from gym_fixed_wing.rollout import MlpPolicy
LIB = os.path.join(ROOT, "fixed-wing-gym_amd", "gym_fixed_wing", "_abl", "libfwgym_timeline.so")
cfg, ckw, skw, n, desc = presets.workload("c5")
vec = FixedWingVecEnv(cfg, num_envs=n, device=0, config_kw=ckw, sim_config_kw=skw, derived_views=False, _lib_path=LIB)
vec.reset()
lib = vec._lib
for precise in (True, False):
    actor = DeviceActor.for_env(vec, seed=7, precise=precise)
    actor.load_policy(MlpPolicy(vec.obs_dim))
    actor.attach(vec)
    nb = (n + 255) // 256
    trace = torch.zeros((nb, 8, 16), dtype=torch.int64, device="cuda")
    lib.fwg_debug_set_actor_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.fwg_debug_set_actor_trace(actor._handle, ctypes.c_void_p(trace.data_ptr()))
    a0 = torch.rand((n, 3), device="cuda") * 2 - 1
    out = {k: torch.zeros((n,) + s, device="cuda") for k, s in (("o", (12,)), ("a", (3,)), ("v", ()), ("l", ()))}
    rows, evs = [], []
    for rep in range(30):
        vec.step_device(a0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        actor.act(vec._obs, reward=vec._rew, done=vec._done, norm_obs=out["o"], action=out["a"], value=out["v"], logp=out["l"])
        e1.record(); torch.cuda.synchronize()
        if rep >= 10:
            rows.append(trace.cpu().numpy().astype(np.float64)); evs.append(e0.elapsed_time(e1) * 1e3)
    T = np.stack(rows)
    rel = T - T[:, :, :, 0:1].min(axis=2, keepdims=True)
    med = np.median(rel.reshape(-1, 16), axis=0)
    spread = np.median(T[:, :, :, 0].max(axis=(1, 2)) - T[:, :, :, 0].min(axis=(1, 2)))
    print("precise={}: event {:.2f} us, first->last block start {:.0f} ticks; stamps (ticks since block start, median over waves): {}".format(
        precise, np.median(evs), spread, "  ".join("{}:{:.0f}".format(i, med[i]) for i in range(7))))
    actor.close()
