#!/usr/bin/env python3
"""Phase stamps of k_rollout (head + env step in one launch) from a -DFWG_TIMELINE build (tools/timeline.py build).
Head stamps (fwgym_actor.h FWG_ATL): 0 entry | 1 weight DMA + loads issued | 2 statistics folded | 3 observations normalised |
4 weights landed | 5 both MLPs done | 6 outputs stored.  Step stamps (fwgym.hip FWG_TL), per role: 0 entry into the step
phase | 1 loads issued | 2 integration done / gym pre-work done | 3 simulator rows stored | 5 gym logic done | 6 bookkeeping
stored | 7 observation built | 8 episode-end branch done | 9 outputs issued.  All in ticks since the workgroup's first stamp.
    gpurun -- python tools/timeline_rollout.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from gym_fixed_wing import presets
from gym_fixed_wing.vec_env import FixedWingVecEnv
from gym_fixed_wing.actor import DeviceActor
from gym_fixed_wing.rollout import MlpPolicy
LIB = os.path.join(ROOT, "fixed-wing-gym_amd", "gym_fixed_wing", "_abl", os.environ.get("TL_LIB", "libfwgym_timeline.so"))
cfg, ckw, skw, n, desc = presets.workload("c5")
vec = FixedWingVecEnv(cfg, num_envs=n, device=0, config_kw=ckw, sim_config_kw=skw, derived_views=False, _lib_path=LIB)
vec.reset()
lib = vec._lib
for precise in (True, False):
    actor = DeviceActor.for_env(vec, seed=7, precise=precise)
    actor.load_policy(MlpPolicy(vec.obs_dim))
    actor.attach(vec)
    assert actor.rollout_available(vec)
    nb = (n + 255) // 256
    tr_head = torch.zeros((nb, 8, 16), dtype=torch.int64, device="cuda")
    tr_step = torch.zeros((nb, 8, 32), dtype=torch.int64, device="cuda")   # (FWG_TL_W)
    lib.fwg_debug_set_actor_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.fwg_debug_set_actor_trace(actor._handle, ctypes.c_void_p(tr_head.data_ptr()))
    lib.fwg_debug_set_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.fwg_debug_set_trace(vec._handle, ctypes.c_void_p(tr_step.data_ptr()))
    out = {k: torch.zeros((n,) + s, device="cuda") for k, s in (("o", (12,)), ("a", (3,)), ("v", ()), ("l", ()))}
    actor.observe(vec._obs)
    H, S, evs = [], [], []
    for rep in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        actor.rollout_step(vec, norm_obs=out["o"], action=out["a"], value=out["v"], logp=out["l"])
        e1.record(); torch.cuda.synchronize()
        if rep >= 10:
            H.append(tr_head.cpu().numpy().astype(np.float64)); S.append(tr_step.cpu().numpy().astype(np.float64)); evs.append(e0.elapsed_time(e1) * 1e3)
    H, S = np.stack(H), np.stack(S)                     # [rep, block, wave, stamp]
    t0 = H[:, :, :, 0].min(axis=2)[:, :, None, None]     # block start
    h = np.median((H - t0).reshape(-1, 16), axis=0)
    phys = np.median((S[:, :, :4] - t0).reshape(-1, 32), axis=0)
    gym = np.median((S[:, :, 4:] - t0).reshape(-1, 32), axis=0)
    end = np.median(np.nanmax(np.where(S > 0, S - t0, np.nan), axis=(2, 3)))
    spread = np.median(H[:, :, :, 0].max(axis=(1, 2)) - H[:, :, :, 0].min(axis=(1, 2)))
    print("precise={}: launch {:.2f} us (eager, events); first->last block start {:.0f} ticks; block lifetime (last stamp) {:.0f} ticks".format(
        precise, np.median(evs), spread, end))
    print("  head   :", "  ".join("{}:{:.0f}".format(i, h[i]) for i in (0, 1, 2, 3, 4, 8, 9, 10, 11, 12, 5, 6)))
    print("  physics:", "  ".join("{}:{:.0f}".format(i, phys[i]) for i in (0, 1, 2, 3)))
    print("  gym    :", "  ".join("{}:{:.0f}".format(i, gym[i]) for i in (0, 2, 4, 5, 6, 14, 15, 7, 11, 8, 9)))
    actor.close()
