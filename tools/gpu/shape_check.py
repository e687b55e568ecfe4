"""Developer check (GPU): a shape instance against the frozen instance of the same configuration, and its speed."""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from gym_fixed_wing import presets
from gym_fixed_wing.vec_env import FixedWingVecEnv

lib = sys.argv[1] if len(sys.argv) > 1 else None
cfg, ckw, skw, n, desc = presets.workload("c3")
ckw = dict(ckw or {}); ckw["steps_max"] = int(os.environ.get("STEPS_MAX", "120"))
n = 8192

def make(force):
    if force: os.environ["FWGYM_SHAPE"] = "force"
    else: os.environ.pop("FWGYM_SHAPE", None)
    kw = dict(_lib_path=lib) if lib else {}
    v = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=n, device=0, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw), derived_views=False,
                        seed=3, obs_log_rows=presets.OBS_LOG_ROWS, specialize=False, **kw)
    return v
a, b = make(False), make(True)
print("instances:", a.spec_index, b.spec_index)
oa, ob = a.reset(), b.reset()
print("reset obs max |diff|", float((oa - ob).abs().max()))
g = torch.Generator(device="cuda"); g.manual_seed(0)
worst = 0.0; nd = 0
for t in range(300):
    act = torch.rand((n, 3), device="cuda", generator=g) * 2 - 1
    oa, ra, da, _ = a.step(act); ob, rb, db, _ = b.step(act)
    assert torch.equal(da, db), t
    nd += int(da.sum())
    worst = max(worst, float((oa - ob).abs().max()), float((ra - rb).abs().max()))
print("300 steps, episode ends", nd, "worst |diff| obs/reward", worst)
for v, name in ((a, "first"), (b, "second")):
    acts = [torch.rand((n, 3), device="cuda") * 2 - 1 for _ in range(8)]
    for t in range(50): v.step_device(acts[t % 8])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(500): v.step_device(acts[t % 8])
    torch.cuda.synchronize(); print(name, "instance", v.spec_index, "eager us/step", (time.perf_counter() - t0) / 500 * 1e6)
