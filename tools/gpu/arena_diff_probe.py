"""Developer probe (GPU): two kernel tiers of one configuration stepped side by side; reports the first state-arena words that differ."""
import copy, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from gym_fixed_wing import presets
from gym_fixed_wing.vec_env import FixedWingVecEnv
import test_shape_instance as T
warnings.simplefilter("ignore")
cfg, ckw, skw, _, _ = presets.workload("c3")
ckw = T._merged(ckw, T.VALUE_TWEAKS); skw = T._merged(skw, {"turbulence_intensity": "light"})
n = 1100
layout = sys.argv[1]
def make(mode):
    os.environ.pop("FWGYM_SHAPE", None)
    kw = dict(specialize=False)
    if mode == "generic": os.environ["FWGYM_SHAPE"] = "0"
    v = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=n, device=0, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw),
                        derived_views=False, seed=5, obs_layout=layout, **kw)
    os.environ.pop("FWGYM_SHAPE", None)
    return v
a, b = make(sys.argv[2]), make(sys.argv[3])
L = a.layout
names = ["sim", "cold", "derived", "gym", "tprop", "goal", "act_ring", "cmd_ring", "end_ring", "lag_ring", "draw", "aero", "aero_next", "fscale", "fscale_next", "model_raw", "model_raw_next", "fin"]
offs = sorted([(int(getattr(L, k)), k) for k in names if hasattr(L, k)])
def region(word):
    r = "?"
    for o, k in offs:
        if o <= word: r = "{}+{}".format(k, word - o)
    return r
print("instances", a.spec_index, b.spec_index, "rows", int(L.rows), "lag_depth", int(L.lag_depth), "lag_groups", int(L.lag_groups))
a.reset(); b.reset()
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
watch = {1015, 814}
seen = set()
for t in range(181):
    act = (torch.rand((n, 3), device="cuda", generator=gen) * 2 - 1) * (2.5 if t % 13 == 0 else 1.3)
    a.step(act); b.step(act)
    sa, sb = a.state.view(torch.int32), b.state.view(torch.int32)     # [rows/4][env][4]
    ne = (sa != sb)
    lo, hi = int(L.lag_ring) // 4, int(L.lag_ring) // 4 + int(L.lag_depth) * int(L.lag_groups)
    d = (a.state[lo:hi] - b.state[lo:hi]).abs()          # [slot*ng+g][env][4]
    bad = torch.nonzero(d.amax(dim=2) > 1e-3)
    for gi, e in bad.tolist()[:6]:
        key = (gi, e)
        if key in seen: continue
        seen.add(key)
        print("step", t, "g mod 9 =", (t) % 9, "env", e, "lane", e % 64, "ring slot", gi // int(L.lag_groups), "group", gi % int(L.lag_groups),
              "a", [round(float(x), 3) for x in a.state[lo + gi, e]], "b", [round(float(x), 3) for x in b.state[lo + gi, e]],
              "age", int(a.field("steps_count")[e]), "done now", int(a._done[e]))
print("done")
