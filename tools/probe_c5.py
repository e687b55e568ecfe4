import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from gym_fixed_wing import presets
from gym_fixed_wing.vec_env import FixedWingVecEnv
def timed(fn, reps=100):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]
def run(name, wl="c5", split="1", mutate=None, stagger=32, act_mode="const", **kw):
    os.environ["FWGYM_SPLIT"] = split
    cfg, ckw, skw, n, desc = presets.workload(wl)
    n = 65536
    cfg = copy.deepcopy(cfg)
    if mutate: mutate(cfg)
    vec = FixedWingVecEnv(cfg, num_envs=n, device=0, config_kw=ckw, sim_config_kw=skw, derived_views=False, **kw)
    vec.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    pool = [torch.rand((n, 3), device="cuda", generator=gen) * 2 - 1 for _ in range(8)]
    per = int(vec.cfg["steps_max"]) // stagger if stagger else 0
    t = 0
    for k in range(stagger):
        vec.reset(indices=np.arange(k, n, stagger))
        for _ in range(per):
            vec.step_device(pool[0] if act_mode == "const" else pool[t % 8], want_obs=False); t += 1
    for _ in range(20): vec.step_device(pool[0] if act_mode == "const" else pool[t % 8]); t += 1
    us = timed(lambda: vec.step_device(pool[0] if act_mode == "const" else pool[3]))
    done = vec._done.float().sum().item()
    term = np.bincount(vec._term.cpu().numpy()[vec._done.cpu().numpy() > 0], minlength=1)
    print("{:34s} {:7.2f} us  spec {} done/step {} term {}".format(name, us, vec.spec_index, done, dict(enumerate(term.tolist()))), flush=True)
    vec.close()
run("c5 staggered const actions")
run("c5 staggered random actions", act_mode="rand")
run("c5 sync", stagger=0)
run("c5 staggered, one-wave kernel", split="0")
run("c5 staggered, no auto reset", auto_reset=False)
def nometrics(c): c["metrics"] = []
run("c5 staggered, no metrics (generic)", mutate=nometrics)
run("c3 staggered const", wl="c3")
run("c3 staggered random", wl="c3", act_mode="rand")
run("c2 cfg staggered random", wl="c2", act_mode="rand")
