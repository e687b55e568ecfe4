import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import configs, parity
from gym_fixed_wing.vec_env import FixedWingVecEnv
from gym_fixed_wing import _native as nat

def run(kind, ckw, skw, n, steps):
    cfg = configs.reference_like(kind)
    acts = np.random.default_rng(1).uniform(-1, 1, size=(steps, n, 3)).astype(np.float32)
    outs = []
    for rep in range(2):
        vec = FixedWingVecEnv(cfg, num_envs=n, device=0, config_kw=ckw, sim_config_kw=skw, seed=9, as_numpy=True)
        hist = [(np.array(vec.reset()), None, None, np.array(parity._np(vec.state)))]
        for t in range(steps):
            obs, rew, done, _ = vec.step(acts[t])
            hist.append((np.array(obs), np.array(rew), np.array(done), np.array(parity._np(vec.state))))
        outs.append(hist)
        L = vec.layout
        vec.close()
    names = [n_ for n_, _ in nat.Layout._fields_]
    offs = sorted([(getattr(L, n_), n_) for n_ in names if n_ not in ("rows", "window", "lag_depth")])
    for t, (a, b) in enumerate(zip(*outs)):
        if not np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32)):
            rows, envs = np.nonzero(a[3].view(np.uint32) != b[3].view(np.uint32))
            sect = lambda r: [nm for o, nm in offs if o <= r][-1]
            print(kind, "t=%d state differs: rows %s (%s) n_envs %d first envs %s" % (t, sorted(set(rows))[:12], sorted(set(sect(r) for r in rows)), len(set(envs)), sorted(set(envs))[:8]))
            r, e = rows[0], envs[0]
            print("   example row %d env %d: %r vs %r" % (r, e, a[3][r, e], b[3][r, e]))
            break
        if not np.array_equal(a[0], b[0]):
            print(kind, "t=%d obs differs" % t, np.argwhere(a[0] != b[0])[:5])
            break
    else:
        print(kind, "deterministic over", steps, "steps")

run("default", None, None, 4096, 8)
run("cnn", None, None, 4096, 8)
run("cnn", {"observation": {"step": 2}}, None, 4096, 12)
run("cnn", {"observation": {"step": 2}}, {"turbulence": True, "turbulence_intensity": "moderate"}, 4096, 12)
run("cnn", {"observation": {"step": 2}}, {"turbulence": True, "turbulence_intensity": "moderate"}, 65536, 12)
