import copy, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import configs, parity
from gym_fixed_wing.vec_env import FixedWingVecEnv
warnings.simplefilter("ignore")
cfg = configs.reference_like("cnn")
ckw = {"observation": {"step": 2}, "steps_max": 45, "simulator": {"states": {6: {"constraint_min": -60, "constraint_max": 60}}}}
skw = {"turbulence": True, "turbulence_intensity": "moderate"}
n, steps = 70, 130
rng = np.random.default_rng(5)
acts = rng.uniform(-1.3, 1.3, (steps, n, 3)).astype(np.float32)
for mode in sys.argv[1:]:
    os.environ.pop("FWGYM_SHAPE", None)
    kw = {}
    if mode == "generic": os.environ["FWGYM_SHAPE"] = "0"; kw["specialize"] = False
    elif mode == "shape": kw["specialize"] = False
    elif mode == "jit": kw["specialize"] = True
    elif mode == "shape_dense": kw["specialize"] = False; kw["obs_layout"] = "dense"
    elif mode == "jit_dense": kw["specialize"] = True; kw["obs_layout"] = "dense"
    vec = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=n, device=0, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw), seed=11, as_numpy=True, **kw)
    orc = parity.make_oracles(cfg, n, 11, config_kw=ckw, sim_config_kw=skw)
    try:
        res = parity.run_gym_parity(vec, orc, steps, lambda t: acts[t], rtol=4e-3, atol=4e-3)
        print(mode, "instance", vec.spec_index, "OK", res)
    except Exception as e:
        print(mode, "instance", vec.spec_index, "FAIL", str(e)[:300])
    vec.close()
