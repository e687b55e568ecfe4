import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")]
import torch, numpy as np
import configs
from gym_fixed_wing.vec_env import FixedWingVecEnv
def run(kind, n, ckw, skw):
    cfg = configs.reference_like(kind)
    vec = FixedWingVecEnv(cfg, num_envs=n, device=0, config_kw=ckw, sim_config_kw=skw, seed=1, derived_views=False)
    vec.reset()
    acts = [torch.rand((n, 3), device="cuda") * 2 - 1 for _ in range(16)]
    for t in range(100): vec.step_device(acts[t % 16])
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 160)()
    vec._lib.fwg_read_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    vec._lib.fwg_read_trace(vec._handle, out)
    a = np.array(out[:], dtype=np.uint64).view(np.int64).reshape(10, 16)
    names = ["start", "sim loads landed", "gym loads issued", "sim_step done", "store_sim issued", "dma landed", "gym logic done", "store_gym issued", "obs built", "done-phase", "write_obs issued", "stores drained"]
    nb = (n // 64 + 96) // 97
    print(kind, n, "spec", vec.spec_index)
    for i, nm in enumerate(names):
        print("%-20s" % nm, " ".join("%7d" % v for v in a[:min(nb, 8), i]))
    vec.close()
turb = {"turbulence": True, "turbulence_intensity": "moderate"}
run("cnn", 65536, {"observation": {"step": 2}}, turb)
run("cnn", 4096, {"observation": {"step": 2}}, turb)
