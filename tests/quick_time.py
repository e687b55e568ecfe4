"""Quick throughput probe (not the bench): times fwg_step on device-resident actions with torch events."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")]
import torch, numpy as np
import configs
from gym_fixed_wing.vec_env import FixedWingVecEnv

def run(kind, n, ckw, skw, steps=200):
    cfg = configs.reference_like(kind)
    vec = FixedWingVecEnv(cfg, num_envs=n, device=0, config_kw=ckw, sim_config_kw=skw, seed=1, derived_views=False)
    assert vec.spec_index >= 0
    vec.reset()
    acts = [torch.rand((n, 3), device="cuda") * 2 - 1 for _ in range(16)]
    for t in range(50): vec.step_device(acts[t % 16])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); e0.record()
    for t in range(steps): vec.step_device(acts[t % 16])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print(json.dumps({"kind": kind, "n": n, "turb": bool(skw), "ms_per_step": ms, "wall_ms": (time.time() - t0) * 1e3 / steps, "env_steps_per_s": n / ms * 1e3}), flush=True)
    vec.close()

if __name__ == "__main__":
    turb = {"turbulence": True, "turbulence_intensity": "moderate"}
    if len(sys.argv) > 1 and sys.argv[1] == "c3only":
        run("cnn", 65536, {"observation": {"step": 2}}, turb, steps=400)
        sys.exit(0)
    run("default", 4096, None, None)
    run("default", 65536, None, None)
    run("cnn", 65536, {"observation": {"step": 2}}, turb)
    run("cnn", 262144, {"observation": {"step": 2}}, turb)
    run("cnn", 1048576, {"observation": {"step": 2}}, turb, steps=50)
