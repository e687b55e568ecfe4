import sys, time
sys.path[:0] = ["/root/repo", "/root/repo/fixed-wing-gym_amd", "/root/repo/tests"]
import torch
from gym_fixed_wing import presets
from gym_fixed_wing.vec_env import FixedWingVecEnv
for wl, rows in (("c3", 0), ("c3", 32), ("c2", 0), ("c5", 0)):
    cfg, ckw, skw, n, desc = presets.workload(wl)
    vec = FixedWingVecEnv(cfg, num_envs=65536, device=0, config_kw=ckw, sim_config_kw=skw, seed=5, derived_views=False, obs_log_rows=rows)
    vec.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    bad = 0; dones = 0
    t0 = time.time()
    for t in range(6000):
        a = (torch.rand((65536, 3), device="cuda", generator=gen) * 2 - 1) * (3.0 if t % 50 == 0 else 1.0)
        o, r, d = vec.step_device(a)
        if t % 100 == 0:
            bad += int((~torch.isfinite(o)).sum()) + int((~torch.isfinite(r)).sum())
            dones += int(d.sum())
    torch.cuda.synchronize()
    red = vec.reduce_success()
    st = vec.state
    print(wl, "log" if rows else "dense", "non-finite:", bad, "state finite:", bool(torch.isfinite(st[:7]).all()), "episodes:", int(red[0]), "%.1fs" % (time.time() - t0), flush=True)
    vec.close()
