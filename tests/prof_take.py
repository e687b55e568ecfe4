"""Measurement: what fwg_reduce_success_device costs per call (episode collection + handing the success sums out), for the two forms
the library has -- two launches (default) and one launch with a device-wide ticket (FWGYM_TAKE=ticket).  HIP events around 200
calls, each after one env step of 65 536 envs in the steady state (~33 finished episodes per call).  python tests/prof_take.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path[:0] = [os.path.join(ROOT, "fixed-wing-gym_amd")]
    import numpy as np
    import torch
    from gym_fixed_wing import presets
    from gym_fixed_wing.vec_env import FixedWingVecEnv
    cfg, ckw, skw, n, _ = presets.workload("c3")
    vec = FixedWingVecEnv(cfg, num_envs=n, device=0, config_kw=ckw, sim_config_kw=skw, derived_views=False, seed=0)
    vec.reset()
    perm = np.random.RandomState(1).permutation(n)
    a = torch.rand((n, 3), device="cuda") * 2 - 1
    for k in range(200):    # a coarse stagger: ends spread over the steps that follow
        vec.reset(indices=np.sort(perm[k::200]))
        for _ in range(10):
            vec.step_device(a, want_obs=False)
    out = torch.zeros(16, device="cuda")
    for _ in range(20):
        vec.step_device(a, want_obs=False)
        vec.reduce_success_device(out)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(400)]
    eps = 0.0
    for i in range(200):
        vec.step_device(a, want_obs=False)
        e[2 * i].record()
        vec.reduce_success_device(out)
        e[2 * i + 1].record()
        eps += float(out[0])
    torch.cuda.synchronize()
    t = sorted(e[2 * i].elapsed_time(e[2 * i + 1]) * 1e3 for i in range(200))
    print("FWGYM_TAKE={:8s} fwg_reduce_success_device: median {:.1f} us, p10 {:.1f}, p90 {:.1f} (HIP events, 200 calls, {:.0f} episodes collected per call)".format(
        os.environ.get("FWGYM_TAKE", "(unset)"), t[100], t[20], t[180], eps / 200))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for mode in ("two", "ticket", "two", "ticket"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, FWGYM_TAKE=mode))
