"""Stress loop (not collected by pytest): fwg_rollout_step (one launch) against fwg_actor_act + fwg_step (two launches), bit for bit,
over many fresh envs / batch sizes / eager and replayed -- to catch timing-dependent differences a single run of
tests/test_rollout.py::test_fused_launch_equals_two_launches_on_gpu may miss.  python tests/soak_rollout.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import configs
import test_rollout as tr
from gym_fixed_wing.vec_env import FixedWingVecEnv

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
cfg = configs.reference_like("examples")
t0, it, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    n = [65536, 4096 + 37, 65536, 16384, 256 * 3 + 1][it % 5]
    graph = it % 3 == 2
    steps, reps = (16, 3) if graph else (24, 2)
    mk = lambda: FixedWingVecEnv(cfg, num_envs=n, device=0, derived_views=False, seed=2 + it)
    two = tr._rollouts(mk, False, steps, reps, graph=graph)
    one = tr._rollouts(mk, True, steps, reps, graph=graph)
    try:
        tr._assert_same_rollouts(one, two)
    except AssertionError as e:
        bad += 1
        print("MISMATCH iteration", it, "n", n, "graph", graph, str(e)[:600].replace("\n", " | "), flush=True)
        (ra, sa, xa, oa), (rb, sb, xb, ob) = one, two
        for rep, (u, v) in enumerate(zip(ra, rb)):
            for k in u:
                d = np.argwhere(u[k] != v[k])
                if len(d):
                    print("   rollout", rep, k, "differs at", len(d), "places; first", d[:5].tolist(), flush=True)
    it += 1
print("iterations", it, "mismatches", bad)
