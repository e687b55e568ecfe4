"""Stress loop in SUITE CONTEXT (not collected by pytest; tests/test_rollout.py runs a short version of it):
fwg_rollout_step (head + env step in one launch) against fwg_actor_act + fwg_step (two launches), bit for bit -- rollout buffers,
running statistics, the env's state arena -- with, between the two paths of every iteration, the kernels the GPU suite runs
around tests/test_rollout.py::test_fused_launch_equals_two_launches_on_gpu: another env's two-wave step kernel with an attached
head (batch moments), fwg_actor_act on a row-log env, replayed hipGraphs of both, and allocator churn (freed and re-used device
memory holding other tests' data).  That test failed twice in ~25 runs of the whole suite in round 4, never in isolation.

    python tests/soak_suite_context.py [seconds] [dump_dir]

Prints one line per mismatch, keeps the first failing pair of buffers as an .npz under dump_dir, and a summary line at the end."""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch

import configs
import test_rollout as tr
from gym_fixed_wing.actor import DeviceActor
from gym_fixed_wing.rollout import FusedRollout, MlpPolicy
from gym_fixed_wing.vec_env import FixedWingVecEnv

TURB = {"turbulence": True, "turbulence_intensity": "moderate"}


def context_kernels(it):
    """What else the suite has on the device around the test: other configurations' kernels, attached heads, graph replays."""
    cnn = configs.reference_like("cnn")
    n = [4096, 8192, 65536][it % 3]
    # (a) C3-like env, dense batch, attached head: k_step2 with the batch moments + k_actor_act, eager and replayed
    vec = FixedWingVecEnv(copy.deepcopy(cnn), num_envs=n, device=0, config_kw={"observation": {"step": 2}}, sim_config_kw=copy.deepcopy(TURB),
                          obs_layout="dense", derived_views=False, seed=100 + it)
    vec.reset()
    actor = DeviceActor.for_env(vec, seed=9 + it)
    actor.load_policy(MlpPolicy(vec.obs_dim))
    ro = FusedRollout(vec, actor, 8, graph=(it % 2 == 0), fused=False)
    for _ in range(2):
        ro.run()
    actor.close(); vec.close()
    # (b) the same workload on the row log: the head reads the window in place
    vec = FixedWingVecEnv(copy.deepcopy(cnn), num_envs=n, device=0, config_kw={"observation": {"step": 2}}, sim_config_kw=copy.deepcopy(TURB),
                          derived_views=False, seed=200 + it)
    vec.reset()
    actor = DeviceActor.for_env(vec, seed=3 + it)
    actor.load_policy(MlpPolicy(60))
    ro = FusedRollout(vec, actor, 6, graph=(it % 2 == 1))
    ro.run()
    actor.close(); vec.close()
    # (c) allocator churn: blocks of other sizes filled with non-zero data, freed again
    junk = [torch.full((int(s),), float(it + 1), device="cuda") for s in (1 << 20, 3 << 20, 7 << 18)]
    del junk
    torch.cuda.synchronize()


def main(budget=120.0, dump_dir=None, quiet=False):
    cfg = configs.reference_like("examples")
    t0, it, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        n = [65536, 4096 + 37, 65536, 16384, 256 * 3 + 1][it % 5]
        graph = it % 3 == 2
        steps, reps = (16, 3) if graph else (16, 2)
        mk = lambda: FixedWingVecEnv(cfg, num_envs=n, device=0, derived_views=False, seed=2 + it)
        two = tr._rollouts(mk, False, steps, reps, graph=graph)
        context_kernels(it)
        one = tr._rollouts(mk, True, steps, reps, graph=graph)
        try:
            tr._assert_same_rollouts(one, two, "iteration {} n={} graph={}:".format(it, n, graph))
        except AssertionError as e:
            bad += 1
            print("MISMATCH", str(e)[:1500].replace("\n", " | "), flush=True)
            if dump_dir and bad == 1:
                os.makedirs(dump_dir, exist_ok=True)
                keep = {}
                for tag, (ro, st, arena, obs) in (("one", one), ("two", two)):
                    for rep, b in enumerate(ro):
                        for k, v in b.items():
                            keep["{}_r{}_{}".format(tag, rep, k)] = v
                    for k, v in st.items():
                        keep["{}_stat_{}".format(tag, k)] = np.asarray(v)
                    keep[tag + "_obs"] = obs
                np.savez_compressed(os.path.join(dump_dir, "soak_first_mismatch_it{}.npz".format(it)), **keep)
        it += 1
    line = "suite-context soak: iterations {} mismatches {} seconds {:.0f}".format(it, bad, time.time() - t0)
    if not quiet:
        print(line, flush=True)
    return it, bad


if __name__ == "__main__":
    b = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    d = sys.argv[2] if len(sys.argv) > 2 else None
    main(b, d)
