#!/usr/bin/env python3
"""The data path of the reference's training script (examples/train_rl_controller.py:141-232 there: SubprocVecEnv +
VecNormalize + PPO2.step + the curriculum callback) on MI355X, without the optimiser: envs sharded over the ranks, the
rollout head (VecNormalize + MlpPolicy + sampling) as HIP kernels, one hipGraph per 128-step rollout, the success sums
all-gathered over RCCL after every rollout and the curriculum level raised by the reference's rule on every rank.

    python examples/collect_rollouts.py --envs 65536 --rollouts 20
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/collect_rollouts.py --envs 524288
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fixed-wing-gym_amd")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from gym_fixed_wing import presets  # noqa: E402
from gym_fixed_wing.actor import DeviceActor  # noqa: E402
from gym_fixed_wing.distributed import CurriculumSchedule, gather_success, make_sharded_env  # noqa: E402
from gym_fixed_wing.rollout import FusedRollout, MlpPolicy  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536, help="total over all ranks")
    ap.add_argument("--rollouts", type=int, default=20)
    ap.add_argument("--n-steps", type=int, default=128)
    args = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    vec = make_sharded_env(presets.preset("examples"), total_envs=args.envs, rank=rank, world_size=world, device=local,
                           derived_views=False, seed=0)
    curriculum = CurriculumSchedule(level=0.25)
    vec.set_curriculum_level(curriculum.level)
    vec.reset()
    torch.manual_seed(0)                                   # the same random-init policy on every rank
    actor = DeviceActor.for_env(vec, seed=1)               # sampling noise keyed by the global env ids
    actor.load_policy(MlpPolicy(vec.obs_dim))
    rollout = FusedRollout(vec, actor, args.n_steps, graph=True, fused="auto")
    rollout.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.rollouts):
        buf = rollout.run()                                # obs, actions, values, logp, rewards, dones: [n_steps, N, ...]
        # ... a PPO update on `buf` / rollout.last_value would go here ...
        summary = gather_success(vec)                      # RCCL all-gather of 16 floats per rank
        level = curriculum.update(vec, summary)
        if rank == 0 and (it % 5 == 0 or it == args.rollouts - 1):
            print("rollout {:3d}: episodes {:6d}  success(all) {:.3f}  curriculum level {:.2f}  mean |reward| {:.3f}".format(
                it, summary["episodes"], summary["success"]["all"], level, float(buf["rewards"].abs().mean())), flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print("{:.3e} env-steps/s over {} rank(s) (rollouts incl. success all-gather and curriculum)".format(
            args.rollouts * args.n_steps * args.envs / dt, world))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
