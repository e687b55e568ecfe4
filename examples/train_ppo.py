#!/usr/bin/env python3
"""Train a PPO attitude controller on the MI355X-native env -- the reference's examples/train_rl_controller.py
(`VecNormalize(SubprocVecEnv(...))` + `PPO2(MlpPolicy, env).learn(5e6, callback=monitor_training)` with the curriculum rule of
its callback, :80-87) with the whole data path on the device: rollouts by the HIP head + env step kernels, advantages by fwg_gae,
the PPO2 update in torch on the same buffers, the success sums all-gathered over RCCL, the curriculum raised on every rank.

    python examples/train_ppo.py --envs 4096 --timesteps 100e6 --out model.npz
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_ppo.py --envs 32768

The reference trains 4 sub-process envs for 5 M steps (~8 h of PyFly on 4 cores; examples/tensorboard.png: success_all ~0.8 at
4-5 M steps).  4 096 device envs deliver 5 M steps in ten rollouts; a batch of 524 288 transitions per update learns less per
SAMPLE than 512 do, so the recipe below spends more samples (default 100 M, about a minute) with 128 minibatches per update and
twice the learning rate -- everything else is PPO2's defaults (gym_fixed_wing/ppo.py PPO2_DEFAULTS).  Measured (MI355X, seed 0):
curriculum level 1 after 58 M steps, success_all 0.79 at 74 M, 0.90 at 148 M (profiles/r06_ppo_training.txt)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fixed-wing-gym_amd")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from gym_fixed_wing import presets  # noqa: E402
from gym_fixed_wing.distributed import CurriculumSchedule, make_sharded_env  # noqa: E402
from gym_fixed_wing.ppo import PPO  # noqa: E402


def train(envs=4096, timesteps=100e6, seed=0, nminibatches=128, noptepochs=4, learning_rate=5e-4, n_steps=128, curriculum=True,
          config="examples", log=print, rank=0, world=1, local=0, fused=None, ent_coef=0.01, on_update=None):
    vec = make_sharded_env(presets.preset(config), total_envs=envs, rank=rank, world_size=world, device=local, derived_views=False,
                           seed=seed)
    sched = CurriculumSchedule(level=0.25 if curriculum else 1.0)     # (train_rl_controller.py:162: curriculum_level = 0.25)
    vec.set_curriculum_level(sched.level)
    vec.reset()
    ppo = PPO(vec, seed=seed, curriculum=sched, n_steps=n_steps, nminibatches=nminibatches, noptepochs=noptepochs,
              learning_rate=learning_rate, fused=fused, ent_coef=ent_coef)
    t0 = time.perf_counter()
    window = []    # success over the last finished episodes (the reference's ep_info_buf holds the last 100)

    def on_info(info):
        if info["episodes"] > 0:
            window.append((info["episodes"], info["success"]["all"], info["level"]))
        if rank == 0 and log is not None and (info["episodes"] > 0 or info["update"] % 20 == 0):
            log("update {:4d}  steps {:10.3e}  level {:.2f}  episodes {:6d}  success_all {}  pg {:+.4f}  vf {:.4f}  kl {:.4f}  std-entropy {:+.2f}  {:.0f} s".format(
                info["update"], info["timesteps"], info["level"], info["episodes"],
                "{:.3f}".format(info["success"]["all"]) if info["episodes"] else "  -  ", info["pg_loss"], info["vf_loss"], info["approx_kl"],
                info["entropy"], time.perf_counter() - t0))
        if on_update is not None:
            on_update(ppo, info)

    ppo.learn(timesteps, log=on_info)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return ppo, {"seconds": dt, "env_steps_per_s": ppo.num_timesteps / dt, "updates": ppo.updates, "episodes_log": window}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096, help="total over all ranks")
    ap.add_argument("--timesteps", type=float, default=100e6)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--nminibatches", type=int, default=128)
    ap.add_argument("--noptepochs", type=int, default=4)
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--ent-coef", type=float, default=0.01)
    ap.add_argument("--disable-curriculum", action="store_true")
    ap.add_argument("--out", default=None, help="save weights + VecNormalize statistics (.npz)")
    ap.add_argument("--curve", default=None, help="write the learning curve (JSON)")
    args = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    ppo, res = train(args.envs, args.timesteps, args.seed, args.nminibatches, args.noptepochs, args.lr, curriculum=not args.disable_curriculum,
                     rank=rank, world=world, local=local, ent_coef=args.ent_coef)
    if rank == 0:
        print("{:.3e} env-steps in {:.1f} s = {:.3e} env-steps/s INCLUDING the optimiser ({} updates)".format(
            ppo.num_timesteps, res["seconds"], res["env_steps_per_s"], res["updates"]))
        if args.out:
            ppo.save(args.out)
        if args.curve:
            with open(args.curve, "w") as f:
                json.dump({"args": vars(args), "history": ppo.history, **{k: v for k, v in res.items()}}, f)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
