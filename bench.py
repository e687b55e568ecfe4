#!/usr/bin/env python3
"""bench.py -- headline benchmark of the FixedWingAircraft.step() hot path on MI355X.

One "step" = one fused fwg_step launch advancing every env of this rank by one control step (dt = 0.01 s) including
observation, reward, done, metrics and auto-reset, on synthetic raw actions already resident in HBM.

Workload (BASELINE.json configs[2], the configuration the north-star target is quoted on): 65 536 envs per GPU, Dryden
turbulence on ("moderate"), observation matrix 5 x 12 with lag step 2.  `--workload c2` selects configs[1]
(4 096 envs, turbulence off, 14-vector).  Multi-GPU: one process per GPU (torchrun), envs sharded per rank with
global env ids (weak scaling), the only collective is the RCCL all-gather of the 16-float success-metric vector every
128 steps (examples/train_rl_controller.py:51-66,80-85 in the reference).

Prints ONE JSON line on rank 0 (see the contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
ALG_BYTES = {"c3": 937, "c2": 417, "c5": 409}   # algorithmic bytes per env-step, SURVEY.md section 8(d)
REDUCE_EVERY = 128


def workload(name):
    from gym_fixed_wing import presets
    return presets.workload(name)


def _cpu_worker(args):
    """Bounded sample of the SAME workload on one host core with the float64 oracle ("port" of the reference's
    per-env Python step)."""
    wl, seconds, seed = args
    import numpy as np
    from oracle.gym_restated import FixedWingOracle
    cfg, ckw, skw, _, _ = workload(wl)
    env = FixedWingOracle(cfg, config_kw=ckw, sim_config_kw=skw)
    env.seed(seed)
    env.reset()
    rng = np.random.default_rng(seed)
    for _ in range(20):
        env.step(rng.uniform(-1, 1, 3))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        _, _, done, _ = env.step(rng.uniform(-1, 1, 3))
        n += 1
        if done:
            env.reset()
    return n, time.perf_counter() - t0


def cpu_baseline(wl, seconds):
    import multiprocessing as mp
    cores = max(1, min(os.cpu_count() or 1, 16))
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(wl, seconds, 100 + i) for i in range(cores)])
    total = sum(n / dt for n, dt in res)
    return {"value": total, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "{} oracle processes (float64 NumPy restatement, 1 env each, same workload config) x {:.0f} s, "
                      "{} env-steps in total".format(cores, seconds, sum(n for n, _ in res))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c5"])
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (default: the workload's)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--obs-layout", default="dense", choices=["dense", "log"],
                    help="c3: 'dense' = fwg_step writes the [N][5][12] batch (default); 'log' = observation history kept "
                         "once as a row log, the observation is a zero-copy strided window of it (same values)")
    ap.add_argument("--rollout", default="auto", choices=["auto", "none", "fused"],
                    help="c5 only: 'fused' (default for c5) = env step + HIP rollout head (VecNormalize + MlpPolicy + "
                         "sampling), replayed from one hipGraph per chunk of steps; 'none' = env step on stored actions")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from gym_fixed_wing.vec_env import FixedWingVecEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("FWG_BENCH_FORCE_DIST", "0") == "1"   # the latter: exercise RCCL on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus {} but WORLD_SIZE {}".format(args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg, ckw, skw, n_envs, desc = workload(args.workload)
    if args.envs:
        n_envs = args.envs
    # derived_views=False: the rollout loop never reads roll/pitch/... back from the arena (they are in the
    # observations), so the kernel does not write those host-view rows
    from gym_fixed_wing import presets as _presets
    log_rows = _presets.OBS_LOG_ROWS if (args.obs_layout == "log" and args.workload == "c3") else 0
    vec = FixedWingVecEnv(cfg, num_envs=n_envs, device=local, config_kw=ckw, sim_config_kw=skw, seed=0,
                          env_id_base=rank * n_envs, auto_reset=True, derived_views=False, obs_log_rows=log_rows)
    vec.reset()
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    pool = [torch.rand((n_envs, 3), device=dev, generator=gen) * 2 - 1 for _ in range(32)]
    red_dev = torch.zeros(16, device=dev)
    gathered = torch.zeros(16 * world, device=dev) if use_dist else None

    def reduce_step():
        vec.reduce_success_device(red_dev)         # local sums, device to device, stream-ordered (no host sync)
        if use_dist:
            dist.all_gather_into_tensor(gathered, red_dev)   # RCCL over xGMI: 64 B per rank

    fused = args.workload == "c5" and args.rollout != "none"
    rollout = None
    if fused:   # BASELINE configs[4]: PPO rollout loop with a random-init 64-64 MlpPolicy, end to end
        from gym_fixed_wing.actor import DeviceActor
        from gym_fixed_wing.rollout import FusedRollout, MlpPolicy
        chunk = next((c for c in (128, 64, 32, 16, 8, 4, 2) if args.steps % c == 0 and args.warmup % c == 0), 0)
        if chunk == 0:
            raise SystemExit("--workload c5: --steps and --warmup must be even (hipGraph chunks)")
        torch.manual_seed(0)
        actor = DeviceActor.for_env(vec, seed=7, env_id_base=rank * n_envs)
        actor.load_policy(MlpPolicy(vec.obs_dim))
        rollout = FusedRollout(vec, actor, chunk, graph=True)

    def run(k, t_offset):
        if fused:
            done_steps = 0
            for _ in range(k // rollout.n_steps):
                rollout.run()
                done_steps += rollout.n_steps
                if done_steps % REDUCE_EVERY == 0:
                    reduce_step()
            return
        for t in range(k):
            vec.step_device(pool[(t_offset + t) % len(pool)])
            if (t + 1) % REDUCE_EVERY == 0:
                reduce_step()

    run(args.warmup, 0)
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(args.steps, args.warmup)
    ev1.record()
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())

    # dominant kernel: average fwg_step launch duration from HIP events on the launch stream, launches measured
    # one by one in a separate short pass (no host work between the event pair)
    durs = []
    for t in range(64):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        vec.step_device(rollout.cur["actions"] if fused else pool[t % len(pool)])
        b.record()
        durs.append((a, b))
    torch.cuda.synchronize(dev)
    kern_ms = sorted(a.elapsed_time(b) for a, b in durs)
    kern_ms = sum(kern_ms[8:-8]) / len(kern_ms[8:-8])
    region_ms = ev0.elapsed_time(ev1) / args.steps

    if rank == 0:
        total_envs = n_envs * world
        value = total_envs * args.steps / wall
        # dominant kernel = k_step, one launch per step: its average duration is the HIP-event time of the timed region
        # divided by the number of launches (back-to-back launches on one stream; includes the 1-in-128 reduction
        # syncs).  `kernel_ms_isolated` (event pair around single launches, separate pass) is reported for comparison.
        alg = ALG_BYTES[args.workload] * n_envs
        achieved = alg / (region_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get(args.workload + ("_log" if log_rows else ""))
        out = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "envs_per_gpu": n_envs, "total_envs": total_envs,
                       "rk4_substeps": int(vec._c.n_substeps), "actuator_microsteps": int(vec._c.actuator_microsteps),
                       "specialised_kernel": vec.spec_index >= 0, "derived_views": False,
                       "obs_layout": "row log [obs_step][{}][N][12] + zero-copy window".format(log_rows) if log_rows else "dense batch",
                       "rollout_head": ("HIP VecNormalize + 64-64 MlpPolicy (bf16 MFMA, split operands) + sampling, "
                                        "hipGraph chunks of {} steps".format(rollout.n_steps)) if fused else None,
                       "success_allgather_every": REDUCE_EVERY},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic["bytes_per_launch"] if traffic else None,
                         "traffic_source": traffic["source"] if traffic else None,
                         "kernel": "k_step + k_actor_act" if fused else "k_step", "kernel_ms": region_ms, "kernel_ms_isolated": kern_ms,
                         "algorithmic_bytes_per_env_step": ALG_BYTES[args.workload]},
        }
        if args.workload == "c3" and not log_rows and world == 1:
            # side measurement (never `value`): the same workload with row-log observations, the opt-in layout that keeps
            # the observation history once and hands out a zero-copy window
            alt = FixedWingVecEnv(cfg, num_envs=n_envs, device=local, config_kw=ckw, sim_config_kw=skw, seed=0,
                                  env_id_base=rank * n_envs, auto_reset=True, derived_views=False,
                                  obs_log_rows=_presets.OBS_LOG_ROWS)
            alt.reset()
            for t in range(200):
                alt.step_device(pool[t % len(pool)])
            torch.cuda.synchronize(dev)
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            for t in range(1000):
                alt.step_device(pool[t % len(pool)])
            a1.record()
            torch.cuda.synchronize(dev)
            alt_ms = a0.elapsed_time(a1) / 1000
            alt.close()
            out["row_log_observations"] = {"ms_per_step": alt_ms, "value": n_envs / alt_ms * 1e3, "unit": "env-steps/s",
                                           "roofline_frac": alg / (alt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           "note": "opt-in obs_log_rows=32 (bench.py --obs-layout log); same observation values "
                                                   "as a strided window, 697 B/env-step moved instead of 1 090"}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_seconds)
    vec.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:   # last, so that library banners (RCCL prints one on teardown) do not follow the result line
        sys.stdout.flush()
        try:   # RCCL's banner sits in the C stdio buffer until exit when stdout is a pipe: push it out first
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
