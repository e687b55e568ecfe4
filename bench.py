#!/usr/bin/env python3
"""bench.py -- headline benchmark of the FixedWingAircraft.step() hot path on MI355X.

One "step" = one fused fwg_step launch advancing every env of this rank by one control step (dt = 0.01 s) including
observation, reward, done, metrics and auto-reset, on synthetic raw actions already resident in HBM.

Workloads (BASELINE.json configs): c3 (default; configs[2], the configuration the north-star target is quoted on):
65 536 envs per GPU, Dryden turbulence on ("moderate"), observation matrix 5 x 12 with lag step 2.  c2 = configs[1]
(4 096 envs, turbulence off, 14-vector).  c4 = configs[3]: the c3 settings at 32 768 envs per GPU (262 144 over 8 GPUs).
c5 = configs[4]: the PPO rollout loop (env step + HIP rollout head).  --total-envs T shards a FIXED total over the ranks
(strong scaling; `--total-envs 65536 --gpus 8` is the north-star headline).

Multi-GPU: one process per GPU.  `python bench.py --gpus N` launches its own ranks through torch.distributed.run when it
is not already running under one (before anything touches the GPU; the parent only waits and relays).  Envs are sharded
per rank with global env ids, the only collective is the RCCL all-gather of the 16-float success-metric vector
(examples/train_rl_controller.py:51-66,80-85 in the reference), once per replayed chunk of steps.

The timed region replays the K launches from hipGraphs (no per-launch host work), bracketed by barrier + synchronize;
`value` and `roofline` are both computed from that ONE wall-clock interval (max over ranks).  Prints ONE JSON line on
rank 0 with `roofline` and (N = 1) `cpu_baseline` objects.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
ALG_BYTES = {"c3": 937, "c4": 937, "c2": 417, "c5": 409}   # algorithmic bytes per env-step, SURVEY.md section 8(d)
MAX_CHUNK = 256                # steps per captured graph; the success reduction runs once per chunk
STEADY_STATE_STEPS = 300       # untimed launches before the W warm-up steps (clocks / caches / code objects), disclosed in the line
SIDE_STEPS = 512               # timed steps of every side measurement (steady state, observation consumer, other configurations)
SIDE_CHUNK = 128               # ... replayed as hipGraphs of one rollout each (success reduction / all-gather once per rollout)


def workload(name):
    from gym_fixed_wing import presets
    if name == "c4":
        cfg, ckw, skw, _, _ = presets.workload("c3")
        return cfg, ckw, skw, 32768, "C4: 32768 envs/GPU (262144 over 8 GPUs), Dryden turbulence moderate, obs 5x12 lag step 2, auto-reset, metrics on"
    return presets.workload(name)


def default_total_envs(world, workload_name, envs, total_envs, emulate):
    """TOTAL envs of the default multi-GPU run (0 = per-GPU sizing): 65 536 sharded over the ranks for c3 on N > 1 GPUs -- the
    north-star point of BASELINE.json -- unless a size was given."""
    if world > 1 and workload_name == "c3" and not envs and not total_envs and not emulate:
        return 65536
    return 0


def source_hash():
    """Identity of the kernel sources a profile / traffic figure belongs to."""
    h = hashlib.sha256()
    base = os.path.join(ROOT, "fixed-wing-gym_amd", "csrc")
    for fn in sorted(os.listdir(base)):
        fp = os.path.join(base, fn)
        if os.path.isfile(fp):
            with open(fp, "rb") as f:
                h.update(fn.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


# ----------------------------------------------------------------------------------------------------------------------
# CPU baselines (reported next to the GPU number, never the thing measured)
# ----------------------------------------------------------------------------------------------------------------------
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


def _usable_cpus(present):
    """Host threads this process may actually keep busy: the affinity mask, capped by the container's CPU quota (cgroup v2
    cpu.max / v1 cfs_quota_us).  os.cpu_count() reports the machine; a pod limited to a dozen CPUs that starts one worker per
    reported core measures time-slicing, not the host (round 3/4: 31 env-steps/s per process at "256 cores" against 750 for a
    process by itself -- a total of ten single-process rates)."""
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = present
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        usable = max(1, min(usable, int(quota)))
    return usable, quota


def _spin_worker(seconds):
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(2000):
            n += 1
    return n / (time.perf_counter() - t0)


def _effective_parallelism(ctx, usable, seconds=1.5):
    """How many processes' worth of interpreter work the host really delivers at once (a pure-Python spin loop, `usable`
    processes against one): catches CPU limits that no file shows (shares, a throttled pod, SMT siblings counted as cores)."""
    with ctx.Pool(1) as pool:
        one = pool.map(_spin_worker, [seconds])[0]
    with ctx.Pool(usable) as pool:
        agg = sum(pool.map(_spin_worker, [seconds] * usable))
    return max(1.0, agg / one)


def cpu_baseline(wl, seconds, max_procs=0):
    import multiprocessing as mp
    present = os.cpu_count() or 1
    usable, quota = _usable_cpus(present)
    cores = max(1, usable if max_procs <= 0 else min(usable, max_procs))
    try:   # one interpreter + NumPy/SciPy per process: stay well inside the host's free memory
        import psutil
        cores = max(1, min(cores, int(psutil.virtual_memory().available // (512 << 20))))
    except Exception:
        pass
    # one process per core means ONE thread per process: NumPy / SciPy would otherwise start a BLAS / OpenMP pool of
    # `present` threads in every worker (256 x 256 threads fighting over 256 cores: 26 env-steps/s per process in round 3
    # against 300 for a process by itself).  Spawned workers inherit the parent's environment, so it is pinned here,
    # before the pool exists
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS")}
    for k in saved:
        os.environ[k] = "1"
    effective = None
    try:
        ctx = mp.get_context("spawn")
        if cores > 16:   # one worker per core the host really gives us, not per core it lists
            try:
                effective = _effective_parallelism(ctx, cores)
                cores = max(1, min(cores, int(round(effective))))
            except Exception:
                effective = None
        with ctx.Pool(1) as pool:   # the single-process rate next to it: shows whether the all-core figure is oversubscribed
            n1, dt1 = pool.map(_cpu_worker, [(wl, min(seconds, 4.0), 99)])[0]
        with ctx.Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(wl, seconds, 100 + i) for i in range(cores)])
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    total = sum(n / dt for n, dt in res)
    out = {"value": total, "unit": "env-steps/s", "cores": cores, "cores_present": present, "cpu_quota": quota, "effective_parallelism": effective, "kind": "port",
           "per_process": total / cores, "single_process": n1 / dt1, "threads_per_process": 1,
           "sample": "{} oracle processes (one per host core the process can really keep busy -- affinity mask, container quota, a "
                     "measured spin-loop calibration --, one thread each; float64 NumPy restatement, 1 env each, same workload "
                     "config) x {:.0f} s, {} env-steps in total; a process by itself: {:.0f} env-steps/s".format(
                         cores, seconds, sum(n for n, _ in res), n1 / dt1)}
    try:   # second CPU number: the product kernel source compiled for the host (tests/emu, lock-step lane
        # emulation, OpenMP over workgroups) through oracle/cpu_native.py, when built
        from oracle import cpu_native
        nat = cpu_native.measure(wl, min(seconds, 8.0), cores=cores)
        if nat is not None:
            out["native"] = nat
    except Exception as e:   # the baseline is optional reporting: never fail the bench line over it
        out["native"] = {"error": str(e)[:200]}
    return out


# ----------------------------------------------------------------------------------------------------------------------
def _busy(dev, seconds=0.03):
    """~30 ms of unrelated device work (an elementwise pass over 64 MB, repeated) before a measurement.  The phases before one
    -- env construction, thousands of host-paced resets and single steps in stagger_ages -- leave the device mostly idle, and a
    box whose power management has clocked it down runs the first milliseconds of the timed region at half speed (one box in
    five: 23-30 us per step instead of 12.5 for the same launches).  Deliberately NOT more env steps: a fresh VecEnv's episodes
    run in lock-step, and a ramp of a few thousand steps would put the timed region onto the step where all of them end."""
    import torch
    x = torch.ones(1 << 24, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            x.mul_(1.0000001).add_(1e-9)
        torch.cuda.synchronize(dev)
    del x


def plan_chunks(k):
    """K timed steps -> (chunk length, replays, single eager steps).  Captured chunks hold an even number of steps."""
    even = k - (k % 2)
    if even == 0:
        return 0, 0, k
    c = min(even, MAX_CHUNK)
    c -= c % 2
    while even % c:
        c -= 2
    return c, even // c, k % 2


def spawn_ranks(args, argv):
    """--gpus N without a launcher: start N ranks (one per GPU) before this process touches the GPU; relay rank 0's line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c4", "c5"])
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (default: the workload's)")
    ap.add_argument("--total-envs", type=int, default=0, help="fixed TOTAL number of envs sharded over the ranks (strong scaling)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-procs", type=int, default=0, help="oracle processes of the CPU baseline (default 0 = one per host core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--obs-layout", default="auto", choices=["auto", "dense", "log"],
                    help="'auto' (default) = the env's default: lagged matrix observations are kept once as a row log and "
                         "handed out as a zero-copy window / read in place by the rollout head; 'dense' = fwg_step writes "
                         "the [N][5][12] batch every step")
    ap.add_argument("--rollout", default="auto", choices=["auto", "none", "fused", "one_launch"],
                    help="c5 only: 'fused' (default for c5) = env step + HIP rollout head (VecNormalize + MlpPolicy + "
                         "sampling), TWO launches per rollout step (the library's default path); 'one_launch' = the same as ONE "
                         "launch per step (fwg_rollout_step, opt-in: see FusedRollout); 'none' = env step on stored actions")
    ap.add_argument("--head-precision", default="split", choices=["split", "bf16"],
                    help="c5: the rollout head's matrix products: 'split' (default) = every fp32 operand as bf16 hi + lo, three MFMA "
                         "products per tile (~1e-5 of a torch fp32 forward); 'bf16' = one plain bf16 product (~1e-2)")
    ap.add_argument("--eager", action="store_true", help="launch every step from the host instead of replaying hipGraphs")
    ap.add_argument("--eager-sync", action="store_true",
                    help="with --eager: drain the device after every step, so that each launch arrives at an idle GPU (profiling: a "
                         "tracer's per-dispatch duration then holds the kernel alone, not its wait in the queue behind its predecessor)")
    ap.add_argument("--lib", default=None, help="measurement builds (tools/ablate.py): path of an alternative libfwgym.so")
    ap.add_argument("--stagger", type=int, default=0,
                    help="S > 0: before the warm-up, reset 1/S of the envs every steps_max/S steps, so that episode ends (metrics, "
                         "success reduction, in-kernel auto-reset, early-episode observation padding) are spread evenly over the "
                         "timed steps -- the steady state of a long run.  Default 0: all envs start together as a fresh VecEnv "
                         "does (no episode ends inside a short timed region); the steady state is then measured as well and "
                         "reported as `steady_state` next to `value` (1 GPU, env-step workloads)")
    ap.add_argument("--no-side", "--no-steady-state", dest="no_side", action="store_true",
                    help="skip the side measurements (steady state, observation consumer, dense layout, c2, integrator 4x64)")
    ap.add_argument("--emulate", action="store_true",
                    help="TEST ONLY: host-emulation build of the kernels + gloo, tiny batch (exercises the launcher, the "
                         "sharding and the collective on a machine without GPUs; the numbers are not measurements)")
    ap.add_argument("--emulate-steps-max", type=int, default=0,
                    help="TEST ONLY (with --emulate): episode time limit, so that episodes end inside a dry run and the success "
                         "all-gather + curriculum rule have something to agree on")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus {} but launched with WORLD_SIZE {}".format(args.gpus, world), file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    from gym_fixed_wing import distributed as fd
    from gym_fixed_wing.vec_env import FixedWingVecEnv

    use_dist = world > 1 or os.environ.get("FWG_BENCH_FORCE_DIST", "0") == "1"   # the latter: exercise RCCL on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.emulate:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))

    cfg, ckw, skw, n_envs, desc = workload(args.workload)
    scaling = "weak"
    if args.envs:
        n_envs = args.envs
    if args.emulate and not args.envs and not args.total_envs:
        n_envs = 128
    first = rank * n_envs
    # N > 1 GPUs, c3, no explicit size: `value` is the NORTH-STAR point of BASELINE.json -- 65 536 envs IN TOTAL sharded over the
    # ranks (strong scaling: the 1/2/4/8-GPU curve the driver assembles from the per-N lines is then the curve BASELINE.json asks
    # for, and it meets the N = 1 line at the same 65 536 envs); weak scaling at 65 536 envs per GPU and BASELINE configs[3]
    # (32 768 per GPU) ride along as side figures
    north_star_default = default_total_envs(world, args.workload, args.envs, args.total_envs, args.emulate) > 0
    if north_star_default:
        args.total_envs = default_total_envs(world, args.workload, args.envs, args.total_envs, args.emulate)
    if args.total_envs:
        first, n_envs = fd.shard(args.total_envs, rank, world)
        scaling = "strong"
        desc += " [{} envs in total, sharded]".format(args.total_envs)
    log_rows = {"auto": None, "dense": 0, "log": None}[args.obs_layout]
    if args.obs_layout == "log":
        from gym_fixed_wing import presets as _presets
        log_rows = _presets.OBS_LOG_ROWS
    kw = {}
    if args.emulate:
        from emu.host_backend import HostBackend, build_emu
        kw = {"_backend": HostBackend(), "_lib_path": build_emu()}
        dev = None
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        kw = {"device": local}
        if args.lib:
            kw["_lib_path"] = args.lib
    fused = args.workload == "c5" and args.rollout != "none"
    graphs = not (args.eager or args.emulate)
    seen = {"episodes": 0.0, "ranks": world}

    class Runner(object):
        """One env batch of this rank + its action pool, captured launch sequences and success reduction."""

        def __init__(self, cfg, ckw, skw, n, first, log_rows, seed=0, extra=None):
            # derived_views=False: the rollout loop never reads roll/pitch/... back from the arena (they are in the
            # observations), so the kernel does not write those host-view rows
            self.vec = FixedWingVecEnv(cfg, num_envs=n, config_kw=ckw, sim_config_kw=skw, seed=seed, env_id_base=first,
                                       auto_reset=True, derived_views=False, obs_log_rows=log_rows, **dict(kw, **(extra or {})))
            self.vec.reset()
            self.n = n
            self.graphs = {}
            if args.emulate:
                import numpy as np
                rng = np.random.default_rng(1234 + rank)
                self.pool = [rng.uniform(-1, 1, (n, 3)).astype(np.float32) for _ in range(8)]
                self.red_dev, self.gathered = None, None
            else:
                gen = torch.Generator(device=dev)
                gen.manual_seed(1234 + rank)
                self.pool = [torch.rand((n, 3), device=dev, generator=gen) * 2 - 1 for _ in range(32)]
                self.red_dev = torch.zeros(16, device=dev)
                self.gathered = torch.zeros(16 * world, device=dev) if use_dist else None
            self.graph_mode = False

        def enable_graphs(self):
            vec = self.vec
            vec.set_graph_mode(True)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):            # lazy initialisation outside capture
                vec.step_device(self.pool[0], want_obs=False), vec.step_device(self.pool[1], want_obs=False)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph_mode = True

        def stagger_ages(self, parts):
            """Resets 1/parts of the envs (a random subset: episode ends of a training run are not aligned with the env
            index) every steps_max/parts steps (parts = steps_max: ages uniform over [0, steps_max)).  An even number of
            steps in total, so that captured launch sequences keep their parity."""
            import numpy as _np
            vec = self.vec
            per = max(1, int(vec.cfg["steps_max"]) // parts)
            perm = _np.random.RandomState(4321 + rank).permutation(self.n)
            for k in range(parts):
                vec.reset(indices=_np.sort(perm[k::parts]))
                for t in range(per):
                    vec.step_device(self.pool[t % 4], want_obs=False)
            if (parts * per) % 2:
                vec.step_device(self.pool[0], want_obs=False)
            torch.cuda.synchronize(dev)

        def step_graph(self, n, offset, want_obs):
            """hipGraph of n consecutive fwg_step launches on pool actions (n even) + the chunk's success sums.  A captured
            sequence has the parity of the step counter baked in (double-buffered ring positions): one graph per parity."""
            vec = self.vec
            key = (n, offset % len(self.pool), vec.global_step & 1, bool(want_obs), getattr(vec, "_graph_obs", "gather"))
            if key not in self.graphs:
                g = torch.cuda.CUDAGraph()
                parity = vec.capture_begin(n if want_obs else None)
                with torch.cuda.graph(g):
                    for t in range(n):
                        vec.step_device(self.pool[(offset + t) % len(self.pool)], want_obs=want_obs)
                    vec.reduce_success_device(self.red_dev)   # device to device, part of the graph
                vec.capture_end()
                self.graphs[key] = (g, parity)
            return self.graphs[key]

        def reduce_step(self, in_graph):
            vec = self.vec
            if args.emulate:
                local_sums = torch.as_tensor(vec.reduce_success(), dtype=torch.float32)
                if use_dist:
                    out = torch.empty(16 * world)
                    dist.all_gather_into_tensor(out, local_sums)
                    seen["allgathers"] = seen.get("allgathers", 0) + 1
                    total = out.view(world, 16).sum(dim=0).numpy().astype("float64")
                else:
                    total = local_sums.numpy().astype("float64")
                seen["episodes"] += float(total[0])
                # the curriculum rule on the GLOBAL sums, on every rank (examples/train_rl_controller.py:80-87)
                seen["level"] = sched.update(vec, fd.summarize(total, vec.target_names))
                return
            if not in_graph:
                vec.reduce_success_device(self.red_dev)     # local sums, device to device, stream-ordered (no host sync)
            if use_dist:
                dist.all_gather_into_tensor(self.gathered, self.red_dev)   # RCCL over xGMI: 64 B per rank

        def precheck(self, c, offset=0, want_obs=False):
            """Looks the chunk's graph up and validates the replay (step parity, window phase, parameter generation) BEFORE a timed
            region: the first replay of the following run() then goes straight to the launch -- what is timed is the launches, not
            the host-side checks in front of them."""
            if c:
                g, parity = self.step_graph(c, offset, want_obs)
                self.vec.replay_check(parity)
                self._prechecked = (g, c, offset, bool(want_obs))

        def run(self, c, r, s, offset=0, want_obs=False, rollout=None):
            """r replays of a c-step chunk (+ one success reduction / all-gather each), then s single launches."""
            vec = self.vec
            done_steps = 0
            for _ in range(r):
                if rollout is not None:
                    rollout(c).run()
                    self.reduce_step(False)
                else:
                    pre = getattr(self, "_prechecked", None)
                    self._prechecked = None
                    if pre is not None and pre[1:] == (c, offset, bool(want_obs)):
                        g = pre[0]
                    else:
                        g, parity = self.step_graph(c, offset, want_obs)
                        vec.replay_check(parity)
                    g.replay()
                    vec.note_replayed_steps(c)
                    self.reduce_step(True)
                done_steps += c
            for t in range(s):
                vec.step_device(self.pool[(offset + done_steps + t) % len(self.pool)], want_obs=want_obs)
                if args.eager_sync:
                    torch.cuda.synchronize(dev)
            if s and not r:
                self.reduce_step(False)

        def time_replays(self, c, reps, want_obs=False):
            """ms per step of reps replays of a c-step chunk (captured beforehand), wall clock around a drained device."""
            _busy(dev)   # (clocks up: see _busy)
            self.run(c, 1, 0, 0, want_obs)
            torch.cuda.synchronize(dev)
            self.precheck(c, 0, want_obs)
            t0 = time.perf_counter()
            self.run(c, reps, 0, 0, want_obs)
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / (reps * c) * 1e3

    if args.emulate and args.emulate_steps_max:
        ckw = dict(ckw or {}, steps_max=args.emulate_steps_max)
    sched = fd.CurriculumSchedule(level=0.25, cooldown=0)
    R = Runner(cfg, ckw, skw, n_envs, first, log_rows)
    vec = R.vec
    if args.stagger and not args.emulate:
        R.stagger_ages(args.stagger)
    chunk, replays, singles = plan_chunks(args.steps) if graphs else (0, 0, args.steps)
    # (fused rollout: an odd warm-up is rounded up by one untimed step -- its chunks keep the parity of the step count)
    warm_eff = args.warmup + 1 if (args.workload == "c5" and args.rollout != "none" and args.warmup % 2) else args.warmup
    wchunk, wreplays, wsingles = plan_chunks(warm_eff) if graphs else (0, 0, warm_eff)
    get_rollout = None
    if fused:   # BASELINE configs[4]: PPO rollout loop with a random-init 64-64 MlpPolicy, end to end
        from gym_fixed_wing.actor import DeviceActor
        from gym_fixed_wing.rollout import FusedRollout, MlpPolicy
        if args.steps % 2:
            raise SystemExit("--workload c5: --steps must be even (hipGraph chunks of the fused rollout keep the parity of the step count)")
        torch.manual_seed(0)
        actor = DeviceActor.for_env(vec, seed=7, env_id_base=first, precise=args.head_precision == "split")
        actor.load_policy(MlpPolicy(vec.obs_dim))
        rollouts = {}

        def get_rollout(n):
            if n not in rollouts:
                rollouts[n] = FusedRollout(vec, actor, n, graph=True, fused="auto" if args.rollout == "one_launch" else False)
            return rollouts[n]

    if graphs and not fused:
        R.enable_graphs()

    # untimed: bring the device to its steady state (power state, caches, code objects, graph instantiation), then the
    # W warm-up steps of the contract.  The untimed steps add up to an EVEN count, so that the timed region's launch
    # sequences are the ones captured here (a captured sequence is tied to the parity of the step counter)
    extra = 0
    if graphs:
        pc = chunk if chunk else 2
        _busy(dev)   # (clocks up before anything is timed or warmed up: see _busy)
        while extra < STEADY_STATE_STEPS:
            R.run(pc, 1, 0, rollout=get_rollout)
            extra += pc
    R.run(wchunk, wreplays, wsingles, rollout=get_rollout)
    if graphs and (args.warmup % 2) and not fused:   # (one more single: the odd warm-up's last step flipped the parity)
        R.run(0, 0, 1)
        extra += 1
    if not args.emulate:
        torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    if not args.emulate:
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if graphs and not fused and replays:
        R.precheck(chunk)      # (host-side validation of the replay, outside the clock)
    if not args.emulate:
        ev0.record()           # (GPU-side bracket for kernel_ms_hip_events; enqueued on the idle, synchronised stream)
    t0 = time.perf_counter()
    R.run(chunk, replays, singles, rollout=get_rollout)
    if not args.emulate:
        ev1.record()
        torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    if not args.emulate:
        torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([wall], dtype=torch.float64, device=dev if not args.emulate else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())
    event_ms = ev0.elapsed_time(ev1) / args.steps if not args.emulate else None

    # ---- side measurements (same launches, other regimes / consumers / configurations); never part of `value`
    sides = {}
    side_ok = graphs and not fused and not args.emulate and not args.no_side
    if side_ok and replays and chunk != SIDE_CHUNK:
        # What the timed region is made of: T(c) = c x k + F per replayed chunk (k: one step kernel in a replayed graph; F: the
        # chunk's fixed cost -- graph launch from an idle device, episode collection + success sums, host launch + wake-up).
        # Two chunk lengths on the same fresh env give both: the timed region's own (c steps) and a 128-step rollout.
        t_long = R.time_replays(SIDE_CHUNK, 4) * 1e-3 * SIDE_CHUNK           # seconds per 128-step chunk
        t_short = wall / max(replays, 1) if not singles else None
        if t_short:
            k_us = (t_long - t_short) / (SIDE_CHUNK - chunk) * 1e6
            sides["timed_region"] = {"chunk_steps": chunk, "us_per_chunk": t_short * 1e6, "kernel_us_per_step": k_us,
                                     "fixed_us_per_chunk": t_short * 1e6 - chunk * k_us, "us_per_step_at_128_step_chunks": t_long / SIDE_CHUNK * 1e6,
                                     "note": "T(c) = c k + F from two chunk lengths on the same fresh env: F = launching one hipGraph on an idle "
                                             "device + the chunk's episode collection and success sums (one launch since round 6) + host launch and "
                                             "wake-up; no host wait policy moves it (profiles/r06_sync_policy.txt)"}
    # side measurements replay chunks of one PPO rollout (128 steps, the n_steps of the reference's shipped models and the
    # all-gather interval of BASELINE configs[3]) whatever --steps is: a 20-step chunk would put the per-chunk costs -- graph
    # launch, episode collection, success sums, all-gather -- on every 20th step, which no training run does
    sc = SIDE_CHUNK
    sreps = max(1, (SIDE_STEPS + sc - 1) // sc)
    alg_b = ALG_BYTES[args.workload]

    def side_entry(ms, n, alg=alg_b, **more):
        e = {"ms_per_step": ms, "value": n / (ms * 1e-3), "unit": "env-steps/s", "roofline_frac": alg * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "steps": sreps * sc, "chunk": sc}
        e.update(more)
        return e

    if side_ok and not args.stagger:
        # the same launches with episode ages spread uniformly (every step then has waves that end episodes, reset envs in
        # the kernel and pad early-episode observation rows): what a long run sees
        R.stagger_ages(int(vec.cfg["steps_max"]))
        per_step = round(n_envs / max(1, int(vec.cfg["steps_max"])))
        sides["steady_state"] = side_entry(R.time_replays(sc, sreps), n_envs,
            note="episode ages uniform over [0, steps_max) (a random 1/steps_max of the envs reset at every step of an untimed "
                 "steps_max-step run): about {} episode ends per step, scattered over the waves".format(per_step))
        if vec.obs_log_rows:
            # the observation in a consumer's hands on EVERY step of a replayed graph, steady state.  Route of round 5: the
            # zero-copy window of the row log -- step_device(want_obs=True) hands out a strided [N][length][n_obs] view per
            # captured step; the views stay right under replay because the chunk is a whole number of window periods
            # (FixedWingVecEnv.set_graph_mode(obs="view"), tests/test_obs_log.py).  No copy, no extra launch: what the consumer
            # then reads is its own traffic, as with any layout.
            per = vec.obs_window_period
            if per and sc % per == 0:
                vec.set_graph_mode(True, obs="view")   # (opt-in: the library default hands out gathered copies)
                sides["obs_delivered"] = side_entry(R.time_replays(sc, sreps, want_obs=True), n_envs, route="row_log_view",
                    window_period_steps=per,
                    note="steady state with the zero-copy observation window handed out on every step of the replayed graph (views "
                         "computed at capture, valid under replay: the {}-step chunk is {} window periods).  Rounds 3-4 published "
                         "the cheaper of a dense copy (obs_layout='dense') and row log + gather under this key: now "
                         "`obs_delivered_copy`".format(sc, sc // per))
            # a dense COPY for a consumer that cannot take a strided view: fwg_obs_gather after every step inside the graph
            vec.set_graph_mode(True, obs="gather")
            sides["obs_gather_row_log"] = side_entry(R.time_replays(sc, sreps, want_obs=True), n_envs,
                note="steady state + fwg_obs_gather after every step inside the replayed graph (set_graph_mode(obs='gather')): a "
                     "dense copy of the observation batch out of a ROW-LOG env.  A consumer that needs a dense batch every step is "
                     "served cheaper by the dense layout (obs_layout='dense'): `dense_layout`, `obs_delivered_copy`")
            vec.set_graph_mode(True, obs="view")
    if side_ok and world == 1 and args.workload == "c3" and not args.stagger and not args.envs and not args.total_envs:
        def side_env(name, wl_cfg, n, rows, alg, note, stag=True, extra=None):
            try:
                S = Runner(wl_cfg[0], wl_cfg[1], wl_cfg[2], n, 0, rows, extra=extra)
                S.enable_graphs()
                S.run(sc, max(1, STEADY_STATE_STEPS // sc), 0)
                fresh = S.time_replays(sc, sreps)
                ent = side_entry(fresh, n, alg, note=note, specialised_kernel=S.vec.spec_index >= 0)
                ent["kernel_instance"] = int(S.vec.spec_index)   # (-1 generic | i frozen configuration | 1000 + i its shape instance)
                if stag:
                    S.stagger_ages(int(S.vec.cfg["steps_max"]))
                    st = S.time_replays(sc, sreps)
                    ent["steady_state_ms_per_step"] = st
                    ent["steady_state_roofline_frac"] = alg * n / (st * 1e-3) / 1e9 / HBM_PEAK_GBS
                S.vec.close()
                sides[name] = ent
            except Exception as e:   # side figures never fail the line
                sides[name] = {"error": str(e)[:300]}

        if vec.obs_log_rows:
            side_env("dense_layout", (cfg, ckw, skw), n_envs, 0, alg_b,
                     "the same workload with the dense [N][5][12] observation batch written by every step (obs_layout='dense')")
            # what delivering the dense batch to a torch consumer on EVERY step of a replayed graph costs, steady state: the
            # cheaper of the two ways the package offers (FixedWingVecEnv warns when graph mode is enabled on a row-log env
            # whose consumer takes the batch every step)
            d, g = sides.get("dense_layout", {}), sides.get("obs_gather_row_log", {})
            cands = [(d.get("steady_state_ms_per_step"), "dense layout (the step kernel writes the batch), steady state"),
                     (g.get("ms_per_step"), "row log + fwg_obs_gather after every step, steady state")]
            cands = [c_ for c_ in cands if c_[0]]
            if cands:
                ms, via = min(cands)
                sides["obs_delivered_copy"] = side_entry(ms, n_envs, via=via, row_log_gather_ms_per_step=g.get("ms_per_step"),
                                                         dense_layout_ms_per_step=d.get("steady_state_ms_per_step"),
                                                         note="a dense [N][5][12] COPY of the observation batch after every step, episode ages "
                                                              "uniform: the cheaper of the dense layout and row log + gather (rounds 3-4 "
                                                              "published this figure as `obs_delivered`)")
        c2 = workload("c2")
        side_env("c2", c2[:3], c2[3], None, ALG_BYTES["c2"], c2[4] + " (BASELINE configs[1]; 64 workgroups: launch-latency bound)", stag=False)
        import copy as _copy
        hi = _copy.deepcopy(skw)
        hi["integrator"] = {"method": "rk4", "substeps": 4, "actuator_microsteps": 64}
        side_env("integrator_4x64", (cfg, ckw, hi), n_envs, log_rows, alg_b,
                 "the same workload with 4 RK4 sub-steps and 64 exact actuator micro-steps per env step: the first scheme clearly "
                 "more accurate than the reference's adaptive RK45 at rtol 1e-3 (profiles/r02_convergence.json)", stag=False)
        # a configuration that differs from the preset in VALUES only (here: steps_max 1999; any scaling, constraint, aircraft
        # constant ... likewise) and no run-time compiler: the preset's SHAPE instance (structure frozen, values read from memory)
        gk = _copy.deepcopy(ckw or {})
        gk["steps_max"] = 1999
        import warnings as _warnings
        with _warnings.catch_warnings():
            _warnings.simplefilter("ignore")
            side_env("shape_instance", (cfg, gk, skw), n_envs, log_rows, alg_b,
                     "the same workload with steps_max 1999 -- no frozen configuration -- and specialize=False: the preset's SHAPE "
                     "instance (every count / type / flag frozen as in the preset's kernel, every value a scalar load from the "
                     "configuration in memory); what a configuration that differs from a preset in values only runs, nothing compiled "
                     "at run time", stag=False, extra={"specialize": False})
            # the GENERIC kernel (what is left when the STRUCTURE matches no preset either and no specialised kernel can be
            # compiled: no hipcc on the machine, or specialize=False): the same configuration with the shape instances switched off
            os.environ["FWGYM_SHAPE"] = "0"
            try:
                side_env("generic_kernel", (cfg, gk, skw), n_envs, log_rows, alg_b,
                         "the same configuration on the GENERIC kernel (configuration interpreted at run time: scalar loads, LDS tables, "
                         "scratch; FWGYM_SHAPE=0), what a configuration whose structure matches no preset runs without a run-time "
                         "specialised kernel (FixedWingVecEnv compiles one by default when hipcc is present, and warns either way)",
                         stag=False, extra={"specialize": False})
            finally:
                os.environ.pop("FWGYM_SHAPE", None)
        from gym_fixed_wing import presets as _pr
        try:   # every env flies its own aircraft (16 parameters re-sampled at every reset); gentle actions: random full-scale
            # actions crash the randomised aircraft within a few dozen steps and the run then measures failure ends
            S = Runner(_pr.preset("cnn_model16"), ckw, skw, n_envs, 0, log_rows)
            S.pool = [p * 0.15 for p in S.pool]
            S.enable_graphs()
            S.run(sc, max(1, STEADY_STATE_STEPS // sc), 0)
            ms = S.time_replays(sc, sreps)
            sides["randomised_aircraft"] = side_entry(ms, n_envs, alg_b + 208 + 32, specialised_kernel=S.vec.spec_index >= 0,
                note="the same workload with simulator.model: 16 aircraft parameters re-sampled per env at every reset (49 per-lane "
                     "force/moment constants: +208 B/env-step read); actions 0.15 x U(-1,1); the next sets are drawn by a 4-wave "
                     "launch working off the queue of the envs reset in the previous step")
            S.vec.close()
        except Exception as e:
            sides["randomised_aircraft"] = {"error": str(e)[:300]}
    if side_ok and world == 1 and args.workload == "c3" and not args.stagger and not args.envs and not args.total_envs:
        # BASELINE configs[4]: the PPO rollout loop (VecNormalize + 64-64 MlpPolicy + sampling + env step, end to end), 128-step
        # rollouts replayed as hipGraphs; head and env step in ONE launch per step where fwg_rollout_step applies
        def side_c5(name, precise, wl="c5", what="BASELINE configs[4]", one_launch=False):
            try:
                from gym_fixed_wing.actor import DeviceActor
                from gym_fixed_wing.rollout import FusedRollout, MlpPolicy
                c5 = workload(wl)
                S = Runner(c5[0], c5[1], c5[2], c5[3], 0, 0)
                torch.manual_seed(0)
                act = DeviceActor.for_env(S.vec, seed=7, env_id_base=0, precise=precise)
                act.load_policy(MlpPolicy(S.vec.obs_dim))
                ro = FusedRollout(S.vec, act, sc, graph=True, fused="auto" if one_launch else False)
                for _ in range(3):
                    ro.run()
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(sreps):
                    ro.run()
                    S.reduce_step(False)
                torch.cuda.synchronize(dev)
                ms = (time.perf_counter() - t1) / (sreps * sc) * 1e3
                sides[name] = side_entry(ms, c5[3], ALG_BYTES[wl], specialised_kernel=S.vec.spec_index >= 0,
                    launches_per_step=1 if ro.fused else 2,
                    note=c5[4] + " (" + what + "): rollout head (VecNormalize statistics + 64-64 MlpPolicy on bf16 MFMA, " +
                         ("operands split hi + lo: three products per tile, ~1e-5 of a torch fp32 forward" if precise else
                          "ONE plain bf16 product per tile, ~1e-2") + " + sampling) and env step, " +
                         ("ONE launch per rollout step (fwg_rollout_step)" if ro.fused else "two launches per rollout step") +
                         ", rollout buffers written in place, {}-step rollouts replayed as hipGraphs".format(sc))
                act.close()
                S.vec.close()
            except Exception as e:
                sides[name] = {"error": str(e)[:300]}

        # `c5` is the path the library runs by DEFAULT (two launches per rollout step: env step with the batch moments attached,
        # then the head).  The one-launch step (fwg_rollout_step) is opt-in -- an intermittent bit-mismatch against the two-launch
        # path seen twice in round 4's suite runs was never root-caused (22 000 clean iterations since, profiles/r05_soak.txt) --
        # and is published under its own keys until it is (rounds 4-5 published it as `c5`).
        side_c5("c5", True)
        side_c5("c5_fused", True, one_launch=True)
        side_c5("c5_fused_bf16", False, one_launch=True)
        # what the user of a SMALL batch runs: policy + env step at BASELINE configs[1]'s 4 096 envs
        side_c5("c2_rollout", True, "c2", "BASELINE configs[1] under the rollout loop: 16 workgroups of 256 envs, launch-latency bound")
        side_c5("c2_rollout_fused", True, "c2", "BASELINE configs[1] under the rollout loop, one launch per step", one_launch=True)
        # BASELINE configs[4] WITH the learner: rollouts + fwg_gae + the PPO2 update (gym_fixed_wing/ppo.py, the
        # recipe of examples/train_ppo.py: 4 epochs x 128 minibatches, graph-captured minibatch steps) -- env-steps/s of training
        try:
            from gym_fixed_wing.ppo import PPO
            c5 = workload("c5")
            S = Runner(c5[0], c5[1], c5[2], c5[3], 0, 0)
            ppo = PPO(S.vec, seed=0, nminibatches=128, learning_rate=5e-4)
            ppo.update(ppo.collect())
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            n_up = 2
            t_roll = 0.0
            for _ in range(n_up):
                t2 = time.perf_counter()
                batch = ppo.collect()
                torch.cuda.synchronize(dev)
                t_roll += time.perf_counter() - t2
                ppo.update(batch)
                S.reduce_step(False)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t1
            n_tr = n_up * ppo.n_steps * c5[3]
            sides["c5_train"] = {"value": n_tr / dt, "unit": "env-steps/s", "ms_per_update": dt / n_up * 1e3, "rollout_ms_per_update": t_roll / n_up * 1e3,
                                 "envs": c5[3], "n_steps": ppo.n_steps, "transitions_per_update": ppo.n_steps * c5[3],
                                 "optimiser_steps_per_update": int(ppo.hp["noptepochs"]) * int(ppo.hp["nminibatches"]),
                                 "note": c5[4] + ": PPO training, env-steps/s INCLUDING advantages (fwg_gae) and the optimiser (PPO2 objective, "
                                         "4 epochs x 128 minibatches of 65 536 transitions, each minibatch step one replayed hipGraph); the "
                                         "rollout itself is `rollout_ms_per_update` of `ms_per_update`"}
            S.vec.close()
        except Exception as e:
            sides["c5_train"] = {"error": str(e)[:300]}
    if side_ok and world > 1 and args.workload == "c3" and north_star_default:
        # multi-GPU side figures: BASELINE configs[3] (32 768 envs per GPU) and weak scaling at the one-GPU workload (65 536 per GPU)
        for name, n_side, first_side in (("c4_32768_per_gpu", 32768, rank * 32768), ("weak_65536_per_gpu", 65536, rank * 65536)):
            try:
                S = Runner(cfg, ckw, skw, n_side, first_side, log_rows)
                S.enable_graphs()
                S.run(sc, max(1, STEADY_STATE_STEPS // sc), 0)
                torch.cuda.synchronize(dev); dist.barrier()
                t1 = time.perf_counter()
                S.run(sc, sreps, 0)
                torch.cuda.synchronize(dev); dist.barrier()
                tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ms = float(tt.item()) / (sreps * sc) * 1e3
                sides[name] = {"ms_per_step": ms, "value": n_side * world / (ms * 1e-3), "scaling": "weak",
                               "unit": "env-steps/s", "envs_per_gpu": n_side, "steps": sreps * sc,
                               "rccl_ranks": dist.get_world_size(), "success_allgather_every": sc}
                S.vec.close()
            except Exception as e:
                sides[name] = {"error": str(e)[:300]}

    shards, levels = [[first, n_envs]], [seen.get("level", sched.level)]
    if args.emulate and use_dist:   # (dry run: every rank's shard and curriculum level, for tests/test_bench_cli.py)
        got = [None] * world
        dist.all_gather_object(got, (first, n_envs, seen.get("level", sched.level)))
        shards, levels = [[g[0], g[1]] for g in got], [g[2] for g in got]
    out = None
    if rank == 0:
        total_envs = args.total_envs if args.total_envs else n_envs * world
        value = total_envs * args.steps / wall
        # ONE clock: the roofline figure is the same wall interval as `value`, for the dominant kernel (k_step, one launch
        # per step; this rank's share of the envs).  kernel_ms_hip_events = HIP events around the same region, for reference.
        alg = ALG_BYTES[args.workload] * n_envs
        achieved = alg * args.steps / wall / 1e9
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        layout_log = bool(vec.obs_log_rows)
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            ent = tj.get(args.workload + ("_log" if layout_log else ""))
            if ent and ent.get("source_hash") == source_hash():   # only a profile of THIS build counts
                traffic, traffic_src = ent["bytes_per_launch"], ent["source"]
            elif ent:
                traffic_src = "profiles/traffic.json holds a figure for another build ({}): not reported".format(ent.get("source_hash"))
        out = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not args.emulate else "synthetic (HOST EMULATION: not a measurement)",
            # ranks that took part in the success-vector all-gather (RCCL; gloo under --emulate); null = no process group
            "rccl_ranks": (dist.get_world_size() if use_dist else None),
            "config": {"workload": desc, "envs_per_gpu": n_envs, "total_envs": total_envs, "ranks": seen["ranks"],
                       "rk4_substeps": int(vec._c.n_substeps), "actuator_microsteps": int(vec._c.actuator_microsteps),
                       "specialised_kernel": vec.spec_index >= 0, "derived_views": False,
                       "obs_layout": ("row log [obs_step][{}][N][12]: the observation is a zero-copy strided window of it "
                                      "(FixedWingVecEnv default for lagged observations)".format(vec.obs_log_rows)) if layout_log else "dense batch",
                       "launch": ("hipGraph replay: {} x {} steps + {} single".format(replays, chunk, singles)) if graphs else "one host launch per step",
                       "steady_state_steps_before_warmup": extra, "staggered_episode_ages": args.stagger,
                       "rollout_head": ("HIP VecNormalize + 64-64 MlpPolicy (bf16 MFMA, {}) + sampling, hipGraph chunks of {} steps".format(
                                            "split operands: three products per tile" if args.head_precision == "split" else "ONE plain bf16 product per tile",
                                            chunk)) if fused else None,
                       "success_allgather_every": chunk if graphs and chunk else args.steps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         # what the fabric actually moved per second (measured bytes / the same interval): the row log neither
                         # re-reads nor re-writes the lagged rows, so this is BELOW `achieved` (algorithmic bytes)
                         "measured_GBs": (traffic * args.steps / wall / 1e9) if traffic else None,
                         "achieved_real": (traffic * args.steps / wall / 1e9) if traffic else None,   # (the same figure under the review's name)
                         "frac_measured": (traffic * args.steps / wall / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "kernel": ("k_rollout (head + env step in one launch)" if fused and getattr(get_rollout(chunk or 2), "fused", False) else
                                    ("k_step2" if vec.spec_index >= 0 and os.environ.get("FWGYM_SPLIT", "1") != "0" else "k_step") + (" + k_actor_act" if fused else "")),
                         "kernel_ms": wall / args.steps * 1e3, "kernel_ms_hip_events": event_ms,
                         "clock": "wall clock of the timed region (the same interval as `value`)",
                         "algorithmic_bytes_per_env_step": ALG_BYTES[args.workload], "source_hash": source_hash()},
        }
        for k, v in sides.items():
            if traffic and isinstance(v, dict) and "ms_per_step" in v and k in ("steady_state",):
                v["measured_GBs"] = traffic / (v["ms_per_step"] * 1e-3) / 1e9
            out[k] = v
        if args.emulate:
            out["emulated_episodes_seen"] = seen["episodes"]
            out["emulated_allgathers"] = seen.get("allgathers", 0)
            out["emulated_shards"] = shards
            out["emulated_curriculum_levels"] = levels
    vec.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not args.emulate:   # after the process group is gone
            out["cpu_baseline"] = cpu_baseline("c3" if args.workload == "c4" else args.workload, args.cpu_seconds, args.cpu_procs)
        sys.stdout.flush()
        try:   # RCCL's banner sits in the C stdio buffer until exit when stdout is a pipe: push it out first
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
