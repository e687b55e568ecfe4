"""CPU baseline #2 for bench.py's cpu_baseline leg (TEST INFRASTRUCTURE, never the product path): the product's OWN kernel
source (fixed-wing-gym_amd/csrc/fwgym.hip) compiled for the host with g++ -O2 against the HIP shim of tests/emu, the
workgroups spread over OpenMP threads ("the build's own C++ backend with the same semantics").
A lane is a fiber and every wave-wide vote is a rendezvous of 64 fibers, so this is a LOWER bound on what a dedicated
scalar C++ implementation would reach; it is pinned to the oracle by the whole CPU test suite (tests/test_emu_parity.py,
tests/test_golden.py run the same build)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _measure(args):
    wl, seconds, n_envs = args
    for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["FWGYM_SPLIT"] = "0"   # one wave per 64 envs: fewer rendezvous in the emulation
    import numpy as np
    from emu.host_backend import HostBackend, build_emu_omp
    from gym_fixed_wing import presets
    from gym_fixed_wing.vec_env import FixedWingVecEnv
    cfg, ckw, skw, _, _ = presets.workload(wl)
    vec = FixedWingVecEnv(cfg, num_envs=n_envs, config_kw=ckw, sim_config_kw=skw, seed=0, as_numpy=True, derived_views=False,
                          _backend=HostBackend(), _lib_path=build_emu_omp())
    vec.reset()
    rng = np.random.default_rng(0)
    acts = [rng.uniform(-1, 1, (n_envs, 3)).astype(np.float32) for _ in range(4)]
    for t in range(2):
        vec.step_device(acts[t], want_obs=False)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        vec.step_device(acts[n % 4], want_obs=False)
        n += 1
    dt = time.perf_counter() - t0
    vec.close()
    return n * n_envs / dt, n


def measure(wl, seconds=8.0, n_envs=0, cores=0):
    """cores: OpenMP threads to use -- the CPUs the process really gets (bench._usable_cpus: affinity mask, container quota),
    not os.cpu_count(), which reports the machine (256 threads on a pod limited to 16 CPUs measure time-slicing)."""
    import multiprocessing as mp
    present = os.cpu_count() or 1
    cores = present if cores <= 0 else min(int(cores), present)
    if n_envs <= 0:   # one 64-env workgroup is the unit of OpenMP work: at least one per core, a few per core on small hosts
        n_envs = 64 * max(cores, 64)
    saved = os.environ.get("OMP_NUM_THREADS")
    os.environ["OMP_NUM_THREADS"] = str(cores)   # (inherited by the spawned process, read by its OpenMP runtime at start-up)
    try:
        with mp.get_context("spawn").Pool(1) as pool:   # a fresh process: OpenMP runtime and emulation state stay out of the bench
            rate, steps = pool.map(_measure, [(wl, seconds, n_envs)])[0]
    finally:
        if saved is None:
            os.environ.pop("OMP_NUM_THREADS", None)
        else:
            os.environ["OMP_NUM_THREADS"] = saved
    return {"value": rate, "unit": "env-steps/s", "cores": cores, "cores_present": present, "kind": "port",
            "sample": "product kernel source compiled for the host (g++ -O2, lock-step lane emulation, OpenMP over workgroups), "
                      "{} envs x {} steps in {:.0f} s".format(n_envs, steps, seconds)}
