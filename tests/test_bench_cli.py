"""bench.py launches its own ranks for --gpus N (CPU dry run: host-emulation kernels + gloo; the GPU runs use the same
launcher, sharding and reduction code with RCCL) and prints ONE JSON line whose fields follow the driver contract."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, env=e, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_gpus_2_spawns_two_ranks_and_reports_them():
    p, out = _bench("--gpus", "2", "--emulate", "--steps", "6", "--warmup", "2", "--no-cpu-baseline")
    assert p.returncode == 0, p.stderr[-2000:]
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2
    assert out["config"]["total_envs"] == 2 * out["config"]["envs_per_gpu"] and out["scaling"] == "weak"
    assert out["steps"] == 6 and out["warmup"] == 2 and out["metric"] == "env-steps/sec"
    # one clock: value and the roofline figure describe the same interval
    frac = out["value"] / out["n_gpus"] * out["roofline"]["algorithmic_bytes_per_env_step"] / 1e9 / out["roofline"]["peak"]
    assert abs(frac - out["roofline"]["frac"]) <= 1e-9 + 1e-6 * frac
    assert abs(out["ms_per_step"] - out["roofline"]["kernel_ms"]) < 1e-12


def test_total_envs_is_strong_scaling():
    p, out = _bench("--gpus", "2", "--emulate", "--total-envs", "96", "--steps", "4", "--warmup", "0", "--no-cpu-baseline",
                    "--workload", "c2")
    assert p.returncode == 0, p.stderr[-2000:]
    assert out["scaling"] == "strong" and out["config"]["total_envs"] == 96 and out["config"]["envs_per_gpu"] == 48


def test_rank_count_mismatch_is_an_error():
    p, out = _bench("--gpus", "4", "--emulate", "--steps", "2", "--warmup", "0", "--no-cpu-baseline",
                    env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 2 and out is None


def test_chunk_plan():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.plan_chunks(20) == (20, 1, 0)
    assert bench.plan_chunks(2000) == (250, 8, 0)
    assert bench.plan_chunks(5) == (4, 1, 1)
    assert bench.plan_chunks(1) == (0, 0, 1)
    c, r, s = bench.plan_chunks(1022)
    assert c * r + s == 1022 and c % 2 == 0 and c <= bench.MAX_CHUNK


def test_cpu_baseline_is_sized_by_the_container_quota(monkeypatch):
    """The GPU box lists 256 hardware threads and gives the process 16 CPUs (cgroup v2 cpu.max "1600000 100000"): one oracle
    process per LISTED core measured time-slicing for three rounds (bench._usable_cpus)."""
    import builtins
    import io
    sys.path.insert(0, ROOT)
    import bench
    real_open = builtins.open

    def fake_open(path, *a, **kw):
        if path == "/sys/fs/cgroup/cpu.max":
            return io.StringIO(fake["cpu.max"])
        return real_open(path, *a, **kw)

    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    fake = {"cpu.max": "1600000 100000\n"}
    assert bench._usable_cpus(256) == (16, 16.0)
    fake["cpu.max"] = "max 100000\n"
    assert bench._usable_cpus(256) == (256, None)
    fake["cpu.max"] = "50000 100000\n"          # half a CPU: still one worker
    assert bench._usable_cpus(256)[0] == 1
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(8)), raising=False)
    fake["cpu.max"] = "1600000 100000\n"
    assert bench._usable_cpus(256)[0] == 8        # the affinity mask is the tighter bound


def test_multi_gpu_default_is_the_north_star_point():
    """bench.py --gpus N (N > 1) with no size given measures BASELINE.json's north-star point -- 65 536 envs in total, sharded --
    so that the driver's 1/2/4/8-GPU lines form the strong-scaling curve the baseline asks for; sizes given on the command line,
    other workloads and the one-GPU run keep per-GPU sizing."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.default_total_envs(8, "c3", 0, 0, False) == 65536
    assert bench.default_total_envs(2, "c3", 0, 0, False) == 65536
    assert bench.default_total_envs(1, "c3", 0, 0, False) == 0
    assert bench.default_total_envs(8, "c4", 0, 0, False) == 0
    assert bench.default_total_envs(8, "c3", 32768, 0, False) == 0
    assert bench.default_total_envs(8, "c3", 0, 131072, False) == 0
    assert bench.default_total_envs(2, "c3", 0, 0, True) == 0


def test_eight_ranks_of_configs3_dry_run():
    """BASELINE configs[3] as the driver would launch it on an 8-GPU node -- `bench.py --gpus 8 --workload c4` -- on the host
    emulation + gloo: 8 x 32 768 envs in contiguous shards keyed by global env ids, ONE success all-gather per replayed chunk
    (64 B per rank), and the curriculum rule applied to the global sums lands on the SAME level on every rank.  (No 8-GPU node
    has reached the driver in six rounds: this is the part of row (e) that can be checked without one.)"""
    p, out = _bench("--gpus", "8", "--emulate", "--workload", "c4", "--envs", "32768", "--emulate-steps-max", "3", "--steps", "4",
                    "--warmup", "2", "--no-cpu-baseline")
    assert p.returncode == 0, p.stderr[-2000:]
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["config"]["ranks"] == 8
    assert out["config"]["envs_per_gpu"] == 32768 and out["config"]["total_envs"] == 262144 and out["scaling"] == "weak"
    assert out["emulated_shards"] == [[r * 32768, 32768] for r in range(8)]
    # 2 warm-up steps + 4 timed steps = one all-gather per launch sequence here (the emulation runs eagerly: one reduction per run() call)
    assert out["emulated_allgathers"] >= 2
    assert out["emulated_episodes_seen"] >= 262144          # 3-step episodes: every env finished at least one, counted over ALL ranks
    lv = out["emulated_curriculum_levels"]
    assert len(lv) == 8 and len(set(lv)) == 1, lv
