"""bench.py contract on the GPU box: one JSON line with the required keys; also through torch.distributed.run with the
RCCL path forced on (one rank), so that init / barrier / all_gather_into_tensor / max-reduce are exercised."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "50",
                          "--cpu-seconds", "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _last_json(out.stdout)
    assert KEYS <= set(j) and j["n_gpus"] == 1 and j["steps"] == 300 and j["scaling"] == "weak" and j["dtype"] == "f32"
    assert j["value"] > 1e9 and 0.2 < j["roofline"]["frac"] < 1.0 and j["roofline"]["bound"] == "hbm"
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0 and j["config"]["specialised_kernel"]


def test_bench_through_torchrun_with_rccl():
    env = dict(os.environ, FWG_BENCH_FORCE_DIST="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "300", "--warmup", "50", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    j = _last_json(out.stdout)
    assert j["n_gpus"] == 1 and j["value"] > 1e9


def test_bench_line_carries_the_rollout_and_training_figures():
    """BASELINE configs[4] (the PPO rollout loop) rides in the default line: `c5` = the library's default two-launch path, `c5_fused`
    (+ `c5_fused_bf16`) = the opt-in one-launch step, `c5_train` = with advantages and the optimiser; `timed_region` splits the
    driver's line into kernel time per step and fixed cost per chunk."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _last_json(out.stdout)
    for k in ("c5", "c5_fused", "c5_fused_bf16", "c5_train", "steady_state", "obs_delivered", "c2", "timed_region"):
        assert k in j and "error" not in j[k], (k, j.get(k))
    assert j["c5"]["value"] > 1.5e9 and j["c5"]["launches_per_step"] == 2 and j["c5_fused"]["launches_per_step"] == 1
    assert j["c5_fused"]["value"] > j["c5"]["value"] and j["rccl_ranks"] is None
    assert j["c5_train"]["value"] > 1e6 and j["c5_train"]["rollout_ms_per_update"] < 0.1 * j["c5_train"]["ms_per_update"]
    tr = j["timed_region"]
    assert 5.0 < tr["kernel_us_per_step"] < 20.0 and 0.0 < tr["fixed_us_per_chunk"] < 200.0
    # the zero-copy window costs what the step costs (a gathered copy would be 1.5x)
    assert j["obs_delivered"]["route"] == "row_log_view" and j["obs_delivered"]["ms_per_step"] < 1.15 * j["steady_state"]["ms_per_step"]


def test_two_rank_rccl_smoke():
    """Two ranks, one GPU each, RCCL all-gather of the success vector: only where the box has two GPUs (the driver's 8-GPU node);
    skips cleanly on the single-GPU boxes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "256", "--warmup", "20",
                          "--no-cpu-baseline", "--no-side"], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr[-3000:]
    j = _last_json(out.stdout)
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["config"]["total_envs"] == 2 * 65536 and j["value"] > 2e9
