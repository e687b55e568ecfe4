#!/usr/bin/env python3
"""Per-wave phase timeline of k_step from s_memtime stamps (measurement build, -DFWG_TIMELINE; never the product).

  python tools/timeline.py build            # here: cross-compiles gym_fixed_wing/_abl/libfwgym_timeline.so
  gpurun -- python tools/timeline.py run    # GPU box: C3 dense and row-log, prints the phase table + JSON

Stamps (csrc/fwgym.hip FWG_TL): 0 entry | 1 loads issued, before the integration | 2 integration done | 3 simulator rows
stored | 4 streamed windows landed (vmcnt 0) | 5 gym logic done | 6 bookkeeping rows stored | 7 observation built |
8 episode-end branch done | 9 outputs issued | 10 all stores acknowledged."""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fixed-wing-gym_amd")
OUT = os.path.join(PKG, "gym_fixed_wing", "_abl")
LIB = os.path.join(OUT, "libfwgym_timeline.so")
NAMES = ["entry->loads issued", "integration (incl. wait for state)", "store sim rows", "wait streamed windows (vmcnt0)",
         "gym logic", "store gym rows", "observation build", "episode-end branch", "outputs issued", "stores acknowledged"]


def build(extra=()):
    os.makedirs(OUT, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=fast",
           "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-DFWG_WITH_SPECS",
           "-DFWG_TIMELINE"] + list(extra) + ["-o", LIB, os.path.join(PKG, "csrc", "fwgym.hip")]
    subprocess.run(cmd, check=True)


def run():
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import numpy as np
    import torch
    from gym_fixed_wing import presets
    from gym_fixed_wing.vec_env import FixedWingVecEnv
    res = {}
    for wl, log_rows in (("c3", 0), ("c3", presets.OBS_LOG_ROWS), ("c2", 0), ("c5", 0)):
        cfg, ckw, skw, n, desc = presets.workload(wl)
        n = 65536
        vec = FixedWingVecEnv(cfg, num_envs=n, device=0, config_kw=ckw, sim_config_kw=skw, seed=1, derived_views=False,
                              obs_log_rows=log_rows, _lib_path=LIB)
        assert vec.spec_index >= 0
        vec.reset()
        trace = torch.zeros((n // 64, 16), dtype=torch.int64, device="cuda")
        vec._lib.fwg_debug_set_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        vec._lib.fwg_debug_set_trace(vec._handle, ctypes.c_void_p(trace.data_ptr()))
        acts = [torch.rand((n, 3), device="cuda") * 2 - 1 for _ in range(16)]
        for t in range(300):
            vec.step_device(acts[t % 16])
        torch.cuda.synchronize()
        rows = []
        for rep in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            vec.step_device(acts[rep % 16])
            e1.record()
            torch.cuda.synchronize()
            t = trace.cpu().numpy().astype(np.float64)
            rows.append((t, e0.elapsed_time(e1) * 1e3))
        T = np.stack([r[0] for r in rows])                    # [rep][block][16]
        span_ticks = (T[:, :, 10].max(axis=1) - T[:, :, 0].min(axis=1))
        key = "{}{}".format(wl, "_log" if log_rows else "")
        d = np.diff(T[:, :, :11], axis=2)                      # [rep][block][10]
        # s_memtime ticks at a constant 100 MHz on this part: report ticks and the implied microseconds
        med = np.median(d.reshape(-1, 10), axis=0)
        p90 = np.percentile(d.reshape(-1, 10), 90, axis=0)
        start_spread = np.median(T[:, :, 0].max(axis=1) - T[:, :, 0].min(axis=1))
        wave_total = np.median((T[:, :, 10] - T[:, :, 0]).reshape(-1))
        res[key] = {"phase_ticks_median": med.tolist(), "phase_ticks_p90": p90.tolist(), "first_to_last_wave_start_ticks": float(start_spread),
                    "wave_lifetime_ticks_median": float(wave_total), "kernel_span_ticks_median": float(np.median(span_ticks)),
                    "event_us_median": float(np.median([r[1] for r in rows]))}
        print("== {} ({}): event {:.2f} us; ticks: kernel span {:.0f}, wave lifetime {:.0f}, start spread {:.0f}".format(
            key, desc, res[key]["event_us_median"], np.median(span_ticks), wave_total, start_spread))
        for i, nm in enumerate(NAMES):
            print("   {:38s} median {:8.1f}  p90 {:8.1f} ticks".format(nm, med[i], p90[i]))
        vec.close()
    print(json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "timeline.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        run()
    else:
        build(sys.argv[2:])
