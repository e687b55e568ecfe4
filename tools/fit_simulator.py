#!/usr/bin/env python3
"""Joint fit of the few simulator constants that the reference's OWN data can identify (PyFly 0.1.2's parameter files are
not available):

  data set A  the airspeed lines hard-coded in the reference's Va "compensate" target (fixed_wing.py:944-972): converged
              airspeed at full throttle, pitch <= -2.5 deg: 28.434 - 40.0841 theta; at 85 % throttle, pitch >= 5 deg:
              26.27 - 41.2529 theta  -> steady-state trim of the oracle (tools/trim_lines.py)
  data set B  the 25 878 per-step rewards of the shipped PID evaluation (examples/evaluations/eval_res_PID_none.npy)
              -> closed-loop replay with the oracle (tools/structure_scan.py)

objective = mean |reward - published| (B) + W_LINES * rms relative line residual (A); Nelder-Mead over multiplicative
factors on the starting values.  Output: profiles/r02_simulator_fit.json (parameter files are edited by hand from it)."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import structure_scan as ss  # noqa: E402
import trim_lines as tl  # noqa: E402

W_LINES = 0.5
NAMES = ["km", "sprop", "cdp", "P_C_m_delta_e", "P_C_l_delta_a", "tau", "dotmax"]
START = {"km": 40.0, "sprop": 0.0175, "cdp": 0.0102, "P_C_m_delta_e": -0.25, "P_C_l_delta_a": 0.1202, "tau": 0.2, "dotmax": 200.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=5)
    ap.add_argument("--maxiter", type=int, default=120)
    ap.add_argument("--free", default=",".join(NAMES))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_simulator_fit.json"))
    args = ap.parse_args()
    free = args.free.split(",")
    import configs
    cfg = configs.reference_like("examples")
    with open(os.path.join(ROOT, "tests", "golden", "test_set_wind_none.json")) as f:
        scen = json.load(f)
    with open(os.path.join(ROOT, "tests", "golden", "eval_res_PID_none_rewards.json")) as f:
        pub_rewards = json.load(f)
    with open(os.path.join(ROOT, "tests", "golden", "eval_res_PID_none.json")) as f:
        pub = json.load(f)
    tmpdir = tempfile.mkdtemp()
    pool = mp.get_context("fork").Pool(args.jobs)
    hist = []

    def evaluate(z):
        v = dict(START)
        for n, s in zip(free, z):
            v[n] = START[n] * float(s)
        if v["km"] < 20 or v["sprop"] <= 0 or v["cdp"] < 0 or v["tau"] <= 0.01 or v["dotmax"] < 20:
            return 1e3
        res = pool.map(ss.fly, [(v, sc, cfg, tmpdir) for sc in scen], chunksize=1)
        s = ss.score(res, pub_rewards, pub)
        lines = tl.line_residuals({"k_motor": v["km"], "S_prop": v["sprop"], "C_D_p": v["cdp"], "C_m_delta_e": v["P_C_m_delta_e"]})
        rms = float(np.sqrt(np.mean(np.square(lines))))
        obj = s["mean_abs_dreward"] + W_LINES * rms
        hist.append({"params": v, "objective": obj, "mean_abs_dreward": s["mean_abs_dreward"], "line_rms": rms,
                     "line_max": float(np.max(np.abs(lines))), "settling_s": s["settling_s"], "rise_s": s["rise_s"],
                     "control_variation": s["control_variation"], "len_err_p90": s["episode_length_rel_err_p90"],
                     "success": s["success_all_%"]})
        print("{:3d} obj {:.5f} dr {:.5f} lines rms {:.4f} max {:.4f} settle {:.2f}/{:.2f}/{:.2f} cv {:.3f} | {}".format(
            len(hist), obj, s["mean_abs_dreward"], rms, np.max(np.abs(lines)), s["settling_s"]["roll"], s["settling_s"]["pitch"],
            s["settling_s"]["Va"], s["control_variation"], " ".join("{}={:.5g}".format(n, v[n]) for n in free)), flush=True)
        best = min(hist, key=lambda h: h["objective"])
        with open(args.out, "w") as f:
            json.dump({"best": best, "start": START, "free": free, "w_lines": W_LINES, "evaluations": len(hist),
                       "published": {"settling_s": s["published_settling_s"], "rise_s": s["published_rise_s"],
                                     "control_variation": s["published_control_variation"]}}, f, indent=1)
        return obj

    from scipy.optimize import minimize
    x0 = np.ones(len(free))
    simplex = [x0] + [x0 + 0.08 * np.eye(len(free))[i] for i in range(len(free))]
    minimize(evaluate, x0, method="Nelder-Mead", options={"maxfev": args.maxiter, "initial_simplex": np.array(simplex),
                                                            "xatol": 2e-3, "fatol": 2e-5})


if __name__ == "__main__":
    main()
