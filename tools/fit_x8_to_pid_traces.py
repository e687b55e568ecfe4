#!/usr/bin/env python3
"""Fits the constants the shipped closed-loop PID traces are sensitive to.

PyFly 0.1.2's x8_param.mat / pyfly_config.json / PID gains are not available; the reference ships,
however, the per-step rewards of 100 deterministic PID episodes (examples/evaluations/eval_res_PID_none.npy ->
tests/golden/eval_res_PID_none_rewards.json).  This script replays those scenarios on the GPU env
(gym_fixed_wing/evaluate.py, ~0.3 s per evaluation of all 100 episodes) and searches over aircraft constants, actuator
time constants and PID gains for the smallest per-step reward distance.  Output: gpurun_out/x8_fit.json; the parameter
files are only changed by hand."""
import copy
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from gym_fixed_wing import evaluate as ev, presets  # noqa: E402
from gym_fixed_wing.config import DEFAULT_PARAMETERS, DEFAULT_SIM_CONFIG  # noqa: E402

AIRCRAFT = ["k_motor", "S_prop", "C_D_p", "C_L_alpha", "C_L_0", "C_l_p", "C_l_delta_a", "C_l_beta", "C_m_q", "C_m_delta_e",
            "C_m_alpha", "C_m_0", "C_n_r", "C_Y_beta"]
SIM = [("throttle", "tau"), ("elevon_left", "omega_0"), ("elevon_left", "dot_max")]
PID = ["k_p_V", "k_i_V", "k_p_phi", "k_d_phi", "k_p_theta", "k_i_theta", "k_d_theta"]


def main():
    with open(os.path.join(ROOT, "tests", "golden", "test_set_wind_none.json")) as f:
        scen = json.load(f)
    with open(os.path.join(ROOT, "tests", "golden", "eval_res_PID_none_rewards.json")) as f:
        pub = [np.array(r) for r in json.load(f)]
    with open(DEFAULT_PARAMETERS) as f:
        base = json.load(f)
    with open(DEFAULT_SIM_CONFIG) as f:
        sim_base = json.load(f)
    from gym_fixed_wing.pid import BatchedPID
    pid0 = BatchedPID(1, device="cpu")
    cfg = presets.preset("examples")
    tmp = tempfile.mkdtemp()
    nA, nS, nP = len(AIRCRAFT), len(SIM), len(PID)
    count = [0]
    best = [1e9, None]

    def objective(z):
        p = dict(base)
        for n, s in zip(AIRCRAFT, z[:nA]):
            p[n] = float(base[n] * s)
        sim = copy.deepcopy(sim_base)
        for (st, key), s in zip(SIM, z[nA:nA + nS]):
            for e in sim["states"]:
                if e["name"] == st or (st == "elevon_left" and e["name"] == "elevon_right"):
                    e[key] = float(e[key] * s)
        ppath, spath = os.path.join(tmp, "p.json"), os.path.join(tmp, "s.json")
        json.dump(p, open(ppath, "w"))
        json.dump(sim, open(spath, "w"))
        gains = {n: float(getattr(pid0, n) * s) for n, s in zip(PID, z[nA + nS:])}
        try:
            res = ev.evaluate_on_set(scen, cfg, device=0, sim_parameter_path=ppath, sim_config_path=spath, pid_gains=gains)
        except Exception as e:  # infeasible parameter combination
            print("eval failed:", e)
            return 10.0
        err, lerr = [], []
        for ours, ref in zip(res["rewards"], pub):
            n = min(len(ours), len(ref))
            d = np.abs(np.array(ours[:n]) - ref[:n])
            err.append(np.mean(d))
            lerr.append(abs(len(ours) - len(ref)) / len(ref))
        j = float(np.mean(err) + 0.1 * np.mean(lerr))
        count[0] += 1
        if j < best[0]:
            best[0], best[1] = j, np.array(z)
            print("eval %4d  J=%.5f  mean|dr| %.5f  len-err %.4f  success %.0f%%" % (
                count[0], j, np.mean(err), np.mean(lerr), 100 * np.mean([bool(s) for s in res["success"]["all"]])), flush=True)
        return j

    from scipy.optimize import minimize
    n_eval = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    z0 = np.ones(nA + nS + nP)
    j0 = objective(z0)
    r = minimize(objective, z0, method="Powell", bounds=[(0.6, 1.6)] * len(z0),
                 options={"maxfev": n_eval, "xtol": 2e-3, "ftol": 1e-5})
    z = best[1]
    out = {"J0": j0, "J": best[0],
           "aircraft": {n: float(base[n] * s) for n, s in zip(AIRCRAFT, z[:nA])},
           "sim_scale": {"{}.{}".format(*k): float(s) for k, s in zip(SIM, z[nA:nA + nS])},
           "pid": {n: float(getattr(pid0, n) * s) for n, s in zip(PID, z[nA + nS:])}}
    print("BEST", json.dumps(out))
    with open(os.path.join(ROOT, "gpurun_out", "x8_fit.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    main()
