#!/usr/bin/env python3
"""Self-convergence of the integration scheme: exact actuator micro-steps + classical RK4 on the 13
rigid-body states.  For N random states and command sequences the oracle is stepped with (RK4 sub-steps, actuator
micro-steps) = (1,16) [the shipped default], (2,16), (4,16), (1,4), (1,64), (4,64) and compared with a fine reference
(32 sub-steps, 1 024 micro-steps): maximum over envs and rigid-body states of |x - x_ref| / max(|x_ref|, scale) after 1
step and after 100 steps.  Also the reference's own kind of integrator -- scipy RK45 at rtol 1e-3 / atol 1e-6 over the
19-state ODE (tools/structure_scan.py sim_step_rk45) -- on a subset.

  python tools/convergence.py [--n 2000] [--out profiles/r02_convergence.json]     (CPU, float64, about a minute)"""
import argparse
import copy
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)
from oracle import physics as ph  # noqa: E402

PKG = os.path.join(ROOT, "fixed-wing-gym_amd", "gym_fixed_wing")
SCALE = np.array([1, 1, 1, 1, 1, 1, 1, 100, 100, 100, 20, 5, 5], dtype=np.float64)
SCHEMES = [(1, 16), (2, 16), (4, 16), (1, 4), (1, 64), (4, 64)]


def make_spec(nsub, micro):
    with open(os.path.join(PKG, "sim_config.json")) as f:
        sim = json.load(f)
    with open(os.path.join(PKG, "x8_param.json")) as f:
        par = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
    sim = copy.deepcopy(sim)
    sim["integrator"] = {"method": "rk4", "substeps": nsub, "actuator_microsteps": micro}
    return ph.SimSpec(sim, par)


def random_states(rng, n):
    vals = {"roll": rng.uniform(-1.9, 1.9, n), "pitch": rng.uniform(-0.78, 0.78, n), "yaw": rng.uniform(-3.1, 3.1, n),
            "omega_p": rng.uniform(-1, 1, n), "omega_q": rng.uniform(-1, 1, n), "omega_r": rng.uniform(-1, 1, n),
            "position_n": np.zeros(n), "position_e": np.zeros(n), "position_d": np.full(n, -100.0),
            "velocity_u": rng.uniform(12, 28, n), "velocity_v": rng.uniform(-5, 5, n), "velocity_w": rng.uniform(-5, 5, n),
            "elevator": rng.uniform(-0.3, 0.3, n), "aileron": rng.uniform(-0.3, 0.3, n), "throttle": rng.uniform(0, 1, n),
            "wind_n": np.zeros(n), "wind_e": np.zeros(n), "wind_d": np.zeros(n)}
    return vals


def rollout(spec, y0, wind, cmds):
    y = y0.copy()
    alive = np.ones(y.shape[0], dtype=bool)
    gust = np.zeros((y.shape[0], 6))
    out = {}
    for t in range(cmds.shape[0]):
        y, ok, _, _, _ = ph.sim_step(spec, y, cmds[t], wind, gust)
        alive &= ok
        if t + 1 in (1, 100):
            out[t + 1] = (y.copy(), alive.copy())
    return out


def run(n=2000, steps=100, seed=0, rk45_subset=40):
    rng = np.random.default_rng(seed)
    spec0 = make_spec(1, 16)
    y0, wind = ph.initial_state(spec0, random_states(rng, n))
    lim = np.radians(30.0)
    cmds = np.stack([rng.uniform([-lim, -lim, 0.0], [lim, lim, 1.0], size=(n, 3)) for _ in range(steps)])
    hold = rng.uniform(size=(steps, n, 1)) < 0.7            # commands jump on 30 % of the steps (rate-limit switching)
    for t in range(1, steps):
        cmds[t] = np.where(hold[t], cmds[t - 1], cmds[t])
    ref = rollout(make_spec(32, 1024), y0, wind, cmds)
    res = {}
    for nsub, micro in SCHEMES:
        got = rollout(make_spec(nsub, micro), y0, wind, cmds)
        row = {}
        for k in (1, steps):
            (yr, ar), (yg, ag) = ref[k], got[k]
            m = ar & ag
            err = np.abs(yg[m, :13] - yr[m, :13]) / np.maximum(np.abs(yr[m, :13]), SCALE)
            row["after_{}".format(k)] = float(err.max())
            row["p99_after_{}".format(k)] = float(np.percentile(err.max(axis=1), 99))
        res["rk4x{}_micro{}".format(nsub, micro)] = row
    if rk45_subset:
        import structure_scan as ss
        sub = np.arange(rk45_subset)
        worst = {1: 0.0, steps: 0.0}
        for i in sub:
            y = y0[i:i + 1].copy()
            okk = True
            for t in range(steps):
                y, ok, _, _, _ = ss.sim_step_rk45(spec0, y, cmds[t, i:i + 1], wind[i:i + 1], np.zeros((1, 6)))
                okk &= bool(ok[0])
                if t + 1 in (1, steps) and okk and ref[t + 1][1][i]:
                    e = np.abs(y[0, :13] - ref[t + 1][0][i, :13]) / np.maximum(np.abs(ref[t + 1][0][i, :13]), SCALE)
                    worst[t + 1] = max(worst[t + 1], float(e.max()))
        res["scipy_rk45_rtol1e-3 ({} envs)".format(rk45_subset)] = {"after_1": worst[1], "after_{}".format(steps): worst[steps]}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_convergence.json"))
    args = ap.parse_args()
    res = run(args.n)
    print("{:34s} {:>12s} {:>12s} {:>14s}".format("scheme", "1 step", "100 steps", "p99 100 steps"))
    for k, v in res.items():
        print("{:34s} {:12.2e} {:12.2e} {:>14s}".format(k, v["after_1"], v["after_100"],
                                                     "{:.2e}".format(v["p99_after_100"]) if "p99_after_100" in v else "-"))
    with open(args.out, "w") as f:
        json.dump({"n": args.n, "reference": "rk4x32_micro1024", "metric": "max |x - x_ref| / max(|x_ref|, scale), 13 rigid-body states",
                   "schemes": res}, f, indent=1)


if __name__ == "__main__":
    main()
