#!/usr/bin/env python3
"""Steady-state airspeed of the oracle aircraft at a held pitch angle and throttle setting, against the two lines the
reference hard-codes in its Va "compensate" target (gym_fixed_wing/fixed_wing.py:944-972):

    full throttle, pitch <= -2.5 deg ("Compensate the effects of gravity on airspeed"):  va_end = 28.434 - 40.0841 theta
    85 % throttle,  pitch >=  5  deg ("Converged velocity at 85 % throttle"):              va_end = 26.27  - 41.2529 theta

These constants are measurements of PyFly 0.1.2's thrust / drag / gravity balance and are the only reference-held data
that pin the propulsion model WITHOUT a controller in the loop.  The trim is exact: the wings-level steady state
(u_dot = w_dot = q_dot = 0 at pitch theta, elevons together, given throttle) of oracle/physics.rhs, solved for
(Va, alpha, elevator) -- no simulation, no PID.

  python tools/trim_lines.py            # table for the shipped parameter file
  python tools/trim_lines.py --fit      # least-squares (k_motor, S_prop, C_D_p) on the 7 line points"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
from oracle import physics as ph  # noqa: E402

PKG = os.path.join(ROOT, "fixed-wing-gym_amd", "gym_fixed_wing")
THETA_FULL = np.radians([-10.0, -7.5, -5.0, -2.5])     # branch 1 of fixed_wing.py:953
THETA_85 = np.radians([5.0, 7.5, 10.0])                # branch 2 of fixed_wing.py:960


def line_full(theta):
    return 28.434 - 40.0841 * theta


def line_85(theta):
    return 26.27 - 41.2529 * theta


def load_spec(**over):
    with open(os.path.join(PKG, "sim_config.json")) as f:
        sim = json.load(f)
    with open(os.path.join(PKG, "x8_param.json")) as f:
        par = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
    par.update(over)
    return ph.SimSpec(sim, par)


def trim(spec, theta, throttle, va0=25.0):
    """(Va, alpha, elevator) of the wings-level steady state at pitch theta."""
    from scipy.optimize import fsolve

    def f(x):
        Va, al, de = x
        y = np.zeros((1, 13))
        y[0, 0:4] = ph.quat_from_euler(np.array([0.0]), np.array([theta]), np.array([0.0]))[0]
        y[0, 10], y[0, 12] = Va * np.cos(al), Va * np.sin(al)
        dy, _ = ph.rhs(spec, y, np.array([[de, de, throttle]]), np.zeros((1, 3)), np.zeros((1, 6)), np.zeros(1, dtype=np.int64))
        return [dy[0, 10], dy[0, 12], dy[0, 5]]

    x, _, ier, _ = fsolve(f, [va0, 0.02, 0.0], full_output=True)
    if ier != 1:
        return np.array([np.nan, np.nan, np.nan])
    return x


def line_residuals(over=None, spec=None):
    """Relative residuals (ours - line) / line on the 7 points."""
    spec = load_spec(**(over or {})) if spec is None else spec
    r = []
    for th in THETA_FULL:
        r.append((trim(spec, th, 1.0, va0=line_full(th))[0] - line_full(th)) / line_full(th))
    for th in THETA_85:
        r.append((trim(spec, th, 0.85, va0=line_85(th))[0] - line_85(th)) / line_85(th))
    return np.array(r)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fit", action="store_true")
    args = ap.parse_args()
    spec = load_spec()
    print("theta[deg]  throttle  Va_oracle  Va_reference_line  rel.err   alpha[deg] elevator[deg]")
    for th, thr, line in [(t, 1.0, line_full) for t in THETA_FULL] + [(t, 0.85, line_85) for t in THETA_85]:
        x = trim(spec, th, thr, va0=line(th))
        print("{:9.1f}  {:8.2f}  {:9.3f}  {:17.3f}  {:+.4f}   {:9.2f} {:9.2f}".format(np.degrees(th), thr, x[0], line(th),
                                                                                  (x[0] - line(th)) / line(th), np.degrees(x[1]), np.degrees(x[2])))
    if args.fit:
        from scipy.optimize import least_squares
        p = spec.params
        sol = least_squares(lambda z: line_residuals({"k_motor": z[0], "S_prop": z[1], "C_D_p": z[2]}),
                            [p["k_motor"], p["S_prop"], p["C_D_p"]], bounds=([20, 1e-4, 0.0], [300, 1.0, 0.5]))
        print("fit (k_motor, S_prop, C_D_p) =", sol.x, " max |rel residual| =", float(np.abs(sol.fun).max()))


if __name__ == "__main__":
    main()
