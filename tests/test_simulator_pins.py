"""Pins of the SIMULATOR half of the oracle to reference-held data (the arithmetic itself lives in the un-vendored
pyfly-fixed-wing==0.1.2, so exact parity is unpinned; these are the two data sets the reference ships that constrain it):

  A  the converged-airspeed lines hard-coded in the reference's Va "compensate" target, fixed_wing.py:944-972
     (full throttle, pitch <= -2.5 deg: 28.434 - 40.0841 theta; 85 % throttle, pitch >= 5 deg: 26.27 - 41.2529 theta):
     exact steady-state trim of oracle/physics.rhs, no controller in the loop (tools/trim_lines.py);
  B  the per-step rewards of the shipped PID evaluation (examples/evaluations/eval_res_PID_none.npy, 25 878 steps, 100
     deterministic episodes; here a fixed subset of 25 episodes to keep the CPU suite short -- the GPU suite flies all
     100, tests/test_evaluate.py): closed loop of oracle gym + oracle simulator + PID.

The bands are what the parameter identification of round 2 achieves (profiles/r02_structure_scan.json);
they GATE: a change of the simulator restatement that moves away from the reference's data fails here."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import trim_lines as tl  # noqa: E402


def test_trim_airspeed_reproduces_the_reference_compensate_lines_within_2_percent():
    r = tl.line_residuals()
    assert r.shape == (7,)
    assert np.max(np.abs(r)) <= 0.02, r
    # and the slope (what the published propulsion constants miss by a factor of 3-4): d Va / d theta ~ -40 m/s/rad
    spec = tl.load_spec()
    va = [tl.trim(spec, th, 0.85, va0=tl.line_85(th))[0] for th in tl.THETA_85]
    slope = (va[-1] - va[0]) / (tl.THETA_85[-1] - tl.THETA_85[0])
    assert -52.0 < slope < -36.0, slope


def test_published_propulsion_constants_do_not_fit_the_lines():
    """The discrepancy is explained by NAMED constants: with the published S_prop / C_D_p / C_m_delta_e the same trim is
    off by 4-70 % (this is why x8_param.json departs from the published table for exactly these three)."""
    r = tl.line_residuals({"S_prop": 0.1018, "C_D_p": 0.0197, "C_m_delta_e": -0.206})
    assert np.min(np.abs(r)) > 0.03 and np.max(np.abs(r)) > 0.5


def test_pid_traces_of_the_reference_within_bands():
    import structure_scan as ss
    import configs
    import tempfile
    cfg = configs.reference_like("examples")
    with open(os.path.join(HERE, "golden", "test_set_wind_none.json")) as f:
        scen = json.load(f)
    with open(os.path.join(HERE, "golden", "eval_res_PID_none_rewards.json")) as f:
        pub = json.load(f)
    idx = list(range(0, 100, 4))
    tmp = tempfile.mkdtemp()
    d, len_err, first, ok = [], [], [], []
    for i in idx:
        rews, info = ss.fly(({}, scen[i], cfg, tmp))
        n = min(len(rews), len(pub[i]))
        d.append(np.abs(np.array(rews[:n]) - np.array(pub[i][:n])))
        len_err.append(abs(len(rews) - len(pub[i])) / len(pub[i]))
        first.append(abs(rews[0] - pub[i][0]))
        ok.append(bool(info["success"]["all"]))
    d = np.concatenate(d)
    report = {"mean_abs_dreward": float(d.mean()), "p90_abs_dreward": float(np.percentile(d, 90)),
              "first_step_max": float(max(first)), "len_err_mean": float(np.mean(len_err)),
              "len_err_p90": float(np.percentile(len_err, 90)), "success": float(np.mean(ok))}
    print(report)
    assert report["success"] == 1.0                      # published: 100 %
    assert report["first_step_max"] < 1e-3               # kinematics / error / reward plumbing (round 1: 6.5e-4)
    assert report["mean_abs_dreward"] < 0.016            # round 1 constants: 0.037 on the same metric
    assert report["p90_abs_dreward"] < 0.032             # round 1: 0.10
    assert report["len_err_mean"] < 0.14                 # round 1: 0.17
    assert report["len_err_p90"] < 0.30                  # round 1: 0.46


def test_integration_scheme_convergence_claim():
    """with the actuators advanced exactly in 16 micro-steps, ONE classical RK4 step per env step is as
    accurate as four (the error is set by the actuator micro-stepping, i.e. by where the rate limit switches), and both
    are at the level of the reference's own adaptive RK45 at rtol 1e-3 -- reproduced by tools/convergence.py
    (profiles/r02_convergence.json holds the 2 000-env table)."""
    import convergence as cv
    res = cv.run(n=300, rk45_subset=12)
    one, four = res["rk4x1_micro16"], res["rk4x4_micro16"]
    assert one["after_100"] < 1.3 * four["after_100"]
    assert one["after_100"] < 8e-3 and one["after_1"] < 4e-3
    assert res["rk4x1_micro4"]["after_100"] > 3 * one["after_100"]          # too few micro-steps does show
    assert res["rk4x4_micro64"]["after_100"] < 0.3 * one["after_100"]        # and the scheme converges when both are refined
    rk45 = [v for k, v in res.items() if k.startswith("scipy_rk45")][0]
    assert one["after_100"] < 3 * max(rk45["after_100"], 1e-3)
