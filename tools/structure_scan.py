#!/usr/bin/env python3
"""One-flag structural A/B of the simulator restatement against the ONLY per-step data the reference ships for the
simulator half: the 25 878 raw rewards of 100 deterministic PID episodes (examples/evaluations/eval_res_PID_none.npy ->
tests/golden/eval_res_PID_none_rewards.json; scenarios examples/test_sets/test_set_wind_none_step20-20-3.npy ->
tests/golden/test_set_wind_none.json; protocol examples/evaluate_controller.py:44-169).

Every variant changes ONE thing in the float64 oracle stack (oracle/gym_restated.py over oracle/pyfly_restated.py) and is
scored on the full trace: mean / p90 |reward - published| over the steps both episodes have, episode-length error,
Va settling time.  Variants:

  integrator   rk45    scipy.solve_ivp (RK45, rtol 1e-3, atol 1e-6) over the 19-state ODE incl. 2nd-order elevons, states
                       conditioned (value / rate clips) inside the right-hand side -- the scheme recalled for PyFly 0.1.2
                        -- instead of exact actuators + RK4
  pid          int_first  integrators updated BEFORE use;  raw_q  pitch damping on body q instead of q cos(phi) - r sin(phi);
               elev30  elevator output limit +-30 deg instead of -30/+35
  actuators    tau=<s> throttle time constant;  elevon35  elevon/elevator/aileron value_max 35 deg
  aero         sideslip_sign  wind->body rotation through (0, alpha, beta) as a plain Euler rotation (sign of the sin(beta)
               terms flipped);  km=<v>, sprop=<v>, cdp=<v>  propulsion / parasitic drag constants;
               trimfit  the (k_motor, S_prop, C_D_p) triple that reproduces the reference's own full-throttle / 85 %-throttle
               airspeed lines (fixed_wing.py:944-972; tools/trim_lines.py)

Usage: python tools/structure_scan.py [--variants base,rk45,...] [--scenarios 100] [--jobs 8] [--out profiles/r02_structure_scan.json]
CPU only (float64 oracle); ~1-3 min per variant on 8 cores."""
import argparse
import copy
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import physics as ph  # noqa: E402
from oracle import pyfly_restated as pf  # noqa: E402
from oracle.gym_restated import FixedWingOracle  # noqa: E402

PKG = os.path.join(ROOT, "fixed-wing-gym_amd", "gym_fixed_wing")
_ORIG_RHS, _ORIG_SIM_STEP = ph.rhs, ph.sim_step


# ----------------------------------------------------------------------------------------------------------------------
# variant machinery
# ----------------------------------------------------------------------------------------------------------------------
class VariantPID(pf.PIDController):
    def __init__(self, dt, int_first=False, raw_q=False, elev30=False, gains=None):
        super().__init__(dt)
        self.int_first, self.raw_q = int_first, raw_q
        if elev30:
            self.delta_e_max = np.radians(30)
        for k, v in (gains or {}).items():
            setattr(self, k, v)

    def get_action(self, phi, theta, va, omega):
        e_V_a, e_phi, e_theta = va - self.va_r, phi - self.phi_r, theta - self.theta_r
        if self.int_first:
            self.int_va += self.dt * e_V_a
            self.int_roll += self.dt * e_phi
            self.int_pitch += self.dt * e_theta
        p = omega[0]
        q = omega[1] if self.raw_q else omega[1] * np.cos(phi) - omega[2] * np.sin(phi)
        delta_a = -self.k_p_phi * e_phi - self.k_i_phi * self.int_roll - self.k_d_phi * p
        delta_e = -self.k_p_theta * e_theta - self.k_i_theta * self.int_pitch - self.k_d_theta * q
        delta_t = -self.k_p_V * e_V_a - self.k_i_V * self.int_va
        if not self.int_first:
            self.int_va += self.dt * e_V_a
            self.int_roll += self.dt * e_phi
            self.int_pitch += self.dt * e_theta
        return np.asarray([np.clip(delta_e, self.delta_e_min, self.delta_e_max),
                           np.clip(delta_a, self.delta_a_min, self.delta_a_max), np.clip(delta_t, 0, 1.0)])


def rhs_sideslip_sign(spec, yb, act, wind, gust, fail):
    """oracle rhs with the aerodynamic force rotated by the plain Euler matrix R(0, alpha, beta) (sin(beta) terms with the
    opposite sign): evaluated by mirroring the side-force/drag coupling -- implemented by calling the oracle rhs on the
    state and correcting the three force rows."""
    dy, fail = _ORIG_RHS(spec, yb, act, wind, gust, fail)
    # recompute the pieces needed for the correction
    P = spec.params
    q4, vel = yb[:, ph.IQ], yb[:, ph.IV]
    Va, alpha, beta, ua, va, wa = ph.airspeed_factors(q4, vel, wind, gust[:, 0:3])
    Va = ph._lim(Va, spec.val_min[ph.VAR_ID["Va"]], spec.val_max[ph.VAR_ID["Va"]])
    pre = 0.5 * spec.rho * Va * Va * P["S_wing"]
    er, el = act[:, 0], act[:, 1]
    elevator, aileron = 0.5 * (er + el), 0.5 * (el - er)
    om = yb[:, ph.IW]
    pa, qa, ra = om[:, 0] - gust[:, 3], om[:, 1] - gust[:, 4], om[:, 2] - gust[:, 5]
    M, a0 = P["M"], P["a_0"]
    oms = 1.0 / ((1.0 + np.exp(M * (alpha - a0))) * (1.0 + np.exp(-M * (alpha + a0))))
    sig = 1.0 - oms
    sa, ca, sb, cb = np.sin(alpha), np.cos(alpha), np.sin(beta), np.cos(beta)
    sgn = np.sign(alpha)
    CL_lin = P["C_L_0"] + P["C_L_alpha"] * alpha
    inv2Va = 0.5 / Va
    CD = P["C_D_p"] + oms * CL_lin ** 2 / (np.pi * P["e"] * P["ar"]) + sig * (2.0 * sgn * sa ** 3)
    CDb = P["C_D_beta1"] * beta + P["C_D_beta2"] * beta * beta
    f_drag = pre * (CD + CDb + P["C_D_q"] * P["c"] * inv2Va * qa + P["C_D_delta_e"] * elevator ** 2)
    bv = P["b"] * inv2Va
    f_y = pre * (P["C_Y_0"] + P["C_Y_beta"] * beta + P["C_Y_p"] * bv * pa + P["C_Y_r"] * bv * ra + P["C_Y_delta_a"] * aileron)
    # difference between the two conventions: terms in sin(beta)
    dfx = 2.0 * ca * sb * f_y          # -ca sb Y  ->  +ca sb Y
    dfy = 2.0 * sb * f_drag            # -sb D     ->  +sb D
    dfz = 2.0 * sa * sb * f_y          # -sa sb Y  ->  +sa sb Y
    im = 1.0 / P["mass"]
    dy = dy.copy()
    dy[:, 10] += dfx * im
    dy[:, 11] += dfy * im
    dy[:, 12] += dfz * im
    return dy, fail


def sim_step_rk45(spec, y, cmd_inputs, wind, gust, rhs_fn=_ORIG_RHS, rtol=1e-3, atol=1e-6):
    """PyFly-style step (N = 1): one scipy solve_ivp(RK45) call over [0, dt] on the 19-state vector (13 rigid-body states,
    elevon_right / elevon_left / throttle values, their rates), command held, the right-hand side evaluated on the
    CONDITIONED state (value limits on values, dot_max on rates) while the ODE vector itself stays unclipped; the end
    state is conditioned once.  Same return convention as oracle.physics.sim_step."""
    from scipy.integrate import solve_ivp
    assert y.shape[0] == 1
    cmd_c, sp = ph.constrain_commands(spec, cmd_inputs)
    names = ("elevon_right", "elevon_left", "throttle")
    vmin = [spec.val_min[ph.VAR_ID[n]] for n in names]
    vmax = [spec.val_max[ph.VAR_ID[n]] for n in names]
    dmax = [spec.act[n]["dot_max"] for n in names[:2]]
    w0 = [spec.act[n]["omega_0"] for n in names[:2]]
    zt = [spec.act[n]["zeta"] for n in names[:2]]
    tau = spec.act["throttle"]["tau"]
    fail = np.zeros(1, dtype=np.int64)
    box = {"fail": fail}

    def cond(z):
        v = [ph._lim(z[13 + k], vmin[k], vmax[k]) for k in range(3)]
        d = [np.clip(z[16 + k], -dmax[k], dmax[k]) for k in range(2)]
        return v, d

    def fun(t, z):
        v, d = cond(z)
        dyb, box["fail"] = rhs_fn(spec, z[None, 0:13], np.array([v]), wind, gust, box["fail"])
        out = np.empty(19)
        out[0:13] = dyb[0]
        for k in range(2):
            out[13 + k] = d[k]
            out[16 + k] = -w0[k] ** 2 * (v[k] - sp[0, k]) - 2.0 * zt[k] * w0[k] * d[k]
        out[15] = (sp[0, 2] - v[2]) / tau
        out[18] = 0.0
        return out

    z0 = np.concatenate([y[0], [0.0]])
    sol = solve_ivp(fun, (0.0, spec.dt), z0, rtol=rtol, atol=atol)
    z = sol.y[:, -1]
    v, d = cond(z)
    yy = np.concatenate([z[0:13], v, d])[None, :]
    yy[:, ph.IQ] /= np.sqrt(np.sum(yy[:, ph.IQ] ** 2, axis=1, keepdims=True))
    fail = box["fail"]
    der = ph.derive(spec, yy, wind, gust)
    for name in ("roll", "pitch", "yaw", "Va", "alpha", "beta"):
        fail = ph._check(spec, fail, name, der[name])
    ok = fail == 0
    y_new = np.where(ok[:, None], yy, y)
    if not ok[0]:
        der = ph.derive(spec, y, wind, gust)
    return y_new, ok, fail - 1, cmd_c, der


class VariantPyFly(pf.PyFly):
    """oracle PyFly with a replaceable step function / rhs."""
    variant = {}

    def step(self, commands):
        v = self.variant
        rhs_fn = rhs_sideslip_sign if v.get("sideslip_sign") else _ORIG_RHS
        if not v.get("rk45") and rhs_fn is _ORIG_RHS:
            return super().step(commands)
        orig_step, orig_rhs = _ORIG_SIM_STEP, _ORIG_RHS
        try:
            if v.get("rk45"):
                ph.sim_step = lambda spec, y, cmd, wind, gust: sim_step_rk45(spec, y, cmd, wind, gust, rhs_fn=rhs_fn)
            else:
                ph.rhs = rhs_fn
            return super().step(commands)
        finally:
            ph.sim_step, ph.rhs = orig_step, orig_rhs


TRIMFIT = {"k_motor": 47.37, "S_prop": 0.012617, "C_D_p": 0.015475}   # tools/trim_lines.py --fit


def parse_variant(name):
    """'a+b+c' -> dict of flags."""
    v = {}
    for tok in name.split("+"):
        if tok in ("base", ""):
            continue
        if "=" in tok:
            k, val = tok.split("=")
            v[k] = float(val)
        else:
            v[tok] = True
    return v


def build_files(v, tmpdir):
    with open(os.path.join(PKG, "x8_param.json")) as f:
        par = json.load(f)
    with open(os.path.join(PKG, "sim_config.json")) as f:
        sim = json.load(f)
    if v.get("trimfit"):
        par.update(TRIMFIT)
    for key, pname in (("km", "k_motor"), ("sprop", "S_prop"), ("cdp", "C_D_p")):
        if key in v:
            par[pname] = v[key]
    for k, val in v.items():
        if k.startswith("P_"):
            par[k[2:]] = val
    for st in sim["states"]:
        if st["name"] == "throttle" and "tau" in v:
            st["tau"] = v["tau"]
        if v.get("elevon35") and st["name"] in ("elevator", "aileron", "elevon_left", "elevon_right"):
            st["value_max"] = 35
        if "dotmax" in v and st["name"].startswith("elevon"):
            st["dot_max"] = v["dotmax"]
        if "w0" in v and st["name"].startswith("elevon"):
            st["omega_0"] = v["w0"]
    tag = "{}_{}".format(os.getpid(), abs(hash(json.dumps(v, sort_keys=True))))
    pp, sp = os.path.join(tmpdir, "p_%s.json" % tag), os.path.join(tmpdir, "s_%s.json" % tag)
    with open(pp, "w") as f:
        json.dump(par, f)
    with open(sp, "w") as f:
        json.dump(sim, f)
    return pp, sp


def fly(args):
    """One scenario under one variant -> (rewards, info)."""
    v, sc, cfg, tmpdir = args
    import oracle.gym_restated as gr
    from gym_fixed_wing import evaluate as ev
    pp, sp = build_files(v, tmpdir)
    VariantPyFly.variant = v
    gr.PyFly = VariantPyFly
    env = FixedWingOracle(cfg, config_kw=ev.evaluation_overrides(True),
                          sim_config_kw={"turbulence": False, "turbulence_intensity": "none"},
                          sim_config_path=sp, sim_parameter_path=pp)
    obs = env.reset(state=sc["state"], target=sc["target"])
    gains = {k[2:]: val for k, val in v.items() if k.startswith("G_")}
    pid = VariantPID(env.simulator.dt, int_first=v.get("int_first", False), raw_q=v.get("raw_q", False),
                     elev30=v.get("elev30", False), gains=gains)
    pid.set_reference(sc["target"]["roll"], sc["target"]["pitch"], sc["target"]["Va"])
    rews, done, info = [], False, None
    while not done:
        if info is not None:
            pid.set_reference(info["target"]["roll"], info["target"]["pitch"], info["target"]["Va"])
        obs, r, done, info = env.step(pid.get_action(obs[0], obs[1], obs[2], obs[3:6]))
        rews.append(float(r))
    keep = {k: info.get(k) for k in ("termination", "settling_time", "rise_time", "control_variation", "success", "overshoot")}
    return rews, keep


def score(results, pub_rewards, pub):
    d_all, len_err, first = [], [], []
    for (rews, info), pr in zip(results, pub_rewards):
        n = min(len(rews), len(pr))
        d_all.append(np.abs(np.array(rews[:n]) - np.array(pr[:n])))
        len_err.append(abs(len(rews) - len(pr)) / len(pr))
        first.append(abs(rews[0] - pr[0]))
    d = np.concatenate(d_all)
    ok = [bool(i["success"]["all"]) for _, i in results]

    def mean_ok(metric, key):
        vals = [i[metric][key] for (_, i), o in zip(results, ok) if o and i[metric].get(key) is not None]
        return float(np.nanmean(vals)) if vals else float("nan")

    # the published metric lists are in episode-COMPLETION order: only whole-set means are comparable
    def pmean(metric, key):
        return float(np.nanmean(np.array([np.nan if x is None else x for x in pub[metric][key]], dtype=np.float64)))

    pub_set = {k: pmean("settling_time", k) * 0.01 for k in ("roll", "pitch", "Va")}
    pub_rise = {k: pmean("rise_time", k) * 0.01 for k in ("roll", "pitch", "Va")}
    return {
        "mean_abs_dreward": float(d.mean()), "p90_abs_dreward": float(np.percentile(d, 90)),
        "first_step_max_abs": float(np.max(first)),
        "episode_length_rel_err_mean": float(np.mean(len_err)), "episode_length_rel_err_p90": float(np.percentile(len_err, 90)),
        "success_all_%": 100.0 * float(np.mean(ok)),
        "settling_s": {k: mean_ok("settling_time", k) * 0.01 for k in ("roll", "pitch", "Va")},
        "rise_s": {k: mean_ok("rise_time", k) * 0.01 for k in ("roll", "pitch", "Va")},
        "control_variation": mean_ok("control_variation", "all"),
        "published_settling_s": pub_set, "published_rise_s": pub_rise,
        "published_control_variation": pmean("control_variation", "all"),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="base,rk45,int_first,raw_q,elev30,elevon35,sideslip_sign,tau=0.5,km=32,trimfit")
    ap.add_argument("--scenarios", type=int, default=100)
    ap.add_argument("--jobs", type=int, default=max(1, min(8, os.cpu_count() or 1)))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_structure_scan.json"))
    args = ap.parse_args()
    import configs
    cfg = configs.reference_like("examples")
    with open(os.path.join(ROOT, "tests", "golden", "test_set_wind_none.json")) as f:
        scen = json.load(f)[:args.scenarios]
    with open(os.path.join(ROOT, "tests", "golden", "eval_res_PID_none_rewards.json")) as f:
        pub_rewards = json.load(f)[:args.scenarios]
    with open(os.path.join(ROOT, "tests", "golden", "eval_res_PID_none.json")) as f:
        pub = json.load(f)
    import multiprocessing as mp
    tmpdir = tempfile.mkdtemp()
    out = {}
    if os.path.exists(args.out):
        with open(args.out) as f:
            out = json.load(f)
    with mp.get_context("fork").Pool(args.jobs) as pool:
        for name in args.variants.split(","):
            v = parse_variant(name)
            t0 = time.time()
            res = pool.map(fly, [(v, sc, cfg, tmpdir) for sc in scen], chunksize=1)
            s = score(res, pub_rewards, pub)
            s["scenarios"] = len(scen)
            out[name] = s
            print("{:28s} mean|dr| {:.5f} p90 {:.5f} first {:.1e} len_err mean {:.3f} p90 {:.3f} succ {:.0f}% settle Va {:.3f} (pub {:.3f}) roll {:.3f} ({:.3f}) pitch {:.3f} ({:.3f}) cv {:.3f} ({:.3f})  [{:.0f}s]".format(
                name, s["mean_abs_dreward"], s["p90_abs_dreward"], s["first_step_max_abs"], s["episode_length_rel_err_mean"],
                s["episode_length_rel_err_p90"], s["success_all_%"], s["settling_s"]["Va"], s["published_settling_s"]["Va"],
                s["settling_s"]["roll"], s["published_settling_s"]["roll"], s["settling_s"]["pitch"],
                s["published_settling_s"]["pitch"], s["control_variation"], s["published_control_variation"],
                time.time() - t0), flush=True)
            with open(args.out, "w") as f:
                json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
