#!/usr/bin/env python3
"""Two-sided scan of the turbulence path against EVERYTHING the reference ships about PyFly under turbulence:
examples/evaluations/eval_res_PID_{light,moderate,severe}.npy -- per-episode metrics and the raw per-step rewards of the
100 PID episodes per intensity (the numbers of examples/README.md:39-47 are their means).

Every variant changes how the Dryden gust is generated / enters the simulator in the float64 oracle stack
(oracle/gym_restated.py over oracle/pyfly_restated.py) and is scored two-sidedly on

  * success rates (roll / pitch / Va / all), mean rise / settling time, overshoot, control variation (successful episodes),
  * the reward level by time bucket over the episodes still running (mean, std, 5th percentile): how hard the gusts throw
    the aircraft around, as a function of time since reset,
  * the episode-length distribution (median, 90th percentile, share that runs to the time limit).

The gust series is pre-simulated per episode as PyFly does (turbulence_sim_length samples at reset, fixed_wing.py:40), with
scipy.signal.lsim over the six MIL-F-8785C transfer functions -- an independent restatement of the same filters the oracle
realises as one joint 8-state system (oracle/physics.py:dryden_continuous); `statespace` switches to the oracle's own
streaming form as a cross-check.

Variant tokens (joined with '+'):
  scale=<k> lin=<k> ang=<k>     multiply all / linear / angular gust components
  w20=<kt>                      wind speed at 20 ft (overrides the intensity table 15 / 30 / 45 kt)
  h=<m> va=<m/s>                nominal altitude / airspeed of the filters
  stationary                    filter states start from their stationary distribution instead of zero
  noisemap=mil                  q_g driven by the w noise and r_g by the v noise (PyFly recalled: q <- noise[1], r <- noise[2])
  angsign=<+1|-1>               sign of q_g and r_g
  nohold                        gust interpolated linearly within the env step (stages at t, t+dt/2, t+dt) instead of held
  accum=<k>                     the steady wind accumulates k * (linear gust) every step (in-place `+=` on the wind vector)
  nedgust                       linear gust added in the NED frame instead of the body frame
  diff / difflin                the gust is the first difference of all six / the three linear filter outputs
  zoh                           noise held over the step in the filter simulation (scipy lsim interp=False) instead of interpolated
  statespace                    the oracle's streaming joint filter with the product's Philox stream (what the kernels run;
                                turbulence_output as in sim_config.json unless `filter` / `increment` is given)
  P_<name>=<v>                  aircraft parameter override
  wind=<k>                      SCENARIO side (round 4): a steady wind of magnitude k x W20 (the intensity's wind speed at 20 ft,
                                15 / 30 / 45 kt) in a random horizontal direction per episode is written into the scenario's
                                state["wind_n/e/d"] (get_initial_state, fixed_wing.py:848-862, records them) WITHOUT re-deriving the
                                body velocities: the initial airspeed vector is off by the wind.  (Re-deriving them is exactly the
                                no-wind episode: d/dt(v - R'w) = -omega x (v - R'w) + f/m, a steady wind drops out of the dynamics.)
  wind3d                        ... direction uniform on the sphere instead of horizontal
  windabs=<m/s>                 ... magnitude in m/s whatever the intensity

Usage: python tools/turbulence_scan.py --intensity severe --variants base,scale=2 [--seeds 1] [--jobs 8]
CPU only; ~1 min per variant and intensity on 8 cores."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import physics as ph  # noqa: E402
from oracle import pyfly_restated as pf  # noqa: E402
from oracle.gym_restated import FixedWingOracle  # noqa: E402
import structure_scan as ss  # noqa: E402
import turbulence_stats as tstats  # noqa: E402



# ----------------------------------------------------------------------------------------------------------------------
# Dryden series, transfer-function form (restated from MIL-F-8785C as PyFly 0.1.2's dryden.py uses it)
# ----------------------------------------------------------------------------------------------------------------------
def dryden_tfs(b, h=100.0, Va=25.0, W20_kt=15.0):
    m2f, kn2fps = 3.281, 1.6878
    h, b, V = h * m2f, b * m2f, Va * m2f
    W20 = W20_kt * kn2fps
    Lu = h / (0.177 + 0.000823 * h) ** 1.2
    Lv, Lw = Lu, h
    sw = 0.1 * W20
    su = sw / (0.177 + 0.000823 * h) ** 0.4
    sv = su
    Ku, Kv, Kw = su * np.sqrt(2 * Lu / (np.pi * V)), sv * np.sqrt(Lv / (np.pi * V)), sw * np.sqrt(Lw / (np.pi * V))
    Tu, Tv1, Tv2, Tw1, Tw2 = Lu / V, np.sqrt(3.0) * Lv / V, Lv / V, np.sqrt(3.0) * Lw / V, Lw / V
    Kp = sw * np.sqrt(0.8 / V) * (np.pi / (4 * b)) ** (1 / 6) / Lw ** (1 / 3)
    Kq = Kr = 1.0 / V
    Tp = 4 * b / (np.pi * V)
    Tq, Tr = Tp, 3 * b / (np.pi * V)
    f2m = 1.0 / m2f
    return {
        "u": ([f2m * Ku], [Tu, 1.0]),
        "v": ([f2m * Kv * Tv1, f2m * Kv], [Tv2 ** 2, 2 * Tv2, 1.0]),
        "w": ([f2m * Kw * Tw1, f2m * Kw], [Tw2 ** 2, 2 * Tw2, 1.0]),
        "p": ([Kp], [Tp, 1.0]),
        "q": ([-Kw * Kq * Tw1, -Kw * Kq, 0.0], [Tq * Tw2 ** 2, Tw2 ** 2 + 2 * Tq * Tw2, Tq + 2 * Tw2, 1.0]),
        "r": ([Kv * Kr * Tv1, Kv * Kr, 0.0], [Tr * Tv2 ** 2, Tv2 ** 2 + 2 * Tr * Tv2, Tr + 2 * Tv2, 1.0]),
    }


def simulate_series(v, b, intensity, length, dt, rng):
    """-> gust[length, 6] (u, v, w, p, q, r), sample k = value used during env step k."""
    from scipy import signal
    w20 = v.get("w20", {"light": 15.0, "moderate": 30.0, "severe": 45.0}[intensity])
    tfs = dryden_tfs(b, v.get("h", 100.0), v.get("va", 25.0), w20)
    burn = int(v.get("burn", 6000)) if v.get("stationary") else 0
    n = length + burn
    t = np.arange(n) * dt
    noise = np.sqrt(np.pi / dt) * rng.standard_normal(size=(4, n))
    if v.get("noisemap") == "mil":
        drive = {"u": 0, "v": 1, "w": 2, "p": 3, "q": 2, "r": 1}
    else:
        drive = {"u": 0, "v": 1, "w": 2, "p": 3, "q": 1, "r": 2}
    out = np.zeros((n, 6))
    for j, k in enumerate("uvwpqr"):
        num, den = tfs[k]
        _, y, _ = signal.lsim(signal.lti(num, den), U=noise[drive[k]], T=t, interp=not v.get("zoh", False))
        out[:, j] = y
    out = out[burn:]
    if v.get("diff") or v.get("difflin"):     # first difference of the filter outputs (what the published Va jitter looks like)
        cols = slice(0, 6) if v.get("diff") else slice(0, 3)
        out[:, cols] = np.diff(out[:, cols], axis=0, prepend=0.0)
    sgn = v.get("angsign", 1.0)
    out[:, 4:6] *= sgn
    out[:, 0:3] *= v.get("scale", 1.0) * v.get("lin", 1.0)
    out[:, 3:6] *= v.get("scale", 1.0) * v.get("ang", 1.0)
    return out


class TurbPyFly(ss.VariantPyFly):
    """oracle PyFly whose gust comes from a per-episode pre-simulated series."""
    intensity = "severe"
    episode_seed = 0
    _series = None

    def reset(self, state=None, turbulence_noise=None, draw=None):
        self._series = None
        super().reset(state=state, turbulence_noise=turbulence_noise, draw=draw)
        v = self.variant
        if self.turbulence and not v.get("statespace"):
            rng = np.random.RandomState(self.episode_seed)
            self._series = simulate_series(v, self._file_span, self.intensity, int(self.cfg.get("turbulence_sim_length", 1500)) + 2,
                                           self.dt, rng)
        self._wind0 = self._wind.copy()

    def _gust(self, spec):
        if not self.turbulence:
            return np.zeros((1, 6))
        if self._series is None:
            g = super()._gust(spec)
            v = self.variant
            g = g.copy()
            g[:, 0:3] *= v.get("scale", 1.0) * v.get("lin", 1.0)
            g[:, 3:6] *= v.get("scale", 1.0) * v.get("ang", 1.0)
            return g
        return self._series[min(self.cur_sim_step, len(self._series) - 1)][None, :].copy()

    def step(self, commands):
        v = self.variant
        if self.turbulence and "accum" in v:
            g = self._gust(self._spec())
            self._wind = self._wind + v["accum"] * g[:, 0:3]
        return super().step(commands)


def fly(args):
    v, sc, cfg, tmpdir, intensity, seed = args
    import oracle.gym_restated as gr
    from gym_fixed_wing import evaluate as ev
    pp, sp = ss.build_files(v, tmpdir)
    TurbPyFly.variant = v
    TurbPyFly.intensity = intensity
    TurbPyFly.episode_seed = seed
    gr.PyFly = TurbPyFly
    skw = {"turbulence": intensity != "none", "turbulence_intensity": intensity if intensity != "none" else "light"}
    if v.get("filter") or v.get("increment"):
        skw["turbulence_output"] = "filter" if v.get("filter") else "increment"
    env = FixedWingOracle(cfg, config_kw=ev.evaluation_overrides(True), sim_config_kw=skw,
                          sim_config_path=sp, sim_parameter_path=pp)
    env.simulator.seed(seed)
    env.simulator.env_id = seed & 0xFFFF
    if "wind" in v or "windabs" in v:   # scenario-side steady wind (see the module docstring)
        w20 = {"none": 0.0, "light": 15.0, "moderate": 30.0, "severe": 45.0}[intensity] * 0.514444
        mag = v["windabs"] if "windabs" in v else v["wind"] * w20
        rs = np.random.RandomState(777 + seed)
        if v.get("wind3d"):
            d = rs.standard_normal(3)
            d /= np.linalg.norm(d)
        else:
            psi = rs.uniform(0, 2 * np.pi)
            d = np.array([np.cos(psi), np.sin(psi), 0.0])
        st = dict(sc["state"])
        st["wind_n"], st["wind_e"], st["wind_d"] = (float(mag * d[0]), float(mag * d[1]), float(mag * d[2]))
        sc = {"state": st, "target": sc["target"]}
    obs = env.reset(state=sc["state"], target=sc["target"])
    pid = ss.VariantPID(env.simulator.dt)
    pid.set_reference(sc["target"]["roll"], sc["target"]["pitch"], sc["target"]["Va"])
    rews, done, info = [], False, None
    while not done:
        if info is not None:
            pid.set_reference(info["target"]["roll"], info["target"]["pitch"], info["target"]["Va"])
        obs, r, done, info = env.step(pid.get_action(obs[0], obs[1], obs[2], obs[3:6]))
        rews.append(float(r))
    keep = {k: info.get(k) for k in ("termination", "settling_time", "rise_time", "control_variation", "success", "overshoot")}
    return rews, keep


# ----------------------------------------------------------------------------------------------------------------------
# statistics: tests/turbulence_stats.py (the same code summarised the published files into tests/golden/eval_turbulence_stats.json)
# ----------------------------------------------------------------------------------------------------------------------
def published_stats(intensity, controller="PID"):
    with open(os.path.join(ROOT, "tests", "golden", "eval_turbulence_stats.json")) as f:
        return json.load(f)[controller][intensity]


def distance(ours, pub):
    """Two-sided score: a sum of squared z-like terms (smaller is better), plus the terms themselves."""
    terms = {}
    n = 100.0
    for k in ("roll", "pitch", "Va", "all"):
        p = pub["success_%"][k] / 100.0
        sd = max(np.sqrt(max(p * (1 - p), 0.01 * 0.99) / n), 0.01)
        terms["succ_" + k] = (ours["success_%"][k] / 100.0 - p) / sd / 3.0
    for m in ("rise_time", "settling_time"):
        for k in ("roll", "pitch", "Va"):
            terms[m[:4] + "_" + k] = (ours[m][k] - pub[m][k]) / (0.25 * pub[m][k])
    for k in ("roll", "pitch", "Va"):
        terms["over_" + k] = (ours["overshoot"][k] - pub["overshoot"][k]) / (0.35 * max(pub["overshoot"][k], 10.0))
    terms["cv"] = (ours["control_variation"] - pub["control_variation"]) / (0.25 * pub["control_variation"])
    for key in ("100-150", "200-300", "500-1000"):
        if key in ours["buckets"] and key in pub["buckets"]:
            terms["rew_" + key] = (ours["buckets"][key]["mean"] - pub["buckets"][key]["mean"]) / (0.3 * abs(pub["buckets"][key]["mean"]))
    terms["timeout"] = (ours["length"]["timeout_%"] - pub["length"]["timeout_%"]) / 5.0
    for k in ("d5_std", "d10_std", "d20_std"):   # spread of the reward change over the first steps (x3.8 at severe in the published traces)
        if k in ours.get("early", {}) and k in pub.get("early", {}):
            terms["early_" + k[:-4]] = (ours["early"][k] - pub["early"][k]) / (0.3 * pub["early"][k] + 0.003)
    jo, jp = ours["jitter"].get("30-130"), pub["jitter"].get("30-130")
    if jo and jp:
        terms["jit_lag1"] = (jo["lag1"] - jp["lag1"]) / 0.1
        terms["jit_white"] = (jo["white_Va"] - jp["white_Va"]) / (0.15 * jp["white_Va"] + 0.005)
        terms["jit_walk"] = (jo["walk_Va"] - jp["walk_Va"]) / (0.3 * jp["walk_Va"] + 0.01)
    return float(np.sqrt(np.mean([t * t for t in terms.values()]))), terms


def fmt(s):
    b = s["buckets"]
    return ("succ {:5.1f}/{:5.1f}/{:5.1f}/{:5.1f}  rise {:.2f}/{:.2f}/{:.2f}  settle {:.2f}/{:.2f}/{:.2f}  over {:3.0f}/{:3.0f}/{:3.0f}  cv {:.3f}  "
            "len med {:.0f} p90 {:.0f} t/o {:.0f}%  rew[100-150] {:.3f}/{:.3f}  [200-300] {:.3f}  [500-1000] {:.3f}  d5/d10/d20 {:.3f}/{:.3f}/{:.3f} jitter[30-130] lag1 {:.2f} white {:.3f} walk {:.3f}").format(
        s["success_%"]["roll"], s["success_%"]["pitch"], s["success_%"]["Va"], s["success_%"]["all"],
        s["rise_time"]["roll"], s["rise_time"]["pitch"], s["rise_time"]["Va"],
        s["settling_time"]["roll"], s["settling_time"]["pitch"], s["settling_time"]["Va"],
        s["overshoot"]["roll"], s["overshoot"]["pitch"], s["overshoot"]["Va"], s["control_variation"],
        s["length"]["median"], s["length"]["p90"], s["length"]["timeout_%"],
        b.get("100-150", {}).get("mean", float("nan")), b.get("100-150", {}).get("p5", float("nan")),
        b.get("200-300", {}).get("mean", float("nan")), b.get("500-1000", {}).get("mean", float("nan")),
        s["early"]["d5_std"], s["early"]["d10_std"], s["early"]["d20_std"],
        s["jitter"].get("30-130", {}).get("lag1", float("nan")), s["jitter"].get("30-130", {}).get("white_Va", float("nan")),
        s["jitter"].get("30-130", {}).get("walk_Va", float("nan")))


def run_variant(pool, name, intensity, scen, cfg, tmpdir, seeds):
    v = ss.parse_variant(name)
    jobs = [(v, sc, cfg, tmpdir, intensity, 1000 * s + i) for s in range(seeds) for i, sc in enumerate(scen)]
    res = pool.map(fly, jobs, chunksize=1)
    metrics = {m: {} for m in ("success", "rise_time", "settling_time", "overshoot", "control_variation")}
    for _, info in res:
        for m in metrics:
            val = info[m] if isinstance(info[m], dict) else {"all": info[m]}
            for k, x in val.items():
                metrics[m].setdefault(k, []).append(x)
    if os.environ.get("TSCAN_DUMP"):
        np.save(os.path.join(os.environ["TSCAN_DUMP"], "%s_%s.npy" % (intensity, name)), np.array([r for r, _ in res], dtype=object), allow_pickle=True)
    return tstats.table_stats(metrics, [r for r, _ in res])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="base")
    ap.add_argument("--intensity", default="severe")
    ap.add_argument("--scenarios", type=int, default=100)
    ap.add_argument("--seeds", type=int, default=1)
    ap.add_argument("--jobs", type=int, default=max(1, min(8, os.cpu_count() or 1)))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_turbulence_scan.json"))
    args = ap.parse_args()
    import configs
    import multiprocessing as mp
    cfg = configs.reference_like("examples")
    with open(os.path.join(ROOT, "tests", "golden", "test_set_wind_none.json")) as f:
        scen = json.load(f)[:args.scenarios]
    tmpdir = tempfile.mkdtemp()
    out = {}
    if os.path.exists(args.out):
        with open(args.out) as f:
            out = json.load(f)
    with mp.get_context("fork").Pool(args.jobs) as pool:
        for intensity in args.intensity.split(","):
            pub = published_stats(intensity)
            print("== {} published: {}".format(intensity, fmt(pub)), flush=True)
            for name in args.variants.split(","):
                t0 = time.time()
                s = run_variant(pool, name, intensity, scen, cfg, tmpdir, args.seeds)
                d, terms = distance(s, pub)
                s["distance"], s["terms"], s["episodes"] = d, terms, args.seeds * len(scen)
                out.setdefault(intensity, {})[name] = s
                print("{:>26s} d={:5.2f}: {}  [{:.0f}s]".format(name, d, fmt(s), time.time() - t0), flush=True)
                with open(args.out, "w") as f:
                    json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
