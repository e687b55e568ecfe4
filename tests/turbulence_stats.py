"""Statistics of evaluation results under turbulence (TEST INFRASTRUCTURE) -- shared by the fixture generator
(tests/golden/make_turbulence_stats.py, run on the reference's published examples/evaluations/*.npy), the parity tests and
tools/turbulence_scan.py, so that published and own numbers come out of the same code.

A result is the reference's layout (examples/evaluate_controller.py:44-169): {metric: {state: [per episode]}} plus
"rewards": [per episode [per step]]."""
import numpy as np

BUCKETS = [(0, 20), (20, 50), (50, 100), (100, 150), (150, 200), (200, 300), (300, 500), (500, 1000), (1000, 1500)]
JITTER_WINDOWS = [(30, 130), (130, 300), (300, 600), (600, 1500)]
METRICS = ("success", "rise_time", "settling_time", "overshoot", "control_variation")


def jitter(rewards, a=30, b=130, scale=1.0, va_scaling=25.0, min_len=80):
    """Step-to-step jitter of the reward over steps [a, b): the increments d_k = r_k - r_(k-1), high-passed (15-step moving
    average removed), have variance c0 and lag-1 covariance c1.  A white component of standard deviation s in the reward
    level contributes 2 s^2 to c0 and -s^2 to c1, a random walk with per-step increment q contributes q^2 to c0 only:
    s = sqrt(-c1), q = sqrt(c0 + 2 c1).  Both are returned in units of airspeed (x va_scaling: the reward's airspeed term is
    |e_Va| / 25, examples/fixed_wing_config.json), together with the lag-1 autocorrelation c1 / c0 (-0.5 = white level
    noise, 0 = random walk, +0.5 = a linearly interpolated random walk)."""
    num, cnt = np.zeros(2), 0
    for r in rewards:
        y = np.asarray(r[a:b], dtype=np.float64) * scale
        if len(y) < min_len or y.min() < -50:      # (failed steps carry the reward steps - steps_max)
            continue
        d = np.diff(y)
        d = (d - np.convolve(d, np.ones(15) / 15, mode="same"))[10:-10]
        num += [np.sum(d * d), np.sum(d[1:] * d[:-1])]
        cnt += len(d)
    if cnt == 0:
        return None
    c0, c1 = num / cnt
    s2 = max(-c1, 0.0)
    return {"n": int(cnt), "lag1": float(c1 / c0), "white_Va": va_scaling * float(np.sqrt(s2)),
            "walk_Va": va_scaling * float(np.sqrt(max(c0 - 2 * s2, 0.0)))}


def table_stats(metrics, rewards, steps_max=1500, reward_scale=1.0):
    """metrics: {metric: {state: [per episode]}} in one common episode order; rewards: [per episode [per step]].
    reward_scale: factor that undoes a reward normalisation (the RL evaluations went through VecNormalize)."""
    ok = np.array([bool(x) for x in metrics["success"]["all"]])
    out = {"success_%": {k: 100.0 * float(np.mean([bool(x) for x in vals])) for k, vals in metrics["success"].items()}}

    def masked(vals):
        a = np.array([np.nan if (x is None or not o) else float(x) for x, o in zip(vals, ok)], dtype=np.float64)
        return float(np.nanmean(a)) if np.any(np.isfinite(a)) else float("nan")

    for m, scale in (("rise_time", 0.01), ("settling_time", 0.01), ("overshoot", 100.0)):
        out[m] = {k: masked(vals) * scale for k, vals in metrics[m].items() if k != "all"}
    out["control_variation"] = masked(metrics["control_variation"]["all"])
    # the reference keeps appending the idle env's rewards to its last scenario (evaluate_controller.py:151-153): lengths
    # above steps_max are that artefact
    L = np.array([min(len(r), steps_max) for r in rewards])
    out["length"] = {"median": float(np.median(L)), "p90": float(np.percentile(L, 90)), "mean": float(np.mean(L)),
                     "timeout_%": 100.0 * float(np.mean(L >= steps_max))}
    out["buckets"] = {}
    for a, b in BUCKETS:
        vals = [np.asarray(r[a:min(b, steps_max)], dtype=np.float64) * reward_scale for r in rewards if len(r) > a]
        vals = [v[v > -50] for v in vals]
        if not vals or sum(len(v) for v in vals) == 0:
            continue
        x = np.concatenate(vals)
        out["buckets"]["%d-%d" % (a, b)] = {"alive": len(vals), "mean": float(np.mean(x)), "std": float(np.std(x)),
                                            "p5": float(np.percentile(x, 5))}
    # growth of the spread over the first steps (initial conditions + the earliest effect of the gusts)
    out["early"] = {"d%d_std" % k: float(np.std([(r[k] - r[0]) * reward_scale for r in rewards if len(r) > k and r[k] > -50]))
                    for k in (1, 5, 10, 20)}
    out["jitter"] = {}
    for a, b in JITTER_WINDOWS:
        j = jitter([r[:steps_max] for r in rewards], a, b, scale=reward_scale)
        if j is not None:
            out["jitter"]["%d-%d" % (a, b)] = j
    return out
