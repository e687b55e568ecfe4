#!/usr/bin/env python3
"""The shipped MLP policy (tests/golden/mlp_controller.json: weights + VecNormalize statistics of the reference's
examples/models/mlp_controller) flown on the float64 oracle through the reference's evaluation protocol
(examples/evaluate_controller.py:44-169, incl. the un-normalised observation its first action of every episode sees), scored
against the SECOND deterministic closed-loop trace the reference publishes: eval_res_RL_MLP_none.npy, un-normalised into
tests/golden/eval_res_RL_MLP_none_rewards.json (first 100 rewards per episode) by tests/golden/make_mlp_rewards.py.

    python tools/mlp_trace.py --variants base,tau=0.3 [--no-quirk] [--jobs 8]

Variant tokens as tools/structure_scan.py (km, sprop, cdp, tau, P_<param>=...).  CPU only, ~1 min per variant on 8 cores."""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)
import structure_scan as ss  # noqa: E402

with open(os.path.join(ROOT, "tests", "golden", "mlp_controller.json")) as f:
    _M = json.load(f)
_W = {k: np.array(v, dtype=np.float64) for k, v in _M["weights"].items()}
_MEAN = np.array(_M["obs_rms"]["mean"], dtype=np.float64)
_STD = np.sqrt(np.array(_M["obs_rms"]["var"], dtype=np.float64) + 1e-8)


def policy_mean(x):
    h = np.tanh(x @ _W["pi_fc0_w"] + _W["pi_fc0_b"])
    h = np.tanh(h @ _W["pi_fc1_w"] + _W["pi_fc1_b"])
    return h @ _W["pi_w"] + _W["pi_b"]


def fly(args):
    v, sc, cfg, tmpdir, quirk = args
    import oracle.gym_restated as gr
    from gym_fixed_wing import evaluate as ev
    from oracle.gym_restated import FixedWingOracle
    pp, sp = ss.build_files(v, tmpdir)
    ss.VariantPyFly.variant = v
    gr.PyFly = ss.VariantPyFly
    env = FixedWingOracle(cfg, config_kw=ev.evaluation_overrides(False), sim_config_kw={"turbulence": False, "turbulence_intensity": "none"},
                          sim_config_path=sp, sim_parameter_path=pp)
    obs = env.reset(state=sc["state"], target=sc["target"])
    rews, done, first, info = [], False, True, None
    while not done:
        x = np.asarray(obs, dtype=np.float64).reshape(-1)
        xn = x if (quirk and first) else np.clip((x - _MEAN) / _STD, -10.0, 10.0)
        first = False
        obs, r, done, info = env.step(policy_mean(xn))
        rews.append(float(r))
    return rews, {k: info.get(k) for k in ("termination", "settling_time", "rise_time", "control_variation", "success", "overshoot")}


def score(res, pub):
    ok = np.array([bool(i["success"]["all"]) for _, i in res])
    d = np.concatenate([np.abs(np.array(a[:min(len(a), len(b))]) - np.array(b[:min(len(a), len(b))])) for (a, _), b in zip(res, pub["rewards"])])
    lens = np.array([len(a) for a, _ in res], dtype=float)
    out = {"success_all_%": 100.0 * float(ok.mean()),
           "settling_s": {k: float(np.nanmean([i["settling_time"][k] if (o and i["settling_time"].get(k) is not None) else np.nan
                                               for (_, i), o in zip(res, ok)])) * 0.01 for k in ("roll", "pitch", "Va")},
           "control_variation": float(np.nanmean([i["control_variation"]["all"] if o else np.nan for (_, i), o in zip(res, ok)])),
           "mean_abs_dreward_first100": float(d.mean()), "p90_abs_dreward_first100": float(np.percentile(d, 90)),
           "second_step_reward_abs_err": float(np.mean([abs(a[1] - b[1]) for (a, _), b in zip(res, pub["rewards"])])),
           "episode_length_ratio_median": float(np.median(lens / np.array(pub["episode_lengths"], dtype=float))),
           "published": pub["table"]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="base")
    ap.add_argument("--no-quirk", action="store_true", help="normalise the first observation too (NOT what the reference's script does)")
    ap.add_argument("--jobs", type=int, default=max(1, min(8, os.cpu_count() or 1)))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_mlp_trace.json"))
    args = ap.parse_args()
    import configs
    import multiprocessing as mp
    cfg = configs.reference_like("mlp")
    with open(os.path.join(ROOT, "tests", "golden", "test_set_wind_none.json")) as f:
        scen = json.load(f)
    with open(os.path.join(ROOT, "tests", "golden", "eval_res_RL_MLP_none_rewards.json")) as f:
        pub = json.load(f)
    tmpdir = tempfile.mkdtemp()
    out = {}
    if os.path.exists(args.out):
        with open(args.out) as f:
            out = json.load(f)
    with mp.get_context("fork").Pool(args.jobs) as pool:
        for name in args.variants.split(","):
            v = ss.parse_variant(name)
            res = pool.map(fly, [(v, sc, cfg, tmpdir, not args.no_quirk) for sc in scen], chunksize=1)
            s = score(res, pub)
            key = name + ("+noquirk" if args.no_quirk else "")
            out[key] = s
            print("{:>28s}: succ {:.0f}  settle {:.3f}/{:.3f}/{:.3f} (pub {:.3f}/{:.3f}/{:.3f})  cv {:.3f} (pub {:.3f})  |dr| mean {:.4f} p90 {:.4f}  "
                  "r1 err {:.4f}  len ratio {:.3f}".format(key, s["success_all_%"], s["settling_s"]["roll"], s["settling_s"]["pitch"], s["settling_s"]["Va"],
                                                         pub["table"]["settling_time_s"]["roll"], pub["table"]["settling_time_s"]["pitch"],
                                                         pub["table"]["settling_time_s"]["Va"], s["control_variation"], pub["table"]["control_variation"],
                                                         s["mean_abs_dreward_first100"], s["p90_abs_dreward_first100"], s["second_step_reward_abs_err"],
                                                         s["episode_length_ratio_median"]), flush=True)
            with open(args.out, "w") as f:
                json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
