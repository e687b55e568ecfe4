#!/usr/bin/env python3
"""Evaluation protocol of the reference (examples/evaluate_controller.py:44-169 there) on the MI355X env: flies a test
set of (initial state, target) scenarios, all scenarios in one batch, with the PID baseline or a stable-baselines MLP
policy (through the HIP rollout head), and prints the table of examples/README.md:33-47.

    python examples/evaluate_controller.py --controller pid
    python examples/evaluate_controller.py --controller mlp --model tests/golden/mlp_controller.json

Test-set format: JSON list of {"state": {...}, "target": {...}} (converted from the reference's .npy test sets; the one
under tests/golden/ is its examples/test_sets/test_set_wind_none_step20-20-3.npy)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fixed-wing-gym_amd")]

from gym_fixed_wing import evaluate as ev, presets  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--controller", default="pid", choices=["pid", "mlp"])
    ap.add_argument("--test-set", default=os.path.join(ROOT, "tests", "golden", "test_set_wind_none.json"))
    ap.add_argument("--model", default=os.path.join(ROOT, "tests", "golden", "mlp_controller.json"),
                    help="JSON with 'weights' (stable-baselines MlpPolicy parameters) and 'obs_rms' {mean, var}")
    ap.add_argument("--turbulence", default="none", choices=["none", "light", "moderate", "severe"])
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()
    with open(args.test_set) as f:
        scenarios = json.load(f)
    if args.controller == "pid":
        res = ev.evaluate_on_set(scenarios, presets.preset("examples"), turbulence_intensity=args.turbulence, device=args.device)
    else:
        from gym_fixed_wing.actor import DeviceActor, weights_from_stable_baselines
        with open(args.model) as f:
            m = json.load(f)
        actor = DeviceActor(len(scenarios), len(m["obs_rms"]["mean"]), training=False, device=args.device)
        actor.load_policy(weights_from_stable_baselines(m["weights"]))
        actor.set_stats(m["obs_rms"]["mean"], m["obs_rms"]["var"], 1e6)
        res = ev.evaluate_on_set(scenarios, presets.preset("mlp"), turbulence_intensity=args.turbulence, device=args.device,
                                 policy=lambda obs: actor.act(obs.reshape(obs.shape[0], -1).contiguous(), deterministic=True)[1])
    t = ev.summarize(res)
    print("controller {}, {} scenarios, turbulence {}".format(args.controller, len(scenarios), args.turbulence))
    print("success %      roll {roll:6.1f}  pitch {pitch:6.1f}  Va {Va:6.1f}  all {all:6.1f}".format(**t["success_%"]))
    for k, unit in (("rise_time", "s"), ("settling_time", "s"), ("overshoot", "%")):
        print("{:<14s} roll {:6.3f}  pitch {:6.3f}  Va {:6.3f}  [{}]".format(k, t[k]["roll"], t[k]["pitch"], t[k]["Va"], unit))
    print("control variation {:.3f}".format(t["control_variation"]["all"]))


if __name__ == "__main__":
    main()
