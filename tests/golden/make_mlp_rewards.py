#!/usr/bin/env python3
"""Generates tests/golden/eval_res_RL_MLP_none_rewards.json -- BUILD CONTAINER ONLY.

Input: the reference's published evaluation of its shipped MLP policy on the no-wind test set
(gym_fixed_wing/examples/evaluations/eval_res_RL_MLP_none.npy: metrics + per-step rewards of the 100 deterministic episodes) and
the VecNormalize return statistics it was run under (examples/models/mlp_controller/ret_rms.pkl).  The evaluation script wraps
the envs in VecNormalize with training = False (evaluate_controller.py:93-100), so every stored reward is
clip(r / sqrt(ret_rms.var + 1e-8), -10, 10): multiplying by sqrt(var + 1e-8) gives the raw rewards back (no stored value is
anywhere near the clip).  Output: the first 100 un-normalised rewards of every episode, the episode lengths and the table means
-- a second deterministic closed-loop trace of real PyFly 0.1.2 next to the PID one (eval_res_PID_none_rewards.json).

    python tests/golden/make_mlp_rewards.py
"""
import json
import os
import pickle
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/gym_fixed_wing/examples"
KEEP = 100


def load_rms(path):
    # the pickle references stable_baselines.common.running_mean_std.RunningMeanStd (attributes mean, var, count): a stand-in class
    mods = {}
    for name in ("stable_baselines", "stable_baselines.common", "stable_baselines.common.running_mean_std"):
        mods[name] = types.ModuleType(name)

    class RunningMeanStd(object):
        pass
    mods["stable_baselines.common.running_mean_std"].RunningMeanStd = RunningMeanStd
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        with open(path, "rb") as f:
            return pickle.load(f)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def main():
    res = np.load(os.path.join(REF, "evaluations", "eval_res_RL_MLP_none.npy"), allow_pickle=True).item()
    rms = load_rms(os.path.join(REF, "models", "mlp_controller", "ret_rms.pkl"))
    scale = float(np.sqrt(float(rms.var) + 1e-8))
    assert max(abs(x) for r in res["rewards"] for x in r) < 9.9, "a stored reward sits at the VecNormalize clip"
    ok = np.array([bool(v) for v in res["success"]["all"]])
    out = {"_source": "gym_fixed_wing/examples/evaluations/eval_res_RL_MLP_none.npy x sqrt(ret_rms.var + 1e-8) "
                      "(examples/models/mlp_controller/ret_rms.pkl, evaluate_controller.py:93-100); first {} rewards per episode".format(KEEP),
           "reward_scale": scale,
           "episode_lengths": [len(r) for r in res["rewards"]],
           "rewards": [[float(x) * scale for x in r[:KEEP]] for r in res["rewards"]],
           "table": {"success_%": {k: 100.0 * float(np.mean([bool(x) for x in v])) for k, v in res["success"].items()},
                     "settling_time_s": {k: float(np.nanmean(np.where(ok, np.array(v, dtype=float), np.nan))) * 0.01
                                         for k, v in res["settling_time"].items()},
                     "rise_time_s": {k: float(np.nanmean(np.where(ok, np.array(v, dtype=float), np.nan))) * 0.01
                                     for k, v in res["rise_time"].items()},
                     "control_variation": float(np.nanmean(np.where(ok, np.array(res["control_variation"]["all"], dtype=float), np.nan)))}}
    with open(os.path.join(HERE, "eval_res_RL_MLP_none_rewards.json"), "w") as f:
        json.dump(out, f)
    print("scale", scale, "table", out["table"])


if __name__ == "__main__":
    main()
