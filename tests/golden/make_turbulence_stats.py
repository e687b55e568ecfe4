#!/usr/bin/env python3
"""Generates tests/golden/eval_turbulence_stats.json -- BUILD CONTAINER ONLY.

Inputs: the evaluation results the reference publishes (gym_fixed_wing/examples/evaluations/eval_res_<controller>_<intensity>.npy,
the per-episode data behind the table of examples/README.md:33-47): metrics per episode and the raw per-step rewards of the
100 test-set episodes, for the PID baseline and the two shipped RL policies under none / light / moderate / severe turbulence.
Output: their summary statistics (tests/turbulence_stats.py) -- table means, reward level by time bucket, early spread, and
the step-to-step jitter decomposition -- a few hundred numbers; the traces themselves stay in the reference.

    python tests/golden/make_turbulence_stats.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import turbulence_stats as ts  # noqa: E402

REF = "/root/reference/gym_fixed_wing/examples/evaluations"


def main():
    out = {"_source": "gym_fixed_wing/examples/evaluations/eval_res_<controller>_<intensity>.npy (reference), summarised by "
                      "tests/turbulence_stats.py; RL rewards un-normalised by matching the mean first-step reward to the PID runs'"}
    r0_pid = None
    for ctl in ("PID", "RL_MLP", "RL_CNN"):
        out[ctl] = {}
        for intensity in ("none", "light", "moderate", "severe"):
            r = np.load(os.path.join(REF, "eval_res_%s_%s.npy" % (ctl, intensity)), allow_pickle=True).item()
            r0 = float(np.mean([x[0] for x in r["rewards"]]))
            if ctl == "PID" and intensity == "none":
                r0_pid = r0
            # VecNormalize scaled the RL rewards by 1 / sqrt(ret_rms.var + eps); the first reward of an episode does not
            # depend on the controller to first order (one step of actuation), which recovers the factor
            scale = 1.0 if ctl == "PID" else r0_pid / r0
            s = ts.table_stats({m: r[m] for m in ts.METRICS}, r["rewards"], reward_scale=scale)
            s["reward_scale"] = scale
            out[ctl][intensity] = s
    with open(os.path.join(HERE, "eval_turbulence_stats.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for ctl in ("PID", "RL_MLP"):
        for intensity in ("light", "moderate", "severe"):
            print(ctl, intensity, out[ctl][intensity]["jitter"])


if __name__ == "__main__":
    main()
