#!/usr/bin/env python3
"""Scans propulsion/drag constants against the two published closed-loop results (PID and shipped MLP policy)."""
import json, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from gym_fixed_wing import evaluate as ev, presets
from gym_fixed_wing.config import DEFAULT_PARAMETERS
scen = json.load(open(os.path.join(ROOT, "tests", "golden", "test_set_wind_none.json")))
m = json.load(open(os.path.join(ROOT, "tests", "golden", "mlp_controller.json")))
pid_pub = json.load(open(os.path.join(ROOT, "tests", "golden", "eval_res_PID_none.json")))
W = {k: torch.tensor(v, dtype=torch.float32, device="cuda") for k, v in m["weights"].items()}
mean = torch.tensor(m["obs_rms"]["mean"], dtype=torch.float32, device="cuda")
std = torch.sqrt(torch.tensor(m["obs_rms"]["var"], dtype=torch.float32, device="cuda") + 1e-8)
def policy(obs):
    x = ((obs.reshape(obs.shape[0], -1) - mean) / std).clamp(-10, 10)
    h = torch.tanh(x @ W["pi_fc0_w"] + W["pi_fc0_b"]); h = torch.tanh(h @ W["pi_fc1_w"] + W["pi_fc1_b"])
    return h @ W["pi_w"] + W["pi_b"]
base = json.load(open(DEFAULT_PARAMETERS))
tmp = tempfile.mkdtemp()
tv = np.array([s["target"]["Va"] for s in scen]); v0 = np.array([s["state"]["Va"] for s in scen])
def run(over):
    p = dict(base); p.update(over)
    path = os.path.join(tmp, "p.json"); json.dump(p, open(path, "w"))
    r1 = ev.evaluate_on_set(scen, presets.preset("examples"), device=0, sim_parameter_path=path)
    r2 = ev.evaluate_on_set(scen, presets.preset("mlp"), policy=policy, device=0, sim_parameter_path=path)
    t1, t2 = ev.summarize(r1), ev.summarize(r2)
    l1 = np.array([len(r) for r in r1["rewards"]]); l2 = np.array([len(r) for r in r2["rewards"]])
    e1 = np.mean(np.abs(l1 - np.array(pid_pub["episode_lengths"])) / np.array(pid_pub["episode_lengths"]))
    e2 = np.mean(np.abs(l2 - np.array(m["published_episode_lengths"])) / np.array(m["published_episode_lengths"]))
    return t1, t2, e1, e2, r2
t1, t2, e1, e2, r2 = run({})
fail = np.array([not bool(x) for x in r2["success"]["Va"]])
print("MLP Va failures: target Va of failing", np.round(np.sort(tv[fail]), 1), " succeeding range", tv[~fail].min(), tv[~fail].max())
print("%-34s PID succ %5.1f VaSettle %.2f lenerr %.3f | MLP succ %5.1f VaSettle %.2f lenerr %.3f" % ("base", t1["success_%"]["all"], t1["settling_time"]["Va"], e1, t2["success_%"]["all"], t2["settling_time"]["Va"], e2))
import itertools
for km, sp in itertools.product((36, 38, 40, 42, 44, 48), (0.07, 0.085, 0.1018, 0.12, 0.14)):
    t1, t2, e1, e2, _ = run({"k_motor": km, "S_prop": sp})
    print("k_motor %4.1f S_prop %.4f  PID succ %5.1f VaRise %.2f VaSettle %.2f CV %.3f lenerr %.3f | MLP succ %5.1f VaRise %.2f VaSettle %.2f CV %.3f lenerr %.3f | sum %.3f" % (km, sp, t1["success_%"]["all"], t1["rise_time"]["Va"], t1["settling_time"]["Va"], t1["control_variation"]["all"], e1, t2["success_%"]["all"], t2["rise_time"]["Va"], t2["settling_time"]["Va"], t2["control_variation"]["all"], e2, e1+e2), flush=True)
