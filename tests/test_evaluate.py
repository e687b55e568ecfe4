"""Evaluation protocol (reference examples/evaluate_controller.py:44-169) on the vectorised
env with the batched PID baseline.

CPU: a few scenarios on the host-emulation build against the same protocol run with the float64 oracle env + its scalar
PID.  GPU: the full shipped test set (100 scenarios, all in parallel) against the results the reference published for
real PyFly 0.1.2 (examples/evaluations/eval_res_PID_none.npy incl. its 25 878 per-step rewards, examples/README.md:33-47):
GATING bands on the per-step rewards, episode lengths, settling / rise times and -- under all four turbulence
settings -- the success rates (see also tests/test_simulator_pins.py and DESIGN.md section 2)."""
import json
import os

import numpy as np
import pytest

import configs
from gym_fixed_wing import evaluate as ev
from oracle.gym_restated import FixedWingOracle
from oracle.pyfly_restated import PIDController

HERE = os.path.dirname(os.path.abspath(__file__))


def _scenarios():
    with open(os.path.join(HERE, "golden", "test_set_wind_none.json")) as f:
        return json.load(f)


def _oracle_eval(scenarios, cfg):
    kw = ev.evaluation_overrides(True)
    out = []
    for sc in scenarios:
        env = FixedWingOracle(cfg, config_kw=kw, sim_config_kw={"turbulence": False, "turbulence_intensity": "none"})
        obs = env.reset(state=sc["state"], target=sc["target"])
        pid = PIDController(env.simulator.dt)
        pid.set_reference(sc["target"]["roll"], sc["target"]["pitch"], sc["target"]["Va"])
        rews, done, info = [], False, None
        while not done:
            if info is not None:
                pid.set_reference(info["target"]["roll"], info["target"]["pitch"], info["target"]["Va"])
            obs, r, done, info = env.step(pid.get_action(obs[0], obs[1], obs[2], obs[3:6]))
            rews.append(r)
        out.append((rews, info))
    return out


def test_protocol_matches_oracle_on_emulated_kernels():
    from emu.host_backend import HostBackend, build_emu
    cfg = configs.reference_like("examples")
    scen = _scenarios()[:5]
    res = ev.evaluate_on_set(scen, cfg, as_numpy=True, _backend=HostBackend(), _lib_path=build_emu())
    want = _oracle_eval(scen, cfg)
    for i, (rews, info) in enumerate(want):
        assert abs(len(res["rewards"][i]) - len(rews)) <= 1, (i, len(res["rewards"][i]), len(rews))
        n = min(len(rews), len(res["rewards"][i]))
        np.testing.assert_allclose(res["rewards"][i][:n], rews[:n], atol=5e-3)
        assert res["termination"][i] == info["termination"] == "success"
        assert bool(res["success"]["all"][i]) is True
        for k in ("roll", "pitch", "Va"):
            assert abs(res["settling_time"][k][i] - info["settling_time"][k]) <= 1
    table = ev.summarize(res)
    assert table["success_%"]["all"] == 100.0 and 1.0 < table["settling_time"]["roll"] < 3.5


@pytest.mark.gpu
def test_pid_baseline_on_shipped_test_set_reports_distance_to_published_results():
    cfg = configs.reference_like("examples")
    scen = _scenarios()
    res = ev.evaluate_on_set(scen, cfg, device=0)
    table = ev.summarize(res)
    with open(os.path.join(HERE, "golden", "eval_res_PID_none.json")) as f:
        pub = json.load(f)
    with open(os.path.join(HERE, "golden", "eval_res_PID_none_rewards.json")) as f:
        pub_rewards = json.load(f)
    lengths = np.array([len(r) for r in res["rewards"]])
    pub_len = np.array(pub["episode_lengths"])
    first = np.array([r[0] for r in res["rewards"]])
    dr = np.concatenate([np.abs(np.array(a[:min(len(a), len(b))]) - np.array(b[:min(len(a), len(b))]))
                         for a, b in zip(res["rewards"], pub_rewards)])
    report = {
        "ours": table,
        "published_README_PID_none": {"success_%": 100, "rise_time": [1.337, 0.226, 1.016],
                                      "settling_time": [2.018, 1.294, 2.203], "overshoot_%": [3, 9, 29],
                                      "control_variation": 0.291},
        "per_step_reward_abs_err_mean": float(dr.mean()), "per_step_reward_abs_err_p90": float(np.percentile(dr, 90)),
        "steps_compared": int(dr.size),
        "episode_length_ratio_median": float(np.median(lengths / pub_len)),
        "episode_length_rel_err_mean": float(np.mean(np.abs(lengths - pub_len) / pub_len)),
        "episode_length_rel_err_p90": float(np.percentile(np.abs(lengths - pub_len) / pub_len, 90)),
        "first_step_reward_max_abs_err": float(np.max(np.abs(first - np.array(pub["first_rewards"])))),
    }
    print(json.dumps(report, indent=1))
    os.makedirs(os.path.join(os.path.dirname(HERE), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(HERE), "gpurun_out", "pid_eval_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    # GATING bands against the data the reference ships (round 1 reported these numbers without asserting them: mean
    # per-step error 0.037, length p90 0.46, Va settling 1.53 s); see tests/test_simulator_pins.py for the CPU counterpart
    assert table["success_%"] == {"roll": 100.0, "pitch": 100.0, "Va": 100.0, "all": 100.0}    # published 100/100/100/100
    assert report["first_step_reward_max_abs_err"] < 1e-3        # kinematics / error / reward plumbing agree
    assert report["per_step_reward_abs_err_mean"] < 0.015 and report["per_step_reward_abs_err_p90"] < 0.030
    assert report["episode_length_rel_err_mean"] < 0.13 and report["episode_length_rel_err_p90"] < 0.28
    assert 0.9 < report["episode_length_ratio_median"] < 1.1
    for k, pubv, tol in (("roll", 2.018, 0.05), ("pitch", 1.294, 0.06), ("Va", 2.203, 0.08)):
        assert abs(table["settling_time"][k] - pubv) <= tol * pubv, (k, table["settling_time"][k], pubv)
    for k, pubv, tol in (("roll", 1.337, 0.12), ("Va", 1.016, 0.10)):
        assert abs(table["rise_time"][k] - pubv) <= tol * pubv, (k, table["rise_time"][k], pubv)
    assert abs(table["rise_time"]["pitch"] - 0.226) < 0.08       # 0.23 s = 23 steps: +-8 steps
    assert abs(table["control_variation"]["all"] - 0.291) <= 0.2 * 0.291


@pytest.mark.gpu
def test_shipped_mlp_controller_flies_the_shipped_test_set():
    """Drop-in check at system level: the MLP policy the reference ships (trained on real PyFly 0.1.2,
    examples/models/mlp_controller) controls THIS simulator on the shipped test set through the evaluation protocol.
    Weights and VecNormalize statistics come from tests/golden/mlp_controller.json (converted data files)."""
    import torch
    with open(os.path.join(HERE, "golden", "mlp_controller.json")) as f:
        m = json.load(f)
    W = {k: torch.tensor(v, dtype=torch.float32, device="cuda") for k, v in m["weights"].items()}
    mean = torch.tensor(m["obs_rms"]["mean"], dtype=torch.float32, device="cuda")
    std = torch.sqrt(torch.tensor(m["obs_rms"]["var"], dtype=torch.float32, device="cuda") + 1e-8)

    def raw_policy(x):  # stable-baselines MlpPolicy, deterministic action = mean
        h = torch.tanh(x.reshape(x.shape[0], -1) @ W["pi_fc0_w"] + W["pi_fc0_b"])
        h = torch.tanh(h @ W["pi_fc1_w"] + W["pi_fc1_b"])
        return h @ W["pi_w"] + W["pi_b"]

    def policy(obs):  # VecNormalize (clip 10) in front of it
        return raw_policy(((obs.reshape(obs.shape[0], -1) - mean) / std).clamp(-10, 10))

    cfg = configs.reference_like("mlp")
    # the reference's protocol to the letter: the first action of every episode is computed from the UN-normalised reset
    # observation (evaluate_controller.py:118); the published per-step rewards carry that step
    res = ev.evaluate_on_set(_scenarios(), cfg, policy=policy, first_step_policy=raw_policy, device=0)
    table = ev.summarize(res)
    lengths = np.array([len(r) for r in res["rewards"]])
    with open(os.path.join(HERE, "golden", "eval_res_RL_MLP_none_rewards.json")) as f:
        pubm = json.load(f)
    dm = np.concatenate([np.abs(np.array(a[:min(len(a), len(b))]) - np.array(b[:min(len(a), len(b))]))
                         for a, b in zip(res["rewards"], pubm["rewards"])])
    second = float(np.mean([abs(a[1] - b[1]) for a, b in zip(res["rewards"], pubm["rewards"])]))
    print("MLP trace (second pin): mean |dr| first 100 steps {:.4f} p90 {:.4f}, second-step reward error {:.5f}".format(
        dm.mean(), np.percentile(dm, 90), second))
    # GATES on the second deterministic trace of real PyFly (un-normalised with the shipped ret_rms.pkl, tests/golden/make_mlp_rewards.py)
    assert second < 2e-3                                              # (0.023 when the first observation is normalised as well)
    assert dm.mean() < 0.016 and np.percentile(dm, 90) < 0.035        # measured 0.011 / 0.027 on the float64 oracle
    assert abs(table["control_variation"]["all"] - 0.410) <= 0.20 * 0.410, table["control_variation"]   # 0.36 (0.29 without the raw first step)
    report = {"ours": table, "published_README_RL_MLP_none": {"success_%": 100, "rise_time": [1.395, 0.336, 0.959],
                                                              "settling_time": [2.085, 1.675, 2.308],
                                                              "overshoot_%": [5, 25, 20], "control_variation": 0.410},
              "episode_length_ratio_median": float(np.median(lengths / np.array(m["published_episode_lengths"])))}
    print(json.dumps(report, indent=1))
    with open(os.path.join(os.path.dirname(HERE), "gpurun_out", "mlp_eval_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    assert table["success_%"]["all"] >= 97.0 and min(table["success_%"].values()) >= 97.0    # published 100/100/100/100
    for k, pubv, tol in (("roll", 2.085, 0.08), ("pitch", 1.675, 0.12), ("Va", 2.308, 0.22)):
        assert abs(table["settling_time"][k] - pubv) <= tol * pubv, (k, table["settling_time"][k], pubv)
    assert 0.85 < report["episode_length_ratio_median"] < 1.15
    # the same controller through the HIP rollout head (fwg_actor_act, statistics frozen, deterministic): the matrix-core
    # MLP must fly the same episodes as the torch fp32 formulation above
    from gym_fixed_wing.actor import DeviceActor, weights_from_stable_baselines
    actor = DeviceActor(len(_scenarios()), 12, training=False, device=0)
    actor.load_policy(weights_from_stable_baselines(m["weights"]))
    actor.set_stats(m["obs_rms"]["mean"], m["obs_rms"]["var"], 1e6)
    res2 = ev.evaluate_on_set(_scenarios(), cfg, policy=lambda obs: actor.act(obs.reshape(obs.shape[0], -1).contiguous(), deterministic=True)[1],
                              first_step_policy=raw_policy, device=0)
    table2 = ev.summarize(res2)
    lengths2 = np.array([len(r) for r in res2["rewards"]])
    print("HIP head: success", table2["success_%"], "episodes with another length:", int((lengths2 != lengths).sum()))
    assert table2["success_%"]["all"] == table["success_%"]["all"]
    assert np.mean(np.abs(lengths2 - lengths)) < 2.0
    np.testing.assert_allclose(table2["settling_time"]["all"], table["settling_time"]["all"], rtol=0.02)


@pytest.mark.gpu
def test_published_table_under_all_four_turbulence_settings():
    """examples/evaluate_controller.py:78 sets turbulence_intensity in {none, light, moderate, severe}; the reference
    publishes the per-episode results of the PID baseline and of the shipped MLP policy for each
    (examples/evaluations/*.npy; their means are the table of examples/README.md:33-47; summarised into
    tests/golden/eval_turbulence_stats.json by tests/golden/make_turbulence_stats.py).  All four settings are flown for both
    controllers on the shipped no-wind scenarios (the reference's own sets for the three turbulence settings,
    test_set_wind_{light,moderate,severe}, are missing blobs) and gated TWO-SIDEDLY on everything those data pin:

      * none and light rows: success rates within the binomial sampling error, settling times within 6 % (MLP: 8 / 10 / 20 %),
        control variation within 20 % -- published and ours;
      * the turbulence itself, all three intensities, both controllers: the white jitter the gusts put on the airspeed
        ((K_u / T_u) dt sqrt(pi / dt): 0.043 / 0.081 / ~0.13 m/s) within 15 % (moderate 25 %, severe 30 %) and its lag-1 signature within
        0.1 at light and moderate; a simulator without turbulence (or with the MIL-F-8785C signal itself as the gust,
        `turbulence_output: filter`) fails these.

    The moderate / severe TABLE rows are recorded next to the published ones (gpurun_out/eval_table.json) but not gated: the
    published traces of those settings start from other initial conditions than the shipped set (their spread over the first
    5 / 10 / 20 steps is 3.8x / 3.9x / 3.8x the no-wind set's at severe, proportional to the intensity from the first step on
    -- before any gust filter has built up --, and two episodes end by an angular-rate constraint within 0.36 s), so their
    success rates and settling times are properties of the missing scenario sets."""
    import turbulence_stats as ts
    from gym_fixed_wing.actor import DeviceActor, weights_from_stable_baselines
    with open(os.path.join(HERE, "golden", "mlp_controller.json")) as f:
        m = json.load(f)
    with open(os.path.join(HERE, "golden", "eval_turbulence_stats.json")) as f:
        pub = json.load(f)
    scen = _scenarios()
    actor = DeviceActor(len(scen), 12, training=False, device=0)
    actor.load_policy(weights_from_stable_baselines(m["weights"]))
    actor.set_stats(m["obs_rms"]["mean"], m["obs_rms"]["var"], 1e6)
    mlp = lambda obs: actor.act(obs.reshape(obs.shape[0], -1).contiguous(), deterministic=True)[1]
    import torch
    W = {k: torch.tensor(v, dtype=torch.float32, device="cuda") for k, v in m["weights"].items()}

    def mlp_raw(x):   # the first action of every episode: the reference hands the UN-normalised reset observation to the model
        h = torch.tanh(x.reshape(x.shape[0], -1) @ W["pi_fc0_w"] + W["pi_fc0_b"])
        h = torch.tanh(h @ W["pi_fc1_w"] + W["pi_fc1_b"])
        return h @ W["pi_w"] + W["pi_b"]
    table = {}
    for intensity in ("none", "light", "moderate", "severe"):
        table[intensity] = {}
        for name, cfg_kind, policy in (("PID", "examples", None), ("RL_MLP", "mlp", mlp)):
            metrics = {k: {} for k in ts.METRICS}
            rewards = []
            for seed in ((0, 1, 2) if intensity != "none" else (0,)):
                res = ev.evaluate_on_set(scen, configs.reference_like(cfg_kind), policy=policy, device=0, seed=seed,
                                         turbulence_intensity=intensity, first_step_policy=None if policy is None else mlp_raw)
                for k in ts.METRICS:
                    for st, vals in res[k].items():
                        metrics[k].setdefault(st, []).extend(vals)
                rewards += res["rewards"]
            row = ts.table_stats(metrics, rewards)
            row["episodes"] = len(rewards)
            row["published"] = {k: pub[name][intensity][k] for k in ("success_%", "settling_time", "rise_time", "overshoot",
                                                                      "control_variation", "jitter", "early", "length")}
            table[intensity][name] = row
    os.makedirs(os.path.join(os.path.dirname(HERE), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(HERE), "gpurun_out", "eval_table.json"), "w") as f:
        json.dump(table, f, indent=1)
    for intensity, rows in table.items():
        for name, row in rows.items():
            p_ = row["published"]
            print(intensity, name, "success", row["success_%"], "published", p_["success_%"], "| settling",
                  {k: round(v, 3) for k, v in row["settling_time"].items()}, "published", {k: round(v, 3) for k, v in p_["settling_time"].items()},
                  "| cv %.3f published %.3f" % (row["control_variation"], p_["control_variation"]),
                  "| jitter", row["jitter"].get("30-130"), "published", p_["jitter"].get("30-130"))
    for intensity in ("none", "light"):
        for name, row in table[intensity].items():
            p_ = row["published"]
            for k, pv in p_["success_%"].items():
                p = pv / 100.0
                band = max(3.0 * np.sqrt(p * (1 - p) / 100.0 + p * (1 - p) / row["episodes"]), 0.04)
                assert abs(row["success_%"][k] / 100.0 - p) <= band, (intensity, name, k, row["success_%"][k], pv)
            for k, pv in p_["settling_time"].items():
                tol = {"roll": 0.08, "pitch": 0.10, "Va": 0.20}[k] if name == "RL_MLP" else 0.06
                assert abs(row["settling_time"][k] - pv) <= tol * pv, (intensity, name, k, row["settling_time"][k], pv)
        row = table[intensity]["PID"]
        assert abs(row["control_variation"] - row["published"]["control_variation"]) <= 0.20 * row["published"]["control_variation"], \
            (intensity, row["control_variation"], row["published"]["control_variation"])
    for name in ("PID", "RL_MLP"):
        assert table["none"][name]["jitter"]["30-130"]["white_Va"] < 0.01
        for intensity, tol in (("light", 0.15), ("moderate", 0.25)):
            got, want = table[intensity][name]["jitter"]["30-130"], table[intensity][name]["published"]["jitter"]["30-130"]
            assert abs(got["white_Va"] - want["white_Va"]) <= tol * want["white_Va"], (name, intensity, got, want)
            assert abs(got["lag1"] - want["lag1"]) <= 0.10, (name, intensity, got, want)
            assert got["walk_Va"] <= want["walk_Va"] + 0.02, (name, intensity, got, want)
    # severe: the published early window is dominated by the violent transients of its (missing) scenario set; the PID runs'
    # later windows give the gust level (0.128 / 0.143 / 0.135 m/s at 130-300 / 300-600 / 600-1500 steps)
    got = table["severe"]["PID"]["jitter"]["30-130"]["white_Va"]
    want = float(np.mean([table["severe"]["PID"]["published"]["jitter"][w]["white_Va"] for w in ("130-300", "300-600", "600-1500")]))
    assert abs(got - want) <= 0.30 * want, (got, want)
