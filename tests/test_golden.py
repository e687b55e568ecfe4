"""Golden vectors recorded from the VERBATIM reference gym module (tests/golden/make_golden.py, build container only).

CPU part (always runs): the float64 oracle reproduces every recorded value (pins oracle/gym_restated.py to the
reference on any machine, the reference itself is not needed); the product's host-side config compiler reproduces the
reference's spaces, normalisation defaults and curriculum ranges.
GPU part: the HIP path replays the RNG-free fixtures (explicit initial state and targets) through the C ABI.
"""
import copy
import glob
import json
import math
import os

import numpy as np
import pytest

import parity
from oracle.gym_restated import FixedWingOracle, MTStream

HERE = os.path.dirname(os.path.abspath(__file__))
G1 = sorted(glob.glob(os.path.join(HERE, "golden", "g1_*.json")))


class ScriptedRNG:  # same generator as tests/golden/make_golden.py
    def __init__(self):
        self.k = 0

    def _u(self):
        self.k += 1
        return ((self.k * 0.6180339887498949) % 1.0)

    def uniform(self, low=0.0, high=1.0):
        return low + (high - low) * self._u()

    def normal(self, loc=0.0, scale=1.0):
        return loc + scale * (2.0 * self._u() - 1.0)

    def choice(self, values, p=None):
        u = self._u()
        p = np.full(len(values), 1.0 / len(values)) if p is None else np.asarray(p, dtype=np.float64)
        return values[min(int(np.searchsorted(np.cumsum(p), u, side="right")), len(values) - 1)]


def _nan(x):
    return math.nan if x is None else x


def _arr(x):
    return np.array([[_nan(v) for v in r] if isinstance(r, list) else _nan(r) for r in x], dtype=np.float64)


def _load(path):
    with open(path) as f:
        return json.load(f)


def _int_keys(kw):
    """JSON turns the int keys of config_kw (list item addressing, fixed_wing.py:24-29) into strings."""
    if not isinstance(kw, dict):
        return kw
    return {(int(k) if isinstance(k, str) and k.isdigit() else k): _int_keys(v) for k, v in kw.items()}


@pytest.mark.parametrize("path", G1, ids=[os.path.basename(p)[3:-5] for p in G1])
def test_oracle_reproduces_reference_vectors(path):
    rec = _load(path)
    env = FixedWingOracle(rec["config"], config_kw=_int_keys(rec["config_kw"]), sim_config_kw=rec["sim_config_kw"])
    env.seed(7)
    env.rng = MTStream(ScriptedRNG())
    n = 0
    for ep in rec["episodes"]:
        obs = env.reset(state=ep["state"], target=ep["target"])
        parity.close(obs, _arr(ep["reset_obs"]), 1e-11, 1e-11, "reset obs")
        for k, v in ep["reset_target"].items():
            parity.close(env.target[k], v, 1e-12, 1e-12, "reset target")
        for k, v in ep.get("sim_params", {}).items():   # simulator["model"]: the aircraft sampled for this episode
            parity.close(float(env.simulator.params[k]), v, 1e-13, 1e-13, "sampled parameter " + k)
        if "sim_params_api" in ep:   # get_simulator_parameters (fixed_wing.py:872-888)
            parity.close(np.array(env.get_simulator_parameters(True)), _arr(ep["sim_params_api"]["normalized"]), 1e-12, 1e-12, "normalised parameters")
            parity.close(np.array(env.get_simulator_parameters(False)), _arr(ep["sim_params_api"]["raw"]), 1e-13, 1e-13, "raw parameters")
        for t, st in enumerate(ep["steps"]):
            obs, rew, done, info = env.step(np.array(st["action"]))
            what = "{} step {}".format(os.path.basename(path), t)
            parity.close(obs, _arr(st["obs"]), 1e-10, 1e-10, what + " obs")
            parity.close(rew, st["reward"], 1e-10, 1e-10, what + " reward")
            assert done == st["done"] and info.get("termination") == st["termination"], what
            for k, v in st["target"].items():
                parity.close(info["target"][k], v, 1e-11, 1e-11, what + " target")
            if done:
                for m, vals in st["metrics"].items():
                    for k, v in vals.items():
                        got = info[m][k]
                        if isinstance(v, bool):
                            assert bool(got) == v, (what, m, k)
                        else:
                            parity.close(float(got), _nan(v), 1e-9, 1e-9, what + " metric {}[{}]".format(m, k))
            n += 1
    assert n > 0


def test_config_compiler_reproduces_reference_spaces():
    from gym_fixed_wing.config import EnvConfig
    g3 = _load(os.path.join(HERE, "golden", "g3_spaces.json"))
    for name, rec in g3.items():
        kw = {"observation": {"normalize": rec["normalize"]}} if rec["normalize"] is not None else None
        ec = EnvConfig(copy.deepcopy(rec["config"]), config_kw=kw)
        np.testing.assert_allclose(ec.obs_low, np.array(rec["obs_low"], dtype=np.float64), rtol=1e-12, err_msg=name)
        np.testing.assert_allclose(ec.obs_high, np.array(rec["obs_high"], dtype=np.float64), rtol=1e-12, err_msg=name)
        np.testing.assert_allclose(ec.action_space_low, rec["act_low"], rtol=1e-12)
        np.testing.assert_allclose(ec.action_space_high, rec["act_high"], rtol=1e-12)
        np.testing.assert_allclose(ec.action_scale_to_low, rec["scale_to_low"], rtol=1e-12)
        np.testing.assert_allclose(ec.action_scale_to_high, rec["scale_to_high"], rtol=1e-12)
        for ov, (mean, var) in zip(ec.cfg["observation"]["states"], rec["norm"]):
            assert ov.get("mean", None) == pytest.approx(mean, rel=1e-12) if mean is not None else ov.get("mean", None) is None
            assert ov.get("var", None) == pytest.approx(var, rel=1e-12) if var is not None else ov.get("var", None) is None


def test_config_compiler_reproduces_reference_curriculum():
    from gym_fixed_wing.config import EnvConfig
    g2 = _load(os.path.join(HERE, "golden", "g2_curriculum.json"))
    ec = EnvConfig(copy.deepcopy(g2["config"]))
    for level, rec in g2["levels"].items():
        ec.set_curriculum_level(float(level))
        for name, props in rec["simulator"].items():
            for p, v in props.items():
                got = getattr(ec.state[name], p)
                assert (got is None) == (v is None), (level, name, p)
                if v is not None:
                    assert got == pytest.approx(v, rel=1e-12, abs=1e-15), (level, name, p)
        for name, props in rec["target"]["states"].items():
            for p, v in props.items():
                got = ec.target_props_init["states"][name][p]
                if isinstance(v, (int, float)) and not isinstance(v, bool):
                    assert got == pytest.approx(v, rel=1e-12, abs=1e-15), (level, name, p)
                else:
                    assert got == v


RNG_FREE = ["default", "spec_c5", "default_short", "examples", "mlp", "success_done", "fail_prone", "reward_mix",
            "reward_mix_potential"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", RNG_FREE)
def test_gpu_replays_reference_vectors(case):
    """The HIP path (1-env slice, no auto-reset, explicit state/targets) against the values the verbatim reference
    produced.  fp32 vs float64 over up to 140 free-running steps: 3e-3 abs+rel; integer metrics within one step."""
    from gym_fixed_wing.vec_env import FixedWingVecEnv
    rec = _load(os.path.join(HERE, "golden", "g1_{}.json".format(case)))
    vec = FixedWingVecEnv(rec["config"], num_envs=1, device=0, config_kw=_int_keys(rec["config_kw"]),
                          sim_config_kw=rec["sim_config_kw"], auto_reset=False, as_numpy=True, seed=7)
    tol = 3e-3
    for ep in rec["episodes"]:
        obs = vec.reset(states=ep["state"], targets=ep["target"])
        parity.close(obs[0], _arr(ep["reset_obs"]), tol, tol, "reset obs")
        for t, st in enumerate(ep["steps"]):
            obs, rew, done, infos = vec.step(np.array(st["action"], dtype=np.float32).reshape(1, 3))
            what = "{} step {}".format(case, t)
            info = infos[0]
            assert bool(done[0]) == st["done"], what
            parity.close(rew[0], st["reward"], tol, tol, what + " reward")
            parity.close(obs[0], _arr(st["obs"]), tol, tol, what + " obs")
            for k, v in st["target"].items():
                parity.close(info["target"][k], v, tol, tol, what + " target")
            if st["done"]:
                assert info["termination"] == st["termination"], what
                for m, vals in st["metrics"].items():
                    for k, v in vals.items():
                        got = info[m][k]
                        if isinstance(v, bool):
                            assert bool(got) == v, (what, m, k)
                        elif m in ("rise_time", "settling_time"):
                            g, e_ = float(got), _nan(v)
                            assert math.isnan(g) == math.isnan(e_) and (math.isnan(g) or abs(g - e_) <= 1.0), (what, m, k, g, e_)
                        else:
                            parity.close(float(got), _nan(v), 1e-2, 1e-3, what + " metric {}[{}]".format(m, k))
    vec.close()
