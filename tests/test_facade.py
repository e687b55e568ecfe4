"""The reference-compatible single-env class (gym_fixed_wing.fixed_wing.FixedWingAircraft) and the VecEnv surface used
by the reference's example scripts, exercised on the host-emulation build and (-m gpu) on the MI355X."""
import os

import numpy as np
import pytest

import configs
import parity
from emu.host_backend import HostBackend, build_emu
from gym_fixed_wing.fixed_wing import FixedWingAircraft
from gym_fixed_wing.vec_env import FixedWingVecEnv

STATE = {"roll": 0.3, "pitch": -0.1, "yaw": 0.5, "omega_p": 0.0, "omega_q": 0.0, "omega_r": 0.0, "position_n": 0.0,
         "position_e": 0.0, "position_d": -100.0, "velocity_u": 20.0, "velocity_v": 0.0, "velocity_w": 1.0,
         "elevator": 0.0, "aileron": 0.0, "throttle": 0.5, "wind_n": 0.0, "wind_e": 0.0, "wind_d": 0.0}
TARGET = {"roll": 0.0, "pitch": 0.05, "Va": 21.0}


BACKENDS = [pytest.param("emu"), pytest.param("gpu", marks=pytest.mark.gpu)]


def _kw(backend):
    """Constructor keywords selecting where the kernels run: host emulation (CPU suite) or the real device."""
    if backend == "emu":
        return {"_backend": HostBackend(), "_lib_path": build_emu()}
    return {"device": 0}


@pytest.mark.parametrize("backend", BACKENDS)
def test_single_env_api_matches_oracle(backend, tmp_path):
    cfg = configs.default()
    ckw = {"steps_max": 40}
    env = FixedWingAircraft(cfg, config_kw=ckw, **_kw(backend))
    orc = parity.make_oracles(cfg, 1, 0, config_kw=ckw)[0]
    assert env.observation_space.shape == (14,) and env.action_space.shape == (3,)
    assert env.seed(0) == [0]
    obs = env.reset(state=STATE, target=TARGET)
    want = orc.reset(state=STATE, target=TARGET)
    assert obs.dtype == np.float64 and obs.shape == (14,)
    np.testing.assert_allclose(obs, want, atol=1e-5)
    assert env.target == pytest.approx(TARGET)
    assert env.simulator.state["roll"].value == pytest.approx(0.3, abs=1e-6)
    rng = np.random.default_rng(0)
    done = False
    steps = 0
    env.training = False
    while not done:
        a = rng.uniform(-1, 1, 3)
        obs, rew, done, info = env.step(a)
        o2, r2, d2, i2 = orc.step(a)
        steps += 1
        np.testing.assert_allclose(obs, o2, atol=2e-3, rtol=2e-3)
        assert rew == pytest.approx(r2, abs=2e-3) and done == d2
        assert set(info["target"]) == {"roll", "pitch", "Va"}
    assert steps == 40 and info["termination"] == "steps" and env.steps_count == 40
    assert set(info["success"]) == {"roll", "pitch", "Va", "all"}
    assert env.get_metric("total_error")["Va"] == pytest.approx(i2["total_error"]["Va"], rel=1e-2)
    assert len(env.history["action"]) == 40 and len(env.history["target"]["Va"]) == 41
    assert len(env.simulator.state["roll"].history) == 41
    # PyFly's actuator histories: "value" one entry per record, "command" one constrained command per step
    # (read by the reference at fixed_wing.py:828 and :1110)
    for n, k in (("elevator", 0), ("aileron", 1), ("throttle", 2)):
        h = env.simulator.state[n].history
        assert len(h["value"]) == 41 and len(h["command"]) == 40
        np.testing.assert_allclose(h["command"], orc.simulator.state[n].history["command"], atol=1e-6)
    cv = np.sum(np.abs(np.diff([env.simulator.state[n].history["command"] for n in ("elevator", "aileron", "throttle")], axis=1)))
    assert cv / (3 * env.simulator.dt * 39) == pytest.approx(env.get_metric("control_variation")["all"], rel=2e-3)
    with pytest.raises(AssertionError):
        env.step(np.array([np.nan, 0, 0]))
    png = str(tmp_path / "render" / "ep.png")
    env.render(show=False, close=True, save_path=png)
    assert os.path.getsize(png) > 10000
    npy = str(tmp_path / "hist.npy")
    env.save_history(npy, ["roll", "Va"])
    saved = np.load(npy, allow_pickle=True).item()
    assert len(saved["roll"]) == 41 and len(saved["Va_target"]) == 41
    rec = env.get_initial_state()
    assert rec["state"]["velocity_u"] == pytest.approx(20.0) and rec["target"]["Va"] == pytest.approx(21.0)
    # while training, render is deferred to the next reset (fixed_wing.py:584-587)
    env.training = True
    assert env.render(show=False, save_path=str(tmp_path / "deferred.png")) is None and env.render_on_reset
    env.reset()
    assert os.path.exists(str(tmp_path / "deferred.png")) and not env.render_on_reset
    env.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_vecenv_contract(backend):
    cfg = configs.default()
    vec = FixedWingVecEnv(cfg, num_envs=4, config_kw={"steps_max": 6}, as_numpy=True, **_kw(backend))
    assert vec.num_envs == 4 and vec.observation_space.shape == (14,)
    obs = vec.reset()
    assert obs.shape == (4, 14) and obs.dtype == np.float32
    vec.step_async(np.zeros((4, 3), dtype=np.float32))
    obs, rew, done, infos = vec.step_wait()
    assert rew.shape == (4,) and done.shape == (4,) and len(infos) == 4 and "target" in infos[0]
    for _ in range(5):
        obs, rew, done, infos = vec.step(np.zeros((4, 3), dtype=np.float32))
    assert done.all()
    info = infos[2]
    assert info["termination"] == "steps" and info["TimeLimit.truncated"] is True
    assert info["terminal_observation"].shape == (14,)
    assert not np.allclose(info["terminal_observation"], obs[2])        # obs is the first one of the next episode
    steps = vec.get_state(["steps_count"])["steps_count"]
    assert (steps == 0).all()
    assert vec.get_attr("cfg")[0]["steps_max"] == 6 and vec.get_attr("simulator")[0].dt == 0.01
    vec.env_method("set_curriculum_level", 0.25)
    assert vec.env_config.state["roll"].init_max == pytest.approx(np.radians(110 * 0.25))
    out = vec.env_method("reset", indices=1, state={"roll": 0.2}, target={"roll": 0.1, "pitch": 0.0, "Va": 20.0})
    assert out[0].shape == (14,) and out[0][0] == pytest.approx(0.2, abs=1e-6) and out[0][6] == pytest.approx(0.1, abs=1e-6)
    vec.set_attr("training", False)
    assert vec.training is False
    vec.check_actions = True
    with pytest.raises(AssertionError):
        vec.step(np.full((4, 3), np.nan, dtype=np.float32))
    vec.close()


@pytest.mark.parametrize("form", ["absolute", "potential"])
@pytest.mark.parametrize("backend", BACKENDS)
def test_goal_env_reproduces_the_reference_goal_env(backend, form):
    """FixedWingAircraftGoal (fixed_wing.py:1165-1277): dict observations, goal limits and compute_reward for substituted
    (relabelled) goals against vectors recorded from the VERBATIM reference class (tests/golden/g4_goal_*.json,
    tests/golden/make_golden.py).  Observations / step rewards come from the kernels (fp32: 2e-3 as elsewhere); the
    relabelled rewards are evaluated on the host in float64 on the kernel's fp32 states."""
    import json
    from gym_fixed_wing.fixed_wing import FixedWingAircraftGoal
    with open(os.path.join(os.path.dirname(__file__), "golden", "g4_goal_{}.json".format(form))) as f:
        rec = json.load(f)
    env = FixedWingAircraftGoal(rec["config"], **_kw(backend))
    env.seed(7)
    assert set(env.observation_space.spaces) == {"desired_goal", "achieved_goal", "observation"}
    assert env.observation_space.spaces["achieved_goal"].shape == (3,)
    obs = env.reset(state=rec["state"], target=rec["target"])
    for k in ("observation", "achieved_goal", "desired_goal"):
        np.testing.assert_allclose(obs[k], np.array(rec["reset_obs"][k], dtype=np.float64), atol=1e-5, err_msg=k)
    lo, hi = env.get_goal_limits()
    np.testing.assert_allclose(lo, rec["goal_limits"][0], rtol=1e-12)
    np.testing.assert_allclose(hi, rec["goal_limits"][1], rtol=1e-12)
    for st in rec["steps"]:
        o, r, d, info = env.step(np.array(st["action"]))
        for k in ("observation", "achieved_goal", "desired_goal"):
            np.testing.assert_allclose(o[k], np.array(st["obs"][k], dtype=np.float64), atol=2e-3, rtol=2e-3, err_msg=k)
        assert r == pytest.approx(st["reward"], abs=2e-3) and d == st["done"]
    for rl in rec["relabel"]:
        got = env.compute_reward(np.array(rl["achieved"]), np.array(rl["desired"]), rl["info"])
        assert got == pytest.approx(rl["reward"], abs=2e-4), (rl["info"]["step"], got, rl["reward"])
    env.close()


@pytest.mark.parametrize("form", ["absolute", "potential"])
@pytest.mark.parametrize("backend", BACKENDS)
def test_goal_compute_reward_agrees_with_the_kernel_reward(backend, form):
    """compute_reward is a second reward implementation (host, float64; it has to be: a hindsight buffer calls it with
    substituted goals long after the step).  Its anchor to the kernel's reward: for the goals a transition really had, it
    must return the reward the step returned -- every step of an episode, both reward forms.  One documented exception
    restates the reference: compute_reward takes `steps_count = info["step"]` (fixed_wing.py:1218-1277), one less than the
    env's counter inside step(), so at the SECOND transition its action-"delta" factor is still switched off
    (`if steps_count > 1`, :692) while the step's was on; the recorded vectors (g4_goal_*.json) hold exactly that value."""
    import json
    from gym_fixed_wing.fixed_wing import FixedWingAircraftGoal
    with open(os.path.join(os.path.dirname(__file__), "golden", "g4_goal_{}.json".format(form))) as f:
        rec = json.load(f)
    env = FixedWingAircraftGoal(rec["config"], **_kw(backend))
    env.seed(7)
    env.reset(state=rec["state"], target=rec["target"])
    prev = [env.simulator.state[s].value for s in env.goal_states]
    for t, st in enumerate(rec["steps"]):
        a = np.array(st["action"])
        o, r, d, _ = env.step(a)
        got = env.compute_reward(o["achieved_goal"], o["desired_goal"], {"step": t, "action": a, "prev_state": list(prev)})
        prev = [env.simulator.state[s].value for s in env.goal_states]
        if t == 1:
            assert abs(got - r) > 1e-3, "the reference's off-by-one of the delta factor at the second transition is gone"
        else:
            assert got == pytest.approx(r, abs=5e-5), (t, got, r)
    env.close()


@pytest.mark.parametrize("fail_prone", [False, True], ids=["time_limit", "failure"])
@pytest.mark.parametrize("backend", BACKENDS)
def test_integrator_observations_across_explicit_resets(backend, fail_prone):
    """integration_window through the single-env class (auto-reset off: fwg_reset between episodes): the integrator entries
    of a reset observation show the PREVIOUS episode's windowed error sum (fixed_wing.py:317-321 builds the observation before
    the histories are re-created), one record shorter after an episode that ended in a failed step."""
    cfg = configs.reference_like("integrator")
    ckw = {"steps_max": 30}
    if fail_prone:
        ckw["simulator"] = {"states": {6: {"constraint_min": -40, "constraint_max": 40}}}
    env = FixedWingAircraft(cfg, config_kw=ckw, **_kw(backend))
    orc = parity.make_oracles(cfg, 1, 0, config_kw=ckw)[0]
    env.seed(0)
    rng = np.random.default_rng(5)
    ends = []
    for ep in range(4):
        st = dict(STATE, roll=0.3 + 0.2 * ep)
        obs, want = env.reset(state=st, target=TARGET), orc.reset(state=st, target=TARGET)
        np.testing.assert_allclose(obs, want, atol=2e-3, rtol=2e-3, err_msg="reset observation of episode {}".format(ep))
        done = False
        while not done:
            a = rng.uniform(-1.8, 1.8, 3) if fail_prone else rng.uniform(-1, 1, 3)
            obs, rew, done, info = env.step(a)
            o2, r2, d2, i2 = orc.step(a)
            np.testing.assert_allclose(obs, o2, atol=2e-3, rtol=2e-3)
            assert rew == pytest.approx(r2, abs=2e-3) and done == d2
        ends.append(info["termination"])
    assert (set(ends) != {"steps"}) == fail_prone, ends
    env.close()
