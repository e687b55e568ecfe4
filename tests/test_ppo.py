"""The learner half of SURVEY 8(f2): fwg_gae against the backward loop of stable-baselines' PPO2 runner, the clipped-surrogate
loss against a hand computation, the training loop on the emulated env (CPU), and -- on the GPU -- a policy trained from random
initialisation to the success criterion of the reference's curriculum (examples/train_rl_controller.py:80-87, tensorboard.png)."""
import copy
import math

import numpy as np
import pytest
import torch

import configs
from gym_fixed_wing import _native as nat
from gym_fixed_wing.ppo import PPO, gae, ppo_loss, sb_init_
from gym_fixed_wing.rollout import MlpPolicy


def _gae_reference(rew, val, done, last_value, gamma, lam):
    """stable-baselines ppo2.py Runner.run, with mb_dones shifted to 'done returned by step t' (our buffers)."""
    T, N = rew.shape
    adv = np.zeros((T, N))
    lastgaelam = np.zeros(N)
    for t in reversed(range(T)):
        nonterminal = 1.0 - done[t]
        nextvalues = last_value if t == T - 1 else val[t + 1]
        delta = rew[t] + gamma * nextvalues * nonterminal - val[t]
        adv[t] = lastgaelam = delta + gamma * lam * nonterminal * lastgaelam
    return adv, adv + val


def _check_gae(lib, mem, T, N, to_np):
    rng = np.random.default_rng(T * 1000 + N)
    rew = rng.normal(0, 1, (T, N)).astype(np.float32)
    val = rng.normal(0, 2, (T, N)).astype(np.float32)
    done = (rng.uniform(size=(T, N)) < 0.07).astype(np.uint8)
    last = rng.normal(0, 2, N).astype(np.float32)
    adv, ret = gae(lib, mem, mem.from_host(rew), mem.from_host(val), mem.from_host(done, "u8"), mem.from_host(last), 0.99, 0.95)
    mem.sync()
    want_adv, want_ret = _gae_reference(rew.astype(np.float64), val.astype(np.float64), done.astype(np.float64), last.astype(np.float64), 0.99, 0.95)
    np.testing.assert_allclose(to_np(adv), want_adv, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(to_np(ret), want_ret, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("T,N", [(1, 1), (7, 70), (128, 300), (19, 513), (260, 100)])
def test_gae_kernel_matches_the_runner_loop_emulated(T, N):
    from emu.host_backend import HostBackend, build_emu
    _check_gae(nat.load_library(build_emu()), HostBackend(), T, N, np.asarray)


@pytest.mark.gpu
@pytest.mark.parametrize("T,N", [(128, 4096), (128, 65536), (5, 1000), (300, 777)])
def test_gae_kernel_matches_the_runner_loop_on_gpu(T, N):
    from gym_fixed_wing.vec_env import _TorchBackend
    _check_gae(nat.load_library(), _TorchBackend(0), T, N, lambda t: t.cpu().numpy())


def test_ppo_loss_is_ppo2s_objective():
    torch.manual_seed(0)
    pol = sb_init_(MlpPolicy(12))
    with torch.no_grad():
        pol.log_std.copy_(torch.tensor([-0.3, 0.1, -0.8]))
        pol.pi[-1].weight.mul_(30.0)
    n = 257
    obs, act = torch.randn(n, 12), torch.randn(n, 3) * 0.7
    old_v, old_lp, adv, ret = torch.randn(n), -3.0 + 0.3 * torch.randn(n), torch.randn(n) * 2 + 0.5, torch.randn(n)
    loss, st = ppo_loss(pol, obs, act, old_v, old_lp, adv, ret, 0.2, 0.01, 0.5)
    # hand computation in float64 numpy
    with torch.no_grad():
        mean, v = pol.pi(obs).double().numpy(), pol.vf(obs).squeeze(-1).double().numpy()
    ls = pol.log_std.detach().double().numpy()
    a, ov, olp, A, R = act.double().numpy(), old_v.double().numpy(), old_lp.double().numpy(), adv.double().numpy(), ret.double().numpy()
    A = (A - A.mean()) / (A.std() + 1e-8)
    neglogp = 0.5 * (((a - mean) / np.exp(ls)) ** 2).sum(1) + 0.5 * np.log(2 * np.pi) * 3 + ls.sum()
    ratio = np.exp(-olp - neglogp)
    pg = np.maximum(-A * ratio, -A * np.clip(ratio, 0.8, 1.2)).mean()
    vc = ov + np.clip(v - ov, -0.2, 0.2)
    vf = 0.5 * np.maximum((v - R) ** 2, (vc - R) ** 2).mean()
    ent = (ls + 0.5 * np.log(2 * np.pi * np.e)).sum()
    np.testing.assert_allclose(float(loss), pg - 0.01 * ent + 0.5 * vf, rtol=2e-5)
    assert 0.0 < float(st["clip_frac"]) < 1.0
    # the sampling log-probability of MlpPolicy.act (and of the HIP head) is the one the loss recomputes: ratio = 1 on fresh data
    torch.manual_seed(1)
    a2, v2, lp2 = pol.act(obs)
    _, st2 = ppo_loss(pol, obs, a2, v2, lp2, adv, ret, 0.2, 0.0, 0.5)
    assert float(st2["approx_kl"]) < 1e-10 and float(st2["clip_frac"]) == 0.0


def test_training_loop_on_the_emulated_env():
    """Two updates of the whole loop on the host emulation: rollout through the HIP head's emulation, fwg_gae, PPO updates, weights
    back into the head, success reduction and curriculum."""
    from emu.host_backend import HostBackend, build_emu
    from gym_fixed_wing.distributed import CurriculumSchedule
    from gym_fixed_wing.vec_env import FixedWingVecEnv
    cfg = configs.reference_like("examples")
    vec = FixedWingVecEnv(cfg, num_envs=70, config_kw={"steps_max": 25}, seed=3, _backend=HostBackend(), _lib_path=build_emu())
    vec.set_curriculum_level(0.25)
    vec.reset()
    ppo = PPO(vec, seed=0, n_steps=16, nminibatches=2, noptepochs=2, curriculum=CurriculumSchedule(level=0.25, cooldown=1))
    before = copy.deepcopy(ppo.policy.state_dict())
    logs = []
    ppo.learn(2 * 16 * 70, log=logs.append)
    assert ppo.updates == 2 and ppo.num_timesteps == 2 * 16 * 70
    assert any(not torch.equal(before[k], v) for k, v in ppo.policy.state_dict().items())
    assert all(math.isfinite(l["pg_loss"]) and math.isfinite(l["vf_loss"]) for l in logs)
    assert logs[-1]["episodes"] >= 0 and sum(l["episodes"] for l in logs) >= 70        # 25-step episodes end inside 32 steps
    # the head carries the UPDATED weights: its value of an observation equals the torch policy's
    st = ppo.actor.get_stats()
    o = np.asarray(vec._obs, dtype=np.float32).reshape(70, -1)
    _, _, value, _, _ = ppo.actor.act(o, deterministic=True)
    normed = np.clip((o - st["obs_mean"]) / np.sqrt(st["obs_var"] + 1e-8), -10, 10)
    with torch.no_grad():
        want = ppo.policy.vf(torch.as_tensor(normed)).squeeze(-1).numpy()
    np.testing.assert_allclose(np.asarray(value), want, rtol=2e-3, atol=2e-3)
    vec.close()


# ----------------------------------------------------------------------------------------------------------------------
# the end-to-end check: a policy trained from random initialisation (GPU)
# ----------------------------------------------------------------------------------------------------------------------
TRAIN_BUDGET = 80e6     # env steps (the reference: 5e6 steps of 4 envs -- 512 transitions per update against 524 288 here)


@pytest.mark.gpu
def test_a_policy_trains_to_the_curriculum_s_success_criterion_on_gpu():
    """examples/train_ppo.py's recipe on 4 096 envs: PPO2's objective and defaults but 128 minibatches per update and lr 5e-4.
    Gates: the reference's curriculum rule (level := min(2 x mean success, 1) whenever the mean success of the finished episodes
    exceeds the level; train_rl_controller.py:80-87) has reached level 1 AND the last full cohort of episodes at level 1
    succeeds in >= 50 % of the cases, within TRAIN_BUDGET env steps.  examples/tensorboard.png of the reference shows
    success_all ~0.8 after 4-5 M steps of 4 envs (~10 000 updates); measured here: level 1 after 58 M steps (110 updates),
    0.79 at 74 M, 0.90 at 148 M (profiles/r06_ppo_training.txt).
    Then the reference's evaluation protocol on the shipped no-wind scenario set with the trained policy (reported next to
    examples/README.md:37, gated loosely: the policy saw 80 M steps, not a tuned run)."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples"))
    import train_ppo
    lines = []
    ppo, res = train_ppo.train(envs=4096, timesteps=TRAIN_BUDGET, seed=0, log=lines.append)
    cohorts = [(n, s, lvl) for n, s, lvl in res["episodes_log"] if n >= 2048]
    print("\n".join(l for l in lines if "success_all  " not in l and " -  " not in l))
    print("env-steps/s incl. the optimiser: {:.3e}; cohorts (episodes, success_all, level): {}".format(res["env_steps_per_s"], cohorts))
    assert ppo.curriculum.level >= 1.0, ppo.curriculum.level
    at_top = [s for n, s, lvl in cohorts if lvl >= 1.0]
    assert at_top and at_top[-1] >= 0.5, cohorts
    # (ii) evaluate_on_set: the shipped 100 no-wind scenarios (tests/golden/test_set_wind_none.json), on_success = done, 100-step
    # streak at fraction 1, bounds 5 deg / 5 deg / 2 m/s, 1500 steps
    from gym_fixed_wing import evaluate as ev
    with open(os.path.join(root, "tests", "golden", "test_set_wind_none.json")) as f:
        scen = json.load(f)
    scen = scen["scenarios"] if isinstance(scen, dict) else scen
    out = ev.evaluate_on_set(scen, configs.reference_like("examples"), policy=ppo.deterministic_policy(), device=0)
    table = ev.summarize(out)
    report = {"trained_here_{:.0e}_steps".format(TRAIN_BUDGET): table,
              "published_README_RL_MLP_none": {"success_%": 100, "rise_time": [1.395, 0.336, 0.959], "settling_time": [2.085, 1.675, 2.308],
                                               "overshoot_%": [5, 25, 20], "control_variation": 0.410}}
    print(json.dumps(report, indent=1))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "ppo_eval_report.json"), "w") as f:
        json.dump({"report": report, "cohorts": cohorts, "env_steps_per_s_incl_optimiser": res["env_steps_per_s"]}, f, indent=1)
    assert table["success_%"]["all"] >= 50.0, table["success_%"]
    ppo.vec.close()
