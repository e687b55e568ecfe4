"""Rollout head (fwg_actor_*: VecNormalize statistics + MlpPolicy on the matrix cores + sampling) against the plain
PyTorch fp32 formulation of the same operations (gym_fixed_wing.rollout.VecNormalizeDevice / MlpPolicy).  CPU: the
host-emulation build of the kernels (MFMA emulated lane by lane); GPU: libfwgym.so."""
import math

import numpy as np
import pytest
import torch

from gym_fixed_wing import _native as nat
from gym_fixed_wing.actor import DeviceActor, weights_from_module, weights_from_stable_baselines
from gym_fixed_wing.rollout import MlpPolicy, VecNormalizeDevice


def _emu():
    from emu.host_backend import HostBackend, build_emu
    return dict(_backend=HostBackend(), _lib=nat.load_library(build_emu()))


def _gpu():
    return dict(device=0)


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


class _Rms64(object):
    """RunningMeanStd of stable-baselines' VecNormalize in float64 (the yardstick for the fp32 statistics)."""

    def __init__(self, shape):
        self.mean, self.var, self.count = np.zeros(shape), np.ones(shape), 1e-4

    def update(self, x):
        x = np.asarray(x, np.float64)
        bm, bv, bc = x.mean(axis=0), x.var(axis=0), x.shape[0]
        delta, tot = bm - self.mean, self.count + bc
        m2 = self.var * self.count + bv * bc + delta ** 2 * self.count * bc / tot
        self.mean, self.var, self.count = self.mean + delta * bc / tot, m2 / tot, tot


def run_actor_parity(mk, n, d, steps=3, precise=True, tol=2e-5):
    torch.manual_seed(d)
    rng = np.random.default_rng(n + d)
    policy = MlpPolicy(d)
    with torch.no_grad():
        policy.log_std.copy_(torch.tensor([-0.3, 0.1, 0.4]))
        for p_ in policy.parameters():
            if p_.dim() == 2:
                p_.mul_(2.5)   # larger pre-activations: exercises tanh away from the linear range
    norm = VecNormalizeDevice((d,), n)
    o64, r64, ret64 = _Rms64((d,)), _Rms64(()), np.zeros(n)
    kw = mk()
    actor = DeviceActor(n, d, precise=precise, seed=11, env_id_base=5, **kw)
    actor.load_policy(policy)
    mem = actor._mem
    scale = rng.uniform(0.5, 4.0, d).astype(np.float32)
    shift = rng.uniform(-5, 5, d).astype(np.float32)
    worst = {"obs": 0.0, "mean": 0.0, "value": 0.0, "rew": 0.0}
    rew = done = None
    for t in range(steps):
        obs = (rng.normal(size=(n, d)).astype(np.float32) * scale + shift)
        obs_d = mem.from_host(obs)
        if t > 0:
            rew = rng.normal(size=n).astype(np.float32) * 3 - 1
            done = (rng.uniform(size=n) < 0.2).astype(np.uint8)
        actor.observe(obs_d, None if rew is None else mem.from_host(rew), None if done is None else mem.from_host(done, "u8"))
        no, act, val, logp, nrew = actor.act(obs_d, reward=None if rew is None else mem.from_host(rew), deterministic=True)
        w_no = norm.obs(torch.from_numpy(obs))   # torch fp32: keeps the statistics comparison below honest
        o64.update(obs)
        w_no = np.clip((obs.astype(np.float64) - o64.mean) / np.sqrt(o64.var + 1e-8), -10, 10)
        with torch.no_grad():   # the networks on OUR normalised observation: isolates the matrix-core arithmetic
            mine = torch.from_numpy(_np(no).copy())
            w_mean, w_val = policy.pi(mine), policy.vf(mine).squeeze(-1)
        worst["obs"] = max(worst["obs"], float(np.abs(_np(no) - _np(w_no)).max()))
        # relative to the largest output of the batch (the outputs are sums of 64 products of magnitude ~1)
        worst["mean"] = max(worst["mean"], float(np.abs(_np(act) - _np(w_mean)).max() / np.abs(_np(w_mean)).max()))
        worst["value"] = max(worst["value"], float(np.abs(_np(val) - _np(w_val)).max() / np.abs(_np(w_val)).max()))
        if rew is not None:
            norm.reward(torch.from_numpy(rew), torch.from_numpy(done))
            ret64 = ret64 * 0.99 + rew
            r64.update(ret64)
            w_r = np.clip(rew / np.sqrt(r64.var + 1e-8), -10, 10)
            ret64[done != 0] = 0.0
            worst["rew"] = max(worst["rew"], float(np.abs(_np(nrew) - _np(w_r)).max()))
        np.testing.assert_allclose(_np(logp), np.full(n, float((-policy.log_std.detach() - 0.5 * math.log(2 * math.pi)).sum())), rtol=1e-5)
    st = actor.get_stats()
    np.testing.assert_allclose(st["obs_mean"], o64.mean, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(st["obs_var"], o64.var, rtol=5e-5)
    np.testing.assert_allclose(st["ret_var"], r64.var, rtol=5e-5)
    np.testing.assert_allclose(st["obs_count"], o64.count, rtol=1e-6)
    np.testing.assert_allclose(_np(norm.obs_rms.var), o64.var, rtol=2e-4)   # the torch formulation agrees too
    assert worst["obs"] < 5e-5 and worst["rew"] < 5e-5, worst   # normalised values up to 10 in magnitude
    assert worst["mean"] < tol and worst["value"] < tol, worst
    actor.close()
    return worst


def run_sampling_checks(mk, n=300, d=12):
    torch.manual_seed(1)
    policy = MlpPolicy(d)
    with torch.no_grad():
        policy.log_std.copy_(torch.tensor([-0.5, 0.0, 0.3]))
    kw = mk()
    actor = DeviceActor(n, d, seed=3, training=False, **kw)
    actor.load_policy(weights_from_module(policy))
    actor.set_stats(np.linspace(-1, 1, d), np.linspace(0.5, 2, d), 1000.0)
    mem = actor._mem
    obs = mem.from_host(np.random.default_rng(0).normal(size=(n, d)).astype(np.float32))
    _, mean, _, _, _ = actor.act(obs, deterministic=True)
    mean = _np(mean).copy()
    _, a1, v1, lp1, _ = actor.act(obs)
    a1, lp1 = _np(a1).copy(), _np(lp1).copy()
    _, a2, _, _, _ = actor.act(obs)
    a2 = _np(a2).copy()
    ls = _np(policy.log_std)
    z = (a1 - mean) / np.exp(ls)
    want = (-0.5 * z * z - ls - 0.5 * math.log(2 * math.pi)).sum(axis=1)
    np.testing.assert_allclose(lp1, want, rtol=2e-4, atol=2e-4)
    assert np.abs(a1 - a2).max() > 0.1            # the act counter advances the noise stream
    zz = np.concatenate([z.ravel(), ((a2 - mean) / np.exp(ls)).ravel()])
    assert abs(zz.mean()) < 0.1 and abs(zz.std() - 1.0) < 0.1
    # same seed, same counter -> same noise; statistics frozen -> same normalised observation
    other = DeviceActor(n, d, seed=3, training=False, **mk())
    other.load_policy(policy)
    other.set_stats(np.linspace(-1, 1, d), np.linspace(0.5, 2, d), 1000.0)
    other.act(obs, deterministic=True)
    _, b1, _, _, _ = other.act(obs)
    np.testing.assert_array_equal(_np(b1), a1)
    st = actor.get_stats()
    np.testing.assert_allclose(st["obs_mean"], np.linspace(-1, 1, d), rtol=1e-6)   # evaluation mode: frozen
    actor.close(), other.close()


@pytest.mark.parametrize("n,d", [(200, 12), (70, 14), (130, 60)])
def test_actor_matches_torch_on_the_emulated_kernels(n, d):
    run_actor_parity(_emu, n, d)


def test_actor_plain_bf16_mode_is_close():
    w = run_actor_parity(_emu, 96, 12, precise=False, tol=6e-2)
    assert w["mean"] > 1e-5   # really the single-product path


def test_actor_sampling_on_the_emulated_kernels():
    run_sampling_checks(_emu)


def test_stable_baselines_layout_conversion():
    rng = np.random.default_rng(0)
    sb = {"pi_fc0_w": rng.normal(size=(12, 64)), "pi_fc0_b": rng.normal(size=64), "pi_fc1_w": rng.normal(size=(64, 64)),
          "pi_fc1_b": rng.normal(size=64), "pi_w": rng.normal(size=(64, 3)), "pi_b": rng.normal(size=3),
          "vf_fc0_w": rng.normal(size=(12, 64)), "vf_fc0_b": rng.normal(size=64), "vf_fc1_w": rng.normal(size=(64, 64)),
          "vf_fc1_b": rng.normal(size=64), "vf_w": rng.normal(size=(64, 1)), "vf_b": rng.normal(size=1)}
    w = weights_from_stable_baselines(sb)
    assert w["pi_w0"].shape == (64, 12) and w["pi_w2"].shape == (3, 64) and w["vf_w2"].shape == (1, 64)
    x = rng.normal(size=12).astype(np.float32)
    np.testing.assert_allclose(w["pi_w0"] @ x, x @ sb["pi_fc0_w"].astype(np.float32), rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("n,d", [(65536, 12), (1000, 14), (4096, 60)])
def test_actor_matches_torch_on_gpu(n, d):
    w = run_actor_parity(_gpu, n, d)
    print("actor parity n={} d={}: {}".format(n, d, w))


@pytest.mark.gpu
def test_actor_sampling_on_gpu():
    run_sampling_checks(_gpu, n=4096)
