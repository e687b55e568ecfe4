"""Parity tests proper: the HIP kernels on a real MI355X, called through the C ABI (libfwgym.so), against the float64
oracle on identical seeds/inputs.

Tolerances
  * single env step from identical state (the north-star bar): |gpu - oracle| <= 1e-5 * max(|oracle|, scale) on the 13
    rigid-body states (+ actuator states), where `scale` is the variable's natural magnitude (parity.STATE_SCALE);
  * free-running rollouts (fp32 vs fp64 drift accumulates over >100 steps): 4e-3 abs+rel on obs/reward/target,
    integer metrics within one step.
"""
import numpy as np
import pytest

import configs
import parity
from oracle import physics as ph

pytestmark = pytest.mark.gpu


def _vec(cfg, n, **kw):
    from gym_fixed_wing.vec_env import FixedWingVecEnv
    return FixedWingVecEnv(cfg, num_envs=n, device=0, **kw)


def _actions(seed, steps, n, scale=1.3):
    rng = np.random.default_rng(seed)
    a = np.zeros((steps, n, 3), dtype=np.float32)
    cur = rng.uniform(-1, 1, size=(n, 3))
    for t in range(steps):
        jump = rng.uniform(size=(n, 1)) < 0.3
        cur = np.where(jump, np.clip(cur + rng.normal(0, 0.4, size=(n, 3)), -scale, scale), cur)
        a[t] = cur
    return a


def test_native_library_is_loaded():
    import ctypes
    from gym_fixed_wing import _native as nat
    lib = nat.load_library()
    assert lib.fwg_abi_version() == nat.FWG_ABI_VERSION
    with open("/proc/self/maps") as f:
        assert "libfwgym.so" in f.read()


@pytest.mark.parametrize("case", configs.CASES, ids=[c[0] for c in configs.CASES])
def test_gym_rollout_matches_oracle(case):
    name, kind, ckw, skw = case
    cfg = configs.reference_like(kind)
    n, steps = 6, 130
    vec = _vec(cfg, n, config_kw=ckw, sim_config_kw=skw, seed=11, as_numpy=True)
    assert (vec.spec_index >= 0) == (name in configs.SPECIALISED_CASES), (name, vec.spec_index)
    orc = parity.make_oracles(cfg, n, 11, config_kw=ckw, sim_config_kw=skw)
    acts = _actions(5, steps, n, scale=1.8 if name == "fail_prone" else 1.3)
    tol = 5e-2 if name == "dev_noise" else 4e-3
    res = parity.run_gym_parity(vec, orc, steps, lambda t: acts[t], rtol=tol, atol=tol)
    print(name, res)
    vec.close()


@pytest.mark.parametrize("case", [c for c in configs.CASES if c[1] == "cnn"], ids=[c[0] + "_dense" for c in configs.CASES if c[1] == "cnn"])
def test_gym_rollout_matches_oracle_dense_batch(case):
    """Lagged matrix observations default to the row log (zero-copy window); the dense [N][5][12] batch written by the
    step kernel is the other layout of the same values and must hold the same parity."""
    name, kind, ckw, skw = case
    cfg = configs.reference_like(kind)
    n, steps = 6, 130
    vec = _vec(cfg, n, config_kw=ckw, sim_config_kw=skw, seed=11, as_numpy=True, obs_log_rows=0)
    assert vec.obs_log_rows == 0 and (vec.spec_index >= 0) == (name in configs.SPECIALISED_CASES)
    orc = parity.make_oracles(cfg, n, 11, config_kw=ckw, sim_config_kw=skw)
    acts = _actions(5, steps, n)
    res = parity.run_gym_parity(vec, orc, steps, lambda t: acts[t], rtol=4e-3, atol=4e-3)
    print(name, "dense", res)
    vec.close()


@pytest.mark.parametrize("turb", [False, True], ids=["calm", "dryden"])
@pytest.mark.parametrize("n", [64, 4096, 65536])
def test_single_step_state_parity_1e5(n, turb):
    """One env step from identical (device) state, all sizes up to BASELINE's 65 536 envs: 1e-5 relative on the state."""
    cfg = configs.reference_like("cnn" if turb else "default")
    skw = {"turbulence": True, "turbulence_intensity": "severe"} if turb else None
    vec = _vec(cfg, n, sim_config_kw=skw, seed=5, as_numpy=True, auto_reset=False)
    spec = parity.oracle_spec_from_env_config(vec.env_config)
    vec.reset()
    rng = np.random.default_rng(n)
    worst = 0.0
    for t in range(6):
        raw = rng.uniform(-1.5, 1.5, size=(n, 3)).astype(np.float32)
        y0, wind, dry = parity.physics_state(vec)
        gust = ph.dryden_output(spec, dry) if turb else np.zeros((n, 6))
        cmd = parity.scaled_actions(vec, raw)
        want, ok, fail, cmd_c, d = ph.sim_step(spec, y0, cmd, wind, gust)
        _, _, done, _ = vec.step(raw)
        y1, _, _ = parity.physics_state(vec)
        assert np.array_equal(np.asarray(done).astype(bool), ~ok)
        err = np.abs(y1 - want) / np.maximum(np.abs(want), parity.STATE_SCALE)
        worst = max(worst, float(err.max()))
        assert err.max() <= 1e-5, (t, np.unravel_index(np.argmax(err), err.shape), err.max())
        S = parity.words(vec)
        L = vec.layout
        for k, nm in enumerate(["roll", "pitch", "yaw", "Va", "alpha", "beta"]):
            e = np.abs(S[L.derived + k] - d[nm])
            e = np.minimum(e, np.abs(e - 2 * np.pi)) if nm in ("roll", "yaw") else e
            assert e.max() <= 2e-5 * max(1.0, float(np.abs(d[nm]).max())), (nm, e.max())
    print("n={} turb={} worst relative state error {:.3e}".format(n, turb, worst))
    vec.close()


def test_full_size_properties():
    """Size-independent properties at BASELINE's full size (65 536 envs, Dryden on, 5x12 observation with step 2):
    determinism (same seeds twice => bitwise-equal outputs), unit quaternions, counters, lag-row structure."""
    cfg = configs.reference_like("cnn")
    kw = dict(config_kw={"observation": {"step": 2}}, sim_config_kw={"turbulence": True, "turbulence_intensity": "moderate"},
              seed=9, as_numpy=True)
    n, steps = 65536, 12
    acts = np.random.default_rng(1).uniform(-1, 1, size=(steps, n, 3)).astype(np.float32)
    outs = []
    for rep in range(2):
        vec = _vec(cfg, n, **kw)
        vec.reset()
        hist = []
        for t in range(steps):
            obs, rew, done, _ = vec.step(acts[t])
            hist.append((np.array(obs), np.array(rew), np.array(done)))
        S = parity.words(vec)
        outs.append((hist, S.copy()))
        L = vec.layout
        q = S[L.sim:L.sim + 4]
        assert np.abs(np.sum(q * q, axis=0) - 1).max() < 1e-5
        assert np.all(np.isfinite(S[:L.gym + 3]))  # float words: simulator block, derived values, targets
        vec.close()
    for (o1, r1, d1), (o2, r2, d2) in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2)
    assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32))
    # lag structure: at t >= 9 row k of the matrix equals row 0 of 2k steps earlier
    hist = outs[0][0]
    alive = ~np.any(np.stack([h[2] for h in hist]), axis=0)
    for k in range(1, 5):
        np.testing.assert_array_equal(hist[11][0][alive, k, :], hist[11 - 2 * k][0][alive, 0, :])


def test_success_reduction_matches_infos():
    cfg = configs.reference_like("default")
    ckw = {"steps_max": 30, "target": {"success_streak_req": 5, "success_streak_fraction": 0.6,
                                       "states": {0: {"bound": 100}, 1: {"bound": 45}, 2: {"bound": 12}}}}
    n = 1000
    vec = _vec(cfg, n, config_kw=ckw, seed=4, as_numpy=True)
    vec.reset()
    rng = np.random.default_rng(0)
    eps, succ = 0, np.zeros(4)
    first = None
    for t in range(65):
        _, _, done, infos = vec.step(rng.uniform(-1, 1, size=(n, 3)).astype(np.float32))
        for i in np.nonzero(np.asarray(done))[0]:
            info = infos[int(i)]
            eps += 1
            succ += [info["success"][k] for k in ("roll", "pitch", "Va", "all")]
        if t == 40:   # the device-resident form (no host sync) takes and clears the sums the same way
            first = vec.reduce_success_device().cpu().numpy().astype(np.float64)
    red = vec.reduce_success() + first
    assert first[0] == n
    assert red[0] == eps == 2 * n
    np.testing.assert_array_equal(red[1:5], succ)
    assert np.all(vec.reduce_success() == 0)
    vec.close()


def test_runtime_specialisation_matches_generic_kernel():
    """A configuration outside the build-time presets: the generic kernel and a run-time specialised build of the same
    source (gym_fixed_wing/jit.py) give the same trajectories (fp32 contraction order may differ: 1e-5), and the
    specialised one is the one that runs."""
    import time
    import torch
    cfg = configs.reference_like("examples")
    ckw = {"steps_max": 1500, "target": {"on_success": "done", "success_streak_fraction": 1, "success_streak_req": 100}}
    n, steps = 4096, 60
    acts = np.random.default_rng(3).uniform(-1, 1, size=(steps, n, 3)).astype(np.float32)
    outs, rates = [], []
    for spec in (False, True):
        vec = _vec(cfg, n, config_kw=ckw, seed=2, as_numpy=True, specialize=spec)
        assert (vec.spec_index >= 0) == spec
        vec.reset()
        hist = []
        for t in range(steps):
            obs, rew, done, _ = vec.step(acts[t])
            hist.append((np.array(obs), np.array(rew), np.array(done)))
        outs.append(hist)
        dev_acts = torch.as_tensor(acts[0]).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(200):
            vec.step_device(dev_acts)
        torch.cuda.synchronize()
        rates.append((time.perf_counter() - t0) / 200 * 1e6)
        vec.close()
    for (o1, r1, d1), (o2, r2, d2) in zip(*outs):
        np.testing.assert_allclose(o1, o2, rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(r1, r2, rtol=2e-4, atol=2e-4)
        assert np.array_equal(d1, d2)
    print("generic {:.1f} us/step, run-time specialised {:.1f} us/step at {} envs".format(rates[0], rates[1], n))
