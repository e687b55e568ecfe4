"""Shape instances (include/fwgym.h, fwg_spec_index / fwg_config_instance): a configuration that differs from a frozen one in
VALUES only -- reward scalings, constraints, the time limit, turbulence intensity, aircraft constants ... -- runs the frozen
configuration's kernel with the values read from memory.  Which tier a configuration lands on is a host-side decision (CPU
tests); that the shape instance computes what the generic kernel and the frozen kernel compute is checked on the GPU."""
import copy
import ctypes
import os

import numpy as np
import pytest

import configs
from gym_fixed_wing import _native as nat, presets
from gym_fixed_wing.config import EnvConfig
from gym_fixed_wing.vec_env import FixedWingVecEnv

VALUE_TWEAKS = {"steps_max": 90, "reward": {"factors": {0: {"scaling": 0.37}, 2: {"scaling": 11.0}}},
                "simulator": {"states": {6: {"constraint_min": -60, "constraint_max": 60}}}}


def _instance(lib, cfg, ckw, skw, rows, lean=True):
    ec = EnvConfig(copy.deepcopy(cfg), config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw))
    c = ec.compile(auto_reset=True, store_derived=not lean, obs_log_rows=rows)
    return int(lib.fwg_config_instance(ctypes.byref(c)))


def _merged(a, b):
    out = copy.deepcopy(a or {})
    for k, v in b.items():
        out[k] = _merged(out.get(k), v) if isinstance(v, dict) and isinstance(out.get(k), dict) else copy.deepcopy(v)
    return out


def test_config_lands_on_the_right_tier():
    lib = nat.load_library()
    n = lib.fwg_num_specs()
    cfg, ckw, skw, _, _ = presets.workload("c3")
    rows = presets.OBS_LOG_ROWS
    frozen = _instance(lib, cfg, ckw, skw, rows)
    assert 0 <= frozen < n                                                 # the preset itself: its frozen kernel
    for tweak in ({"steps_max": 1999}, VALUE_TWEAKS, {"reward": {"factors": {1: {"scaling": 5.0}}}}):
        assert _instance(lib, cfg, _merged(ckw, tweak), skw, rows) == nat.INSTANCE_SHAPE + frozen, tweak   # values only: its shape instance
    assert _instance(lib, cfg, ckw, _merged(skw, {"turbulence_intensity": "light"}), rows) == nat.INSTANCE_SHAPE + frozen
    assert _instance(lib, cfg, _merged(ckw, {"observation": {"step": 3}}), skw, rows) == nat.INSTANCE_GENERIC   # another structure
    assert _instance(lib, cfg, _merged(ckw, {"observation": {"length": 4}}), skw, rows) == nat.INSTANCE_GENERIC
    # dense batch / derived host views are other frozen configurations with shape instances of their own
    dense = _instance(lib, cfg, ckw, skw, 0)
    assert 0 <= dense < n and dense != frozen
    assert _instance(lib, cfg, _merged(ckw, VALUE_TWEAKS), skw, 0) == nat.INSTANCE_SHAPE + dense
    os.environ["FWGYM_SHAPE"] = "0"
    try:
        assert _instance(lib, cfg, _merged(ckw, {"steps_max": 1999}), skw, rows) == nat.INSTANCE_GENERIC
    finally:
        del os.environ["FWGYM_SHAPE"]


def test_emulation_build_has_no_shape_instances():
    from emu.host_backend import build_emu
    lib = nat.load_library(build_emu())
    cfg, ckw, skw, _, _ = presets.workload("c3")
    assert _instance(lib, cfg, _merged(ckw, {"steps_max": 1999}), skw, presets.OBS_LOG_ROWS) == nat.INSTANCE_GENERIC


def _run_pair(make_a, make_b, n, steps, scale=1.3):
    import torch
    a, b = make_a(), make_b()
    oa, ob = a.reset(), b.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    worst, ends, fails = float((oa - ob).abs().max()), 0, 0
    for t in range(steps):
        act = (torch.rand((n, 3), device="cuda", generator=gen) * 2 - 1) * (2.5 if t % 13 == 0 else scale)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(da, db), "step {}".format(t)
        assert torch.equal(a._term, b._term), "step {}".format(t)
        ends += int(da.sum())
        worst = max(worst, float((oa - ob).abs().max()), float((ra - rb).abs().max()))
        if bool(da.any()):   # the terminal observations of the envs that ended (two-wave kernel, row log: the lagged rows are
            m = da.bool()    # copied by the physics wave, a few ending lanes per wave sharing the words -- partner_rows)
            worst = max(worst, float((a._term_obs[m] - b._term_obs[m]).abs().max()))
    ia, ib = a.spec_index, b.spec_index
    a.close(), b.close()
    return worst, ends, ia, ib


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["row_log", "dense"])
def test_shape_instance_equals_generic_kernel_on_gpu(layout):
    """Same configuration (the C3 workload with other values), same seed, same actions: the shape instance against the generic
    kernel, through time-limit and failure episode ends.  Both read the values from memory; one has the structure folded."""
    cfg, ckw, skw, _, _ = presets.workload("c3")
    ckw = _merged(ckw, VALUE_TWEAKS)
    skw = _merged(skw, {"turbulence_intensity": "light"})
    n = 1100

    def make(shape):
        def f():
            if not shape:
                os.environ["FWGYM_SHAPE"] = "0"
            try:
                return FixedWingVecEnv(copy.deepcopy(cfg), num_envs=n, device=0, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw),
                                       derived_views=False, seed=5, obs_layout=layout, specialize=False)
            finally:
                os.environ.pop("FWGYM_SHAPE", None)
        return f
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        worst, ends, ia, ib = _run_pair(make(True), make(False), n, 200)
    assert ia >= nat.INSTANCE_SHAPE and ib == -1, (ia, ib)
    assert ends >= 2 * n
    assert worst <= 2e-4, worst   # (different instruction selection around folded branches; chaotic amplification over 90-step episodes)


@pytest.mark.gpu
def test_shape_instance_equals_frozen_kernel_on_gpu():
    """The preset itself, once on its frozen kernel and once forced onto its shape instance (FWGYM_SHAPE=force): immediates
    against scalar loads of the same numbers."""
    cfg, ckw, skw, _, _ = presets.workload("c3")
    n = 4096

    def make(force):
        def f():
            if force:
                os.environ["FWGYM_SHAPE"] = "force"
            try:
                return FixedWingVecEnv(copy.deepcopy(cfg), num_envs=n, device=0, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw),
                                       derived_views=False, seed=5, obs_log_rows=presets.OBS_LOG_ROWS)
            finally:
                os.environ.pop("FWGYM_SHAPE", None)
        return f
    worst, ends, ia, ib = _run_pair(make(False), make(True), n, 150)
    assert 0 <= ia < nat.INSTANCE_SHAPE and ib == nat.INSTANCE_SHAPE + ia, (ia, ib)
    assert worst <= 2e-4, worst


@pytest.mark.gpu
def test_shape_instance_matches_oracle_on_gpu():
    """Oracle parity THROUGH a shape instance: the cnn test configuration with short episodes and tight rate constraints
    differs from the shipped cnn preset in values only."""
    import parity
    cfg = configs.reference_like("cnn")
    ckw = {"observation": {"step": 2}, "steps_max": 45, "simulator": {"states": {6: {"constraint_min": -60, "constraint_max": 60}}}}
    skw = {"turbulence": True, "turbulence_intensity": "moderate"}
    n, steps = 70, 130
    vec = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=n, device=0, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw), seed=11,
                          as_numpy=True, specialize=False)
    assert vec.spec_index >= nat.INSTANCE_SHAPE, vec.spec_index
    orc = parity.make_oracles(cfg, n, 11, config_kw=ckw, sim_config_kw=skw)
    rng = np.random.default_rng(5)
    acts = rng.uniform(-1.3, 1.3, (steps, n, 3)).astype(np.float32)
    res = parity.run_gym_parity(vec, orc, steps, lambda t: acts[t], rtol=4e-3, atol=4e-3)
    assert res["episodes"] >= n
    vec.close()
