"""Build-container-only check: the restated gym oracle equals the VERBATIM reference module step by step (float64),
for the shipped config variants and the feature switches, with the reference's own MT19937 consumption order.
Skipped where /root/reference is not mounted (GPU box) -- there the committed golden vectors pin the oracle."""
import copy

import numpy as np
import pytest

import configs
from golden._ref_harness import load_reference
from oracle.gym_restated import FixedWingOracle

ref = load_reference()
pytestmark = pytest.mark.skipif(ref is None, reason="reference not mounted")


def _cmp(a, b, path=""):
    if isinstance(a, dict):
        assert set(a.keys()) == set(b.keys()), (path, a.keys(), b.keys())
        return max([_cmp(a[k], b[k], path + "/" + str(k)) for k in a] + [0])
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (path, a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b)), (path, a, b)
    return float(np.max(np.abs(np.where(np.isnan(a), 0, a - b)))) if a.size else 0.0


CASES = [c for c in configs.CASES if c[0] not in ("spec_c3",)] + configs.ORACLE_ONLY_CASES


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_restated_gym_equals_verbatim_reference(case, tmp_path):
    import json
    name, kind, ckw, skw = case
    cfg = configs.reference_like(kind)
    path = str(tmp_path / "cfg.json")
    with open(path, "w") as f:
        json.dump(cfg, f)
    e1 = ref.FixedWingAircraft(path, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw))
    e2 = FixedWingOracle(cfg, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw))
    e1.seed(3)
    e2.seed(3)
    worst = _cmp(e1.reset(), e2.reset(), "reset")
    rng = np.random.default_rng(1)
    a = rng.uniform(-1, 1, 3)
    episodes = 0
    for t in range(260):
        if rng.uniform() < 0.3:
            a = np.clip(a + rng.normal(0, 0.4, 3), -1.8, 1.8)
        o1, r1, d1, i1 = e1.step(a.copy())
        o2, r2, d2, i2 = e2.step(a.copy())
        assert d1 == d2 and i1.get("termination") == i2.get("termination"), t
        worst = max(worst, _cmp(o1, o2, "obs"), _cmp(r1, r2, "reward"))
        worst = max(worst, _cmp({k: v for k, v in i1.items() if k != "termination"},
                                {k: v for k, v in i2.items() if k != "termination"}, "info"))
        if d1:
            episodes += 1
            worst = max(worst, _cmp(e1.reset(), e2.reset(), "reset"))
    assert worst < 1e-9, worst


# ---- the presets against the configuration FILES the reference ships ------------------------------------------------------
# keys a preset may carry / lack without changing what the environment computes, each with the reason it is inert
INERT = {
    "mlp": {"+": {"action.scale_high": "read only when action.scale_space is true (fixed_wing.py:349-354); the mlp file sets it false",
                  "action.scale_low": "as scale_high"}, "-": {}},
    "cnn": {"+": {}, "-": {"action.bounds_outside_cost": "never read by fixed_wing.py (grep: no occurrence)"}},
}
SHIPPED = {"default": "fixed_wing_config.json", "dev": "fixed_wing_config_dev.json", "examples": "examples/fixed_wing_config.json",
           "mlp": "examples/models/mlp_controller/fixed_wing_config.json", "cnn": "examples/models/cnn_controller/fixed_wing_config.json"}


def _flat(d, prefix=""):
    out = {}
    if isinstance(d, dict):
        for k, v in d.items():
            out.update(_flat(v, prefix + ("." if prefix else "") + str(k)))
    elif isinstance(d, list):
        for i, v in enumerate(d):
            out.update(_flat(v, prefix + "[{}]".format(i)))
    else:
        out[prefix] = d
    return out


@pytest.mark.parametrize("kind", sorted(SHIPPED))
def test_preset_equals_the_shipped_configuration_file(kind):
    """presets.preset(kind) -- what the kernels are frozen for and what every test calls 'the reference's configuration' -- is
    the file the reference ships, key by key and value by value, modulo the stated inert keys."""
    import json
    import os
    from gym_fixed_wing import presets
    with open(os.path.join("/root/reference/gym_fixed_wing", SHIPPED[kind])) as f:
        shipped = _flat(json.load(f))
    ours = _flat({k: v for k, v in presets.preset(kind).items() if k != "_comment"})
    extra = {k: v for k, v in ours.items() if k not in shipped}
    missing = {k: v for k, v in shipped.items() if k not in ours}
    differ = {k: (ours[k], shipped[k]) for k in ours if k in shipped and ours[k] != shipped[k]}
    allowed = INERT.get(kind, {"+": {}, "-": {}})
    assert set(extra) == set(allowed["+"]), ("keys only the preset has", extra)
    assert set(missing) == set(allowed["-"]), ("keys only the shipped file has", missing)
    assert not differ, differ
    # the inert keys really are: the reference module never reads `bounds_outside_cost`
    if kind == "cnn":
        with open("/root/reference/gym_fixed_wing/fixed_wing.py") as f:
            assert "bounds_outside_cost" not in f.read()
