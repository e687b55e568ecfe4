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
