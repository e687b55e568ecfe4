"""Differential fuzzer on the host emulation of the kernels (CPU suite, seeded): random draws over

    {observation structure, layout (dense batch / row log), turbulence mode, fail-prone constraints, steps_max 7-60, on_success,
     target resampling, reward form, aircraft randomisation, derived host views}
  x {masked resets at random steps, jumpy actions with bursts, the order in which the emulator schedules the waves of a workgroup}

and, for every drawn configuration, THREE kernel tiers of the same source -- the generic one-wave kernel (k_step<*, -1>, the
configuration interpreted), the frozen one-wave kernel (k_step<*, 0>, FWGYM_SPLIT=0) and the frozen two-wave kernel (k_step2: a
physics wave and a gym wave per 64 envs that meet through LDS messages) -- compared (1) each against the float64 oracle on
everything step()/reset() return incl. terminal observations and metrics, (2) pairwise on the outputs, and (3) pairwise on the
state arena word by word after the last step (the sections whose content is a matter of scheduling are left out, see ARENA_SKIP).

Rounds 1-5 compared tiers only by accident (tests/test_shape_instance.py, where round 5's two terminal-observation bugs
surfaced).  FWGYM_FUZZ_SEED / FWGYM_FUZZ_CONFIGS widen the search (every new configuration costs one g++ build of the emulation,
~30 s, cached under tests/emu/); the defaults are what the CPU suite runs."""
import copy
import os

import numpy as np
import pytest

import configs
import oracle_pool as op
import parity
from emu.host_backend import HostBackend, build_emu, build_emu_spec
from gym_fixed_wing import presets
from gym_fixed_wing.config import EnvConfig
from gym_fixed_wing.vec_env import FixedWingVecEnv

SEED = int(os.environ.get("FWGYM_FUZZ_SEED", "6"))
N_CONFIGS = int(os.environ.get("FWGYM_FUZZ_CONFIGS", "5"))
MUT_SRC, MUT_TAG = os.environ.get("FWGYM_MUTANT_SRC"), os.environ.get("FWGYM_MUTANT_TAG", "")

# arena sections not compared between tiers: the prepared draw of the NEXT episode (`draw`, `aero_next`, `fscale_next`,
# `model_raw_next`) is computed piecewise on steps the tiers choose differently (the two-wave kernel spreads it over the gym
# wave's waiting time); the finished-episode record (`fin`) of an env that ended is consumed by fwg_finish_episodes
ARENA_SKIP = ("draw", "aero_next", "fscale_next", "model_raw_next", "fin")
SECTIONS = ["sim", "cold", "derived", "gym", "tprop", "goal", "act_ring", "cmd_ring", "end_ring", "lag_ring", "draw", "aero",
            "aero_next", "fscale", "fscale_next", "model_raw", "model_raw_next", "fin"]


def draw_case(rng):
    kind = str(rng.choice(["default", "examples", "mlp", "cnn", "cnn"]))
    cfg = configs.reference_like(kind)
    ckw = {"steps_max": int(rng.integers(7, 61))}
    if kind == "cnn":
        ckw["observation"] = {"step": int(rng.choice([1, 2, 2]))}
    skw = None
    turb = str(rng.choice(["off", "increment", "filter", "increment"]))
    if turb != "off":
        skw = {"turbulence": True, "turbulence_intensity": str(rng.choice(["light", "moderate", "severe"]))}
        if turb == "filter":
            skw["turbulence_output"] = "filter"
    if rng.uniform() < 0.6:     # fail-prone: a tight roll-rate constraint (degrees per second)
        lim = float(rng.integers(40, 90))
        ckw["simulator"] = {"states": {6: {"constraint_min": -lim, "constraint_max": lim}}}
    on_success = str(rng.choice(["none", "none", "done", "new"]))
    tgt = {"on_success": on_success}
    if on_success != "none" or rng.uniform() < 0.3:
        tgt.update({"success_streak_req": int(rng.integers(5, 13)), "success_streak_fraction": float(rng.choice([0.5, 0.75, 0.9])),
                    "states": {0: {"bound": 90}, 1: {"bound": 40}, 2: {"bound": 10}}})
    if rng.uniform() < 0.35:
        tgt["resample_every"] = int(rng.integers(9, 30))
    ckw["target"] = tgt
    if rng.uniform() < 0.4:
        ckw["reward"] = {"form": "potential"}
    if rng.uniform() < 0.25:
        cfg["simulator"]["model"] = copy.deepcopy(configs.reference_like("model_gaussian")["simulator"]["model"])
    lean = bool(rng.uniform() < 0.5)
    want_log = bool(rng.uniform() < 0.6)
    order = str(rng.choice(["0", "1", "r{}".format(int(rng.integers(1, 1000)))]))
    return {"kind": kind, "cfg": cfg, "ckw": ckw, "skw": skw, "lean": lean, "want_log": want_log, "order": order}


def draw_case_v2(rng):
    """Seeds >= 100: a wider feature space (the draws of the seeds below 100 stay what they were, PINNED refers to them) --
    + linear / sinusoidal targets, the mixed reward (quadratic / exponential classes, success and goal factors), integrator
    observations + int_error reward, sampled simulator keys, randomised reward scalings, normalised observations."""
    kind = str(rng.choice(["default", "examples", "mlp", "cnn", "dynamic_targets", "reward_mix", "integrator", "sim_keys",
                           "reward_random_scaling", "model_uniform"]))
    cfg = configs.reference_like(kind)
    ckw = {"steps_max": int(rng.integers(7, 61))}
    if kind == "cnn":
        ckw["observation"] = {"step": int(rng.choice([1, 2, 3]))}
    elif kind == "integrator" and rng.uniform() < 0.5:
        ckw["observation"] = {"length": int(rng.choice([2, 3])), "shape": "matrix", "step": int(rng.choice([1, 2]))}
    elif rng.uniform() < 0.25:
        ckw["observation"] = {"normalize": True}
    skw = None
    turb = "increment" if kind == "sim_keys" else str(rng.choice(["off", "increment", "filter"]))
    if turb != "off":
        skw = {"turbulence": True, "turbulence_intensity": str(rng.choice(["light", "moderate", "severe"]))}
        if turb == "filter" or (kind == "sim_keys" and rng.uniform() < 0.5):
            skw["turbulence_output"] = "filter"
    if rng.uniform() < 0.6:
        lim = float(rng.integers(40, 90))
        ckw["simulator"] = {"states": {6: {"constraint_min": -lim, "constraint_max": lim}}}
    on_success = str(rng.choice(["none", "none", "done", "new"]))
    tgt = {"on_success": on_success}
    if kind != "reward_mix" and (on_success != "none" or rng.uniform() < 0.3):
        tgt.update({"success_streak_req": int(rng.integers(5, 13)), "success_streak_fraction": float(rng.choice([0.5, 0.75, 0.9])),
                    "states": {0: {"bound": 90}, 1: {"bound": 40}, 2: {"bound": 10}}})
    if rng.uniform() < 0.35:
        tgt["resample_every"] = int(rng.integers(9, 30))
    ckw["target"] = tgt
    if rng.uniform() < 0.4:
        ckw["reward"] = {"form": "potential"}
    lean = bool(rng.uniform() < 0.5)
    want_log = bool(rng.uniform() < 0.6)
    order = str(rng.choice(["0", "1", "r{}".format(int(rng.integers(1, 1000)))]))
    return {"kind": kind, "cfg": cfg, "ckw": ckw, "skw": skw, "lean": lean, "want_log": want_log, "order": order}


# draws of other seeds that FOUND something (kept in the default run): (seed, index) -- s7c4 / s8c1: the shipped cnn
# configuration's observation (5 rows at step 1) on the row log, a step that fails on the log's wrap step: the oldest row of the
# terminal observation was read one plane past the log (log_plane, csrc/fwgym_env.h), rounds 1-5
PINNED = [(7, 4), (8, 1)]


def _draws(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        c = draw_case(rng) if seed < 100 else draw_case_v2(rng)
        try:   # (a drawn combination the configuration compiler refuses is redrawn)
            EnvConfig(copy.deepcopy(c["cfg"]), config_kw=copy.deepcopy(c["ckw"]), sim_config_kw=copy.deepcopy(c["skw"])).compile()
        except (NotImplementedError, ValueError, KeyError):
            continue
        c["seed"], c["index"] = seed, len(out)
        c["id"] = "s{}c{}_{}_T{}_{}_{}".format(seed, len(out), c["kind"], c["ckw"]["steps_max"], "log" if c["want_log"] else "dense",
                                             "turb" if c["skw"] else "calm")
        out.append(c)
    return out


def _cases():
    out = _draws(SEED, N_CONFIGS)
    if "FWGYM_FUZZ_SEED" not in os.environ:
        for seed, index in PINNED:
            out.append(_draws(seed, index + 1)[index])
    return out


def _schedule(rng, n, steps):
    acts = np.zeros((steps, n, 3), dtype=np.float32)
    cur = rng.uniform(-1, 1, size=(n, 3))
    for t in range(steps):
        jump = rng.uniform(size=(n, 1)) < 0.3
        scale = 2.5 if t % 13 == 0 else 1.5
        cur = np.where(jump, np.clip(cur + rng.normal(0, 0.5, size=(n, 3)), -scale, scale), cur)
        acts[t] = cur
    resets = {}
    for t in sorted(rng.choice(np.arange(3, steps - 3), size=4, replace=False)):
        k = int(rng.integers(1, n // 2))
        resets[int(t)] = sorted(int(i) for i in rng.choice(n, size=k, replace=False))
    return acts, resets


def _arena_diff(a, b, skip):
    """Word-by-word comparison of two state arenas [word][env]: bit-equal, or two ordinary floats within 2e-4 (the tiers fold
    constants differently; integer words -- counters, bit rings, packed fixed point -- read as floats are denormal or differ
    in their high bits, and must be bit-equal)."""
    L = a.layout
    offs = sorted((int(getattr(L, k)), k) for k in SECTIONS if hasattr(L, k) and int(getattr(L, k)) >= 0)
    wa, wb = parity.words(a), parity.words(b)
    ia, ib = wa.view(np.uint32), wb.view(np.uint32)
    with np.errstate(invalid="ignore", over="ignore"):
        near = np.isfinite(wa) & np.isfinite(wb) & (np.abs(wa) > 1e-30) & (np.abs(wb) > 1e-30) & \
            (np.abs(wa - wb) <= 2e-4 * np.maximum(1.0, np.abs(wa)))
    bad = (ia != ib) & ~near
    out = []
    for w in np.nonzero(bad.any(axis=1))[0]:
        sec = "?"
        for o, k in offs:
            if o <= w:
                sec = k
        if sec in skip:
            continue
        e = int(np.nonzero(bad[w])[0][0])
        out.append("word {} ({}+{}), first env {}: {!r} / 0x{:08x} vs {!r} / 0x{:08x}".format(
            int(w), sec, int(w) - dict((k, o) for o, k in offs)[sec], e, float(wa[w, e]), int(ia[w, e]), float(wb[w, e]), int(ib[w, e])))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["id"])
def test_tiers_agree_with_each_other_and_with_the_oracle(case):
    cfg, ckw, skw = case["cfg"], case["ckw"], case["skw"]
    n, steps = 70, 140                                          # two workgroups, the second partially filled
    rng = np.random.default_rng(case["seed"] * 1000 + case["index"])
    acts, resets = _schedule(rng, n, steps)
    probe = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=1, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw),
                            _backend=HostBackend(), _lib_path=build_emu())
    rows = presets.OBS_LOG_ROWS if (case["want_log"] and probe._row_log_applies()) else 0
    probe.close()
    ec = EnvConfig(copy.deepcopy(cfg), config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw))
    spec_lib = build_emu_spec(ec, auto_reset=True, store_derived=not case["lean"], obs_log_rows=rows, src=MUT_SRC, tag=MUT_TAG)
    tiers = [("generic", build_emu(), None, "0"), ("frozen one-wave", spec_lib, "0", "0"), ("frozen two-wave", spec_lib, None, case["order"])]
    if MUT_SRC is not None:
        tiers = tiers[1:]
    recs, vecs = {}, {}
    try:
        for name, lib, split, order in tiers:
            os.environ["FWG_EMU_ORDER"] = order
            if split is not None:
                os.environ["FWGYM_SPLIT"] = split
            try:
                vec = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=n, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw),
                                      seed=21, as_numpy=True, derived_views=not case["lean"], obs_log_rows=rows,
                                      _backend=HostBackend(), _lib_path=lib)
            finally:
                os.environ.pop("FWGYM_SPLIT", None)
            assert (vec.spec_index == 0) == (name != "generic"), (name, vec.spec_index)
            recs[name] = op.record_run(vec, acts, resets=resets)
            vecs[name] = vec
    finally:
        os.environ.pop("FWG_EMU_ORDER", None)
    tr = op.run_traces(copy.deepcopy(cfg), list(range(n)), acts, 21, config_kw=ckw, sim_config_kw=skw, resets=resets)
    summary = {}
    for name, rec in recs.items():
        summary[name] = op.compare(rec, tr, 4e-3, 4e-3, what="{} [{}] vs oracle".format(case["id"], name))
    assert tr["done"].sum() >= n, "the drawn case ends too few episodes"
    names = list(recs)
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            a, b = recs[names[i]], recs[names[j]]
            tier_as_trace = dict(b, env_ids=list(range(n)))
            op.compare(a, tier_as_trace, 2e-4, 2e-4, metric_rtol=2e-3, what="{}: [{}] vs [{}]".format(case["id"], names[i], names[j]))
            diff = _arena_diff(vecs[names[i]], vecs[names[j]], ARENA_SKIP)
            assert not diff, "{}: arena of [{}] vs [{}] after {} steps:\n  ".format(case["id"], names[i], names[j], steps) + "\n  ".join(diff[:12])
    print(case["id"], "order", case["order"], "rows", rows, {k: (v["episodes"], v["terminations"]) for k, v in summary.items()})
    for v in vecs.values():
        v.close()
