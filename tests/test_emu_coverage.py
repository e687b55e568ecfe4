"""CPU-suite versions of the round-6 oracle-coverage tests (the GPU ones are tests/test_gpu_oracle_coverage.py): the unchanged
kernel source, compiled for the host (tests/emu), in the regimes the oracle suite of rounds 1-5 never entered --

  * a FROZEN preset kernel (the benched k_step2<true, 6>: c3_cnn_step2_dryden_lean_log) through its real 2 000-step time limit;
  * the staggered steady state (bench.py's stagger_ages) of a batch in which every launch mixes ending, failing, early-episode
    and drawing lanes, with the checked env ids chosen AFTER the run from what happened (coverage_runs.steady_state_sampled).

tools/mutation_check.py re-runs this file against kernel sources with the two terminal-observation bugs of rounds 1-5 put back
(FWGYM_MUTANT_SRC / FWGYM_MUTANT_TAG): both must fail here."""
import copy
import os

import numpy as np
import pytest

import configs
import coverage_runs as cr
from emu.host_backend import HostBackend, build_emu, build_emu_spec
from gym_fixed_wing import presets
from gym_fixed_wing.config import EnvConfig
from gym_fixed_wing.vec_env import FixedWingVecEnv

MUT_SRC, MUT_TAG = os.environ.get("FWGYM_MUTANT_SRC"), os.environ.get("FWGYM_MUTANT_TAG", "")

# short episodes, tight roll-rate constraint, Dryden turbulence, lagged 5 x 12 observation at step 2: failure ends, time-limit
# ends and steps that fail ON the time-limit step all occur within a few hundred steps of a few hundred envs
FAIL_PRONE = ("cnn", {"observation": {"step": 2}, "steps_max": 45,
                      "simulator": {"states": {6: {"constraint_min": -60, "constraint_max": 60}}}},
              {"turbulence": True, "turbulence_intensity": "moderate"})


def _spec_lib(cfg, ckw, skw, rows, lean=True):
    ec = EnvConfig(copy.deepcopy(cfg), config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw))
    return build_emu_spec(ec, auto_reset=True, store_derived=not lean, obs_log_rows=rows, src=MUT_SRC, tag=MUT_TAG)


@pytest.mark.parametrize("regime,layout", [("staggered", "row_log"), ("staggered", "dense"), ("lockstep", "row_log"), ("lockstep", "dense")])
def test_steady_state_sampled_against_oracles_emulated(regime, layout):
    """staggered: bench.py's steady state (every launch hosts a few ends of every kind).  lockstep: a fresh VecEnv, whole cohorts
    of lanes reach the time limit in ONE launch, among them lanes whose last step fails -- the regime in which round 5's second
    bug showed (the partner installed the next episode over lag-ring slots the failed step's terminal observation still read)."""
    kind, ckw, skw = FAIL_PRONE
    cfg = configs.reference_like(kind)
    rows = presets.OBS_LOG_ROWS if layout == "row_log" else 0
    lib = _spec_lib(cfg, ckw, skw, rows)
    n = 256 if regime == "staggered" else 448   # (lock-step: ~1 lane in 200 fails ON the step its time limit runs out)
    vec = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=n, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw), seed=11,
                          derived_views=False, obs_log_rows=rows, _backend=HostBackend(), _lib_path=lib)
    assert vec.spec_index == 0 and vec.obs_log_rows == rows        # the two-wave kernel k_step2 of this configuration
    res = cr.steady_state_sampled(vec, cfg, ckw, skw, 11, window=100 if regime == "staggered" else 50, sample=64,
                                  parts=None if regime == "staggered" else 0, what="{} {}".format(regime, layout))
    print(regime, layout, res)
    assert res["failure_ends"] >= 60 and res["time_limit_ends"] >= 150, res
    assert res["failed_on_the_limit_step_checked"] >= 1, res      # (the lanes of round 5's second bug are among the checked)
    assert res["sampled_ends"] >= 50, res
    vec.close()


@pytest.mark.skipif(MUT_SRC is not None, reason="mutants are built as run-time specialisations only")
def test_benched_frozen_kernel_through_its_time_limit_emulated():
    """k_step2<true, 6> -- the instance bench.py's headline runs (c3_cnn_step2_dryden_lean_log, derived_views=False, row log) --
    through steps_max = 2 000: time-limit ends foreseen by the gym wave and installed by the physics wave, the terminal
    observation, the metrics of a 2 000-step episode (42-bit end-error ring, rise/settling latches), the reset observation."""
    name = "c3_cnn_step2_dryden_lean_log"
    _, kind, ckw, skw = [e for e in presets.SPECIALISED if e[0] == name][0]
    cfg = presets.preset(kind)
    n = 12
    vec = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=n, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw), seed=11,
                          as_numpy=True, derived_views=False, _backend=HostBackend(), _lib_path=build_emu())
    assert vec.spec_index == [e[0] for e in presets.SPECIALISED].index(name) and int(vec.cfg["steps_max"]) == 2000
    res = cr.through_time_limit(vec, cfg, ckw, skw, 11, 2030, atol=1.5e-2, what=name)   # (atol: tests/test_gpu_oracle_coverage.py LONG_ATOL)
    print(name, res)
    assert res["episodes"] >= n and res["terminations"].get("steps", 0) >= n - 2
    vec.close()
