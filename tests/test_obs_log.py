"""Row-log observations (FixedWingVecEnv(obs_log_rows=L), include/fwgym.h "Row-log observations"): the zero-copy window
must hold exactly the values of the dense observation batch -- through episode ends, auto-reset, failed simulator
steps, early-episode padding and the wrap of the log -- and the oracle parity must hold in that mode too."""
import copy

import numpy as np
import pytest

import configs
import parity
from gym_fixed_wing.vec_env import FixedWingVecEnv

TURB = {"turbulence": True, "turbulence_intensity": "moderate"}


def _pair(mk, cfg, n, rows, **kw):
    return mk(cfg, n, 0, **kw), mk(cfg, n, rows, **kw)


def _emu_env(cfg, n, rows, **kw):
    from emu.host_backend import HostBackend, build_emu
    return FixedWingVecEnv(copy.deepcopy(cfg), num_envs=n, as_numpy=True, _backend=HostBackend(), _lib_path=build_emu(),
                           obs_log_rows=rows, seed=3, **kw)


def _gpu_env(cfg, n, rows, **kw):
    # (bitwise comparison of the two layouts ON THE SAME KERNEL TIER: the dense variant of these configurations differs from a
    # preset in values only and would land on its shape instance, the 32-row log on the generic kernel; both generic here --
    # the frozen kernels' layouts are compared in test_log_window_specialised_kernel_on_gpu)
    import os
    os.environ["FWGYM_SHAPE"] = "0"
    try:
        return FixedWingVecEnv(copy.deepcopy(cfg), num_envs=n, device=0, as_numpy=True, obs_log_rows=rows, seed=3, **kw)
    finally:
        os.environ.pop("FWGYM_SHAPE", None)


def run_log_vs_dense(mk, n, rows, steps, config_kw, sim_kw=None, auto_reset=True, scale=1.0, burst=2.5):
    cfg = configs.reference_like("cnn")
    dense, log = _pair(mk, cfg, n, rows, config_kw=config_kw, sim_config_kw=sim_kw, auto_reset=auto_reset)
    oa, ob = dense.reset(), log.reset()
    np.testing.assert_array_equal(oa, ob)
    rng = np.random.default_rng(1)
    terms = {}
    alive = np.ones(n, dtype=bool)
    for t in range(steps):
        act = rng.uniform(-1, 1, (n, 3)).astype(np.float32) * (burst if t % 13 == 0 else scale)
        oa, ra, da, ia = dense.step(act)
        ob, rb, db, ib = log.step(act)
        np.testing.assert_array_equal(da, db)
        np.testing.assert_array_equal(ra[alive], rb[alive])
        np.testing.assert_array_equal(oa[alive], ob[alive], err_msg="step {}".format(t))
        for i in np.nonzero(da & alive)[0]:
            terms[ia[int(i)]["termination"]] = terms.get(ia[int(i)]["termination"], 0) + 1
            if auto_reset:
                np.testing.assert_array_equal(ia[int(i)]["terminal_observation"], ib[int(i)]["terminal_observation"])
        if not auto_reset:
            alive &= ~da.astype(bool)   # a finished env that is not reset is outside the contract
            if not alive.any():
                break
    dense.close(), log.close()
    return terms


@pytest.mark.parametrize("rows", [8, 11, 32])
def test_log_window_equals_dense_batch_emulated(rows):
    terms = run_log_vs_dense(_emu_env, 7, rows, 150, {"observation": {"step": 2}, "steps_max": 37}, TURB)
    assert terms.get("steps", 0) >= 14


def test_log_window_through_failed_steps_emulated():
    # tight rate constraints (list index 6 = omega_p, as in configs.CASES "fail_prone"): episodes end in failed simulator
    # steps, the branch whose observation takes its state rows from one record further back
    tight = {"observation": {"step": 2}, "steps_max": 60,
             "simulator": {"states": {6: {"constraint_min": -40, "constraint_max": 40}}}}
    terms = run_log_vs_dense(_emu_env, 9, 12, 200, tight, None)
    assert any(k not in ("steps", "success") for k in terms), terms


def test_log_window_without_auto_reset_emulated():
    run_log_vs_dense(_emu_env, 5, 9, 80, {"observation": {"step": 3}, "steps_max": 25}, None, auto_reset=False)


def test_log_mode_oracle_parity_emulated():
    from emu.host_backend import HostBackend, build_emu
    cfg = configs.reference_like("cnn")
    ckw = {"observation": {"step": 2}, "steps_max": 40}
    vec = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=4, as_numpy=True, _backend=HostBackend(), _lib_path=build_emu(),
                          obs_log_rows=12, seed=11, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(TURB))
    oracles = parity.make_oracles(cfg, 4, 11, config_kw=ckw, sim_config_kw=TURB)
    rng = np.random.default_rng(5)
    acts = rng.uniform(-1, 1, (100, 4, 3)).astype(np.float32)
    res = parity.run_gym_parity(vec, oracles, 100, lambda t: acts[t], rtol=4e-3, atol=4e-3)
    assert res["episodes"] >= 8


def test_log_mode_rejects_what_it_cannot_represent():
    from emu.host_backend import HostBackend, build_emu
    from gym_fixed_wing import _native as nat
    with pytest.raises(nat.NativeError):   # vector observation: nothing to stack
        FixedWingVecEnv(configs.reference_like("examples"), num_envs=2, _backend=HostBackend(), _lib_path=build_emu(), obs_log_rows=8)
    with pytest.raises(nat.NativeError):   # too short for the window
        FixedWingVecEnv(configs.reference_like("cnn"), num_envs=2, _backend=HostBackend(), _lib_path=build_emu(), obs_log_rows=4)


@pytest.mark.gpu
def test_log_window_equals_dense_batch_on_gpu():
    tight = {"observation": {"step": 2}, "steps_max": 90,
             "simulator": {"states": {6: {"constraint_min": -60, "constraint_max": 60}}}}
    terms = run_log_vs_dense(_gpu_env, 1000, 32, 260, tight, TURB)
    assert terms.get("steps", 0) >= 100
    assert any(k not in ("steps", "success") for k in terms), terms   # failed steps occurred too


@pytest.mark.gpu
def test_log_window_specialised_kernel_on_gpu():
    from gym_fixed_wing import presets
    cfg, ckw, skw, _, _ = presets.workload("c3")
    dense = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=4096, device=0, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw),
                            derived_views=False, seed=2, obs_log_rows=0)
    log = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=4096, device=0, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(skw),
                          derived_views=False, seed=2, obs_log_rows=presets.OBS_LOG_ROWS)
    assert dense.spec_index >= 0 and log.spec_index >= 0 and dense.spec_index != log.spec_index
    import torch
    oa, ob = dense.reset(), log.reset()
    assert torch.equal(oa, ob.contiguous())
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    ndone = 0
    for t in range(2030):   # past steps_max = 2000 of the preset: every env ends an episode and is reset in-kernel
        act = (torch.rand((4096, 3), device="cuda", generator=gen) * 2 - 1) * (2.5 if t % 11 == 0 else 1.0)
        oa, ra, da = dense.step_device(act)
        ob, rb, db = log.step_device(act)
        assert torch.equal(da, db) and torch.equal(ra, rb)
        assert torch.equal(oa.reshape(4096, 5, 12), ob), "step {}".format(t)
        ndone += int(da.sum())
    assert ndone >= 4096


@pytest.mark.gpu
def test_log_window_under_graph_replay_on_gpu(monkeypatch):
    """Row-log positions come from the device-resident step counter under hipGraph replay; the host view follows."""
    import torch
    monkeypatch.setenv("FWGYM_SHAPE", "0")   # (bitwise comparison: both layouts on the generic kernel, see _gpu_env)
    cfg = configs.reference_like("cnn")
    kw = dict(config_kw={"observation": {"step": 2}, "steps_max": 50}, sim_config_kw=copy.deepcopy(TURB), seed=9)
    dense = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=512, device=0, obs_log_rows=0, **copy.deepcopy(kw))
    log = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=512, device=0, obs_log_rows=12, **copy.deepcopy(kw))
    dense.reset(), log.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    acts = [torch.rand((512, 3), device="cuda", generator=gen) * 2 - 1 for _ in range(8)]
    log.set_graph_mode(True)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for a in acts[:2]:
            log.step_device(a)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    for a in acts[:2]:
        dense.step_device(a)
    g = torch.cuda.CUDAGraph()
    log.capture_begin()
    with torch.cuda.graph(g):
        for a in acts:
            log.step_device(a)
    log.capture_end()
    for rep in range(9):   # 72 steps: past steps_max = 50 and several wraps of the 12-row log
        g.replay(); log.note_replayed_steps(8); torch.cuda.synchronize()
        for a in acts:
            od, _, _ = dense.step_device(a)
        assert torch.equal(od.reshape(512, 5, 12), log.obs_dense().reshape(512, 5, 12)), "replay {}".format(rep)


@pytest.mark.gpu
def test_zero_copy_windows_stay_right_under_graph_replay_on_gpu():
    """View mode (the default): the windows handed out while a chunk is CAPTURED are host-side views; they stay right under
    replay because the chunk (64 steps) is a whole number of window periods (obs_step x depth = 2 x 32 for the default log).
    A consumer inside the graph copies every step's window into a buffer; seven replays (448 steps, through steps_max = 50 nine
    times) against the dense layout stepped directly."""
    import torch
    from gym_fixed_wing import presets
    cfg = configs.reference_like("cnn")
    kw = dict(config_kw={"observation": {"step": 2}, "steps_max": 50}, sim_config_kw=copy.deepcopy(TURB), seed=9)
    dense = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=512, device=0, obs_log_rows=0, **copy.deepcopy(kw))
    log = FixedWingVecEnv(copy.deepcopy(cfg), num_envs=512, device=0, **copy.deepcopy(kw))
    assert log.obs_log_rows == presets.OBS_LOG_ROWS and log.obs_window_period == 64
    dense.reset(), log.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    K = 64
    acts = [torch.rand((512, 3), device="cuda", generator=gen) * 2 - 1 for _ in range(K)]
    log.set_graph_mode(True, obs="view")   # (opt-in: the default hands out gathered copies, right for any chunk length)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for a in acts[:2]:
            log.step_device(a)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    for a in acts[:2]:
        dense.step_device(a)
    seen = torch.zeros((K, 512, 5, 12), device="cuda")
    g = torch.cuda.CUDAGraph()
    parity = log.capture_begin(n_steps=K)
    with torch.cuda.graph(g):
        for t, a in enumerate(acts):
            o, _, _ = log.step_device(a)
            seen[t].copy_(o)          # the consumer: reads the zero-copy window of step t
    assert log.capture_end() is parity and parity.uses_views and parity.gstep == 2
    for rep in range(7):
        log.replay_check(parity)
        g.replay(); log.note_replayed_steps(K); torch.cuda.synchronize()
        for t, a in enumerate(acts):
            od, _, _ = dense.step_device(a)
            assert torch.equal(od.reshape(512, 5, 12), seen[t]), "replay {} step {}".format(rep, t)
        assert torch.equal(od.reshape(512, 5, 12), log._obs.reshape(512, 5, 12))   # the host view after the replay
    # two direct steps move the phase: the captured views would be stale, and the check says so
    log.step_device(acts[0]), log.step_device(acts[1])
    with pytest.raises(RuntimeError, match="window periods"):
        log.replay_check(parity)
    dense.close(), log.close()


def test_replay_check_guards_the_phase_of_captured_windows_emulated():
    """The same guard on the host emulation (captured calls execute at once there): a chunk that is not a multiple of the
    window period may be captured, but not replayed once its views were handed out; gather mode has no such limit."""
    cfg = configs.reference_like("cnn")
    vec = _emu_env(cfg, 5, 10, config_kw={"observation": {"step": 2}, "steps_max": 30}, sim_config_kw=copy.deepcopy(TURB))
    assert vec.obs_window_period == 2 * (10 - 4)
    vec.reset()
    a = np.zeros((5, 3), dtype=np.float32)
    for mode, ok in (("view", False), ("gather", True)):
        vec.set_graph_mode(True, obs=mode)
        parity = vec.capture_begin()
        for _ in range(4):
            vec.step_device(a)
        vec.capture_end()
        vec.replay_check(parity)            # the first replay starts where the capture did
        vec.note_replayed_steps(4)          # (the emulation executed the captured calls)
        if ok:
            vec.replay_check(parity)
        else:
            with pytest.raises(RuntimeError, match="window periods"):
                vec.replay_check(parity)    # 4 steps later: not a whole number of 12-step periods
        vec.set_graph_mode(False)
    # a chunk of one period replays for ever, and want_obs=False never pins the phase
    vec.set_graph_mode(True, obs="view")
    with pytest.raises(ValueError, match="multiple of the observation window period"):
        vec.capture_begin(n_steps=4)        # (told the length up front, the capture is refused instead of its second replay)
    period = vec.capture_begin(n_steps=12)
    for _ in range(12):
        vec.step_device(a)
    vec.capture_end()
    vec.note_replayed_steps(12)             # (the emulation executed the captured calls)
    # ... and a SECOND graph of the same env, captured later, has a token of its own: the first one's phase rule is not lost
    blind = vec.capture_begin()
    for _ in range(4):
        vec.step_device(a, want_obs=False)
    vec.capture_end()
    vec.note_replayed_steps(4)
    assert period.uses_views and not blind.uses_views and period.gstep + 12 == blind.gstep
    with pytest.raises(RuntimeError, match="window periods"):
        vec.replay_check(period)            # 16 steps after its capture: off phase, and still guarded
    for _ in range(2):
        vec.replay_check(blind)
        vec.note_replayed_steps(4)          # 24 steps past the capture of `period`: two whole periods
    for _ in range(3):
        vec.replay_check(period)
        vec.note_replayed_steps(12)
    # the default mode (gather) accepts any chunk length
    vec.set_graph_mode(False)
    vec.set_graph_mode(True)
    tok = vec.capture_begin(n_steps=4)
    for _ in range(4):
        vec.step_device(a)
    vec.capture_end()
    assert not tok.uses_views
    for _ in range(3):
        vec.replay_check(tok)
        vec.note_replayed_steps(4)
    vec.close()


def test_row_log_env_accepts_curriculum_and_simulator_updates_emulated():
    """set_curriculum_level / set_simulator_attr re-upload the configuration: the row-log layout must survive that
    (the re-compile has to carry obs_log_rows, otherwise fwg_update_config sees another state layout)."""
    cfg = configs.reference_like("cnn")
    vec = _emu_env(cfg, 5, 10, config_kw={"observation": {"step": 2}, "steps_max": 30}, sim_config_kw=copy.deepcopy(TURB))
    dense = _emu_env(cfg, 5, 0, config_kw={"observation": {"step": 2}, "steps_max": 30}, sim_config_kw=copy.deepcopy(TURB))
    for v in (vec, dense):
        v.reset()
        v.set_curriculum_level(0.5)
        v.set_simulator_attr("turbulence_intensity", "severe")
    rng = np.random.default_rng(0)
    for t in range(45):
        act = rng.uniform(-1, 1, (5, 3)).astype(np.float32)
        oa, ra, da, _ = dense.step(act)
        ob, rb, db, _ = vec.step(act)
        np.testing.assert_array_equal(oa, ob)
        np.testing.assert_array_equal(ra, rb)
    assert vec.obs_log_rows == 10
    vec.close(), dense.close()


def test_default_layout_is_the_row_log_where_it_applies_emulated():
    from emu.host_backend import HostBackend, build_emu
    from gym_fixed_wing import presets
    mk = lambda kind, **kw: FixedWingVecEnv(configs.reference_like(kind), num_envs=3, as_numpy=True, _backend=HostBackend(),
                                            _lib_path=build_emu(), **kw)
    assert mk("cnn").obs_log_rows == presets.OBS_LOG_ROWS            # lagged matrix observation, no noise
    assert mk("cnn", obs_log_rows=0).obs_log_rows == 0               # explicit dense batch
    assert mk("examples").obs_log_rows == 0                          # nothing to stack
    assert mk("default").obs_log_rows == 0                           # observation noise re-draws every row each step
    noisy = configs.reference_like("cnn")
    noisy["observation"]["noise"] = {"mean": 0, "var": 0.1}
    assert FixedWingVecEnv(noisy, num_envs=3, as_numpy=True, _backend=HostBackend(), _lib_path=build_emu()).obs_log_rows == 0
    # the same choice by name
    assert mk("cnn", obs_layout="dense").obs_log_rows == 0
    assert mk("cnn", obs_layout="row_log").obs_log_rows == presets.OBS_LOG_ROWS
    assert mk("cnn", obs_layout="auto").obs_log_rows == presets.OBS_LOG_ROWS
    with pytest.raises(ValueError):
        mk("cnn", obs_layout="dense", obs_log_rows=8)
    with pytest.raises(ValueError):
        mk("examples", obs_layout="row_log")
    with pytest.raises(ValueError):
        mk("cnn", obs_layout="sparse")


@pytest.mark.parametrize("kind,rows,ckw", [("cnn", 10, {"observation": {"step": 2}, "steps_max": 23}),
                                           ("cnn", 0, {"observation": {"step": 2}, "steps_max": 23}),
                                           ("default", 0, {"steps_max": 17})])
def test_device_resident_positions_equal_host_positions_emulated(kind, rows, ckw):
    """Graph mode keeps the ring positions on the device and advances them by increments (no division); a run in that
    mode must be indistinguishable from a run with host-computed positions -- incl. the row-log window that consumers
    gather on the device (fwg_obs_gather) -- over many wraps of every ring and through resets."""
    cfg = configs.reference_like(kind)
    a = _emu_env(cfg, 6, rows, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(TURB))
    b = _emu_env(cfg, 6, rows, config_kw=copy.deepcopy(ckw), sim_config_kw=copy.deepcopy(TURB))
    oa, ob = a.reset(), b.reset()
    rng = np.random.default_rng(3)
    for t in range(7):   # an odd number of direct steps first: the device copy starts at an odd parity
        act = rng.uniform(-1, 1, (6, 3)).astype(np.float32)
        a.step(act), b.step(act)
    b.set_graph_mode(True)
    np.testing.assert_array_equal(np.asarray(a._obs).reshape(6, -1), np.asarray(b._obs).reshape(6, -1))
    for t in range(131):
        act = rng.uniform(-1, 1, (6, 3)).astype(np.float32) * (2.5 if t % 13 == 0 else 1.0)
        oa, ra, da, ia = a.step(act)
        ob, rb, db, ib = b.step(act)
        np.testing.assert_array_equal(da, db)
        np.testing.assert_array_equal(ra, rb)
        np.testing.assert_array_equal(np.asarray(oa).reshape(6, -1), np.asarray(ob).reshape(6, -1), err_msg="step {}".format(t))
        if t == 60:   # a masked reset in the middle (k_reset derives the positions of the last completed step itself)
            oa, ob = a.reset(indices=[1, 4]), b.reset(indices=[1, 4])
            np.testing.assert_array_equal(np.asarray(oa).reshape(6, -1), np.asarray(ob).reshape(6, -1))
    b.set_graph_mode(False)   # back to host positions: the host count is taken from the device
    for t in range(9):
        act = rng.uniform(-1, 1, (6, 3)).astype(np.float32)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        np.testing.assert_array_equal(np.asarray(oa).reshape(6, -1), np.asarray(ob).reshape(6, -1))
    a.close(), b.close()

