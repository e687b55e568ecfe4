"""N>1 path on CPU: two gloo ranks, each with its shard of the envs (host-emulation build of the kernels), check that
(1) trajectories do not depend on the sharding (global env ids drive the RNG streams) and (2) the success-metric
all-gather reproduces the single-process totals and drives the curriculum identically on every rank."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

TOTAL, STEPS = 12, 45
CKW = {"steps_max": 20, "target": {"success_streak_req": 5, "success_streak_fraction": 0.6,
                                   "states": {0: {"bound": 100}, 1: {"bound": 45}, 2: {"bound": 12}}}}


def _run(vec, first, acts):
    vec.reset()
    obs_hist, totals = [], np.zeros(16)
    for t in range(STEPS):
        obs, rew, done, _ = vec.step(acts[t, first:first + vec.num_envs])
        obs_hist.append(np.array(obs))
    return np.stack(obs_hist)


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu.host_backend import HostBackend, build_emu
    from gym_fixed_wing import distributed as fd
    from gym_fixed_wing import presets
    from gym_fixed_wing.vec_env import FixedWingVecEnv
    acts = np.random.default_rng(0).uniform(-1, 1, size=(STEPS, TOTAL, 3)).astype(np.float32)
    first, n = fd.shard(TOTAL, rank, world)
    vec = FixedWingVecEnv(presets.default(), num_envs=n, config_kw=CKW, seed=5, env_id_base=first, as_numpy=True,
                          _backend=HostBackend(), _lib_path=build_emu())
    obs = _run(vec, first, acts)
    summary = fd.gather_success(vec)
    sched = fd.CurriculumSchedule(level=0.25, cooldown=0)
    level = sched.update(vec, summary)
    np.save(os.path.join(out_dir, "obs_{}.npy".format(rank)), obs)
    np.save(os.path.join(out_dir, "sum_{}.npy".format(rank)),
            np.array([summary["episodes"], summary["success"]["all"], summary["success"]["roll"], level]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_helpers():
    from gym_fixed_wing import distributed as fd
    assert [fd.shard(10, r, 3) for r in range(3)] == [(0, 4), (4, 3), (7, 3)]
    assert fd.shard(262144, 7, 8) == (229376, 32768)


def test_two_rank_sharded_run_matches_single_process(tmp_path):
    sys.path.insert(0, HERE)
    from emu.host_backend import HostBackend, build_emu
    from gym_fixed_wing import distributed as fd
    from gym_fixed_wing import presets
    from gym_fixed_wing.vec_env import FixedWingVecEnv
    lib = build_emu()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    acts = np.random.default_rng(0).uniform(-1, 1, size=(STEPS, TOTAL, 3)).astype(np.float32)
    vec = FixedWingVecEnv(presets.default(), num_envs=TOTAL, config_kw=CKW, seed=5, as_numpy=True,
                          _backend=HostBackend(), _lib_path=lib)
    ref_obs = _run(vec, 0, acts)
    ref = fd.summarize(vec.reduce_success(), vec.target_names)
    o0, o1 = np.load(tmp_path / "obs_0.npy"), np.load(tmp_path / "obs_1.npy")
    np.testing.assert_array_equal(np.concatenate([o0, o1], axis=1), ref_obs)   # bitwise: sharding-independent
    s0, s1 = np.load(tmp_path / "sum_0.npy"), np.load(tmp_path / "sum_1.npy")
    np.testing.assert_array_equal(s0, s1)                                      # every rank sees the global summary
    assert s0[0] == ref["episodes"] == 2 * TOTAL
    assert s0[1] == pytest.approx(ref["success"]["all"]) and s0[2] == pytest.approx(ref["success"]["roll"])


# ---- the learner over two ranks (data-parallel PPO: one all-reduce of the gradient per minibatch step) ------------------------
def _ppo_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "fixed-wing-gym_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu.host_backend import HostBackend, build_emu
    from gym_fixed_wing import distributed as fd
    from gym_fixed_wing import presets
    from gym_fixed_wing.ppo import PPO
    from gym_fixed_wing.vec_env import FixedWingVecEnv
    first, n = fd.shard(128, rank, world)
    vec = FixedWingVecEnv(presets.preset("examples"), num_envs=n, config_kw={"steps_max": 12}, seed=5, env_id_base=first,
                          _backend=HostBackend(), _lib_path=build_emu())
    vec.set_curriculum_level(0.25)
    vec.reset()
    ppo = PPO(vec, seed=100 + rank, n_steps=8, nminibatches=2, noptepochs=1,          # (different seeds: rank 0's weights are broadcast)
              curriculum=fd.CurriculumSchedule(level=0.25, cooldown=0))
    w0 = torch.cat([p.detach().reshape(-1) for p in ppo.policy.parameters()]).clone()
    logs = []
    ppo.learn(3 * 8 * 128, log=logs.append)
    w1 = torch.cat([p.detach().reshape(-1) for p in ppo.policy.parameters()])
    torch.save({"w0": w0, "w1": w1, "timesteps": ppo.num_timesteps, "updates": ppo.updates, "level": ppo.curriculum.level,
                "episodes": [l["episodes"] for l in logs]}, os.path.join(out_dir, "ppo_{}.pt".format(rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_learner_keeps_one_set_of_weights(tmp_path):
    """Both ranks start from rank 0's weights, average their gradients at every minibatch step and therefore hold the SAME
    weights after every update; timesteps count all ranks' transitions; the episode counts and the curriculum level are global."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_ppo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "ppo_0.pt"), torch.load(tmp_path / "ppo_1.pt")
    assert torch.equal(a["w0"], b["w0"])                      # the broadcast
    assert not torch.equal(a["w0"], a["w1"])                  # the updates happened
    assert torch.allclose(a["w1"], b["w1"], rtol=0, atol=1e-7), float((a["w1"] - b["w1"]).abs().max())
    assert a["timesteps"] == b["timesteps"] == 3 * 8 * 128 and a["updates"] == b["updates"] == 3
    assert a["episodes"] == b["episodes"] and sum(a["episodes"]) >= 128 and a["level"] == b["level"]
