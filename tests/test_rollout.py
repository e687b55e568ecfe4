"""PPO-style rollout collection (BASELINE.json configs[4]): device-resident normalisation + MLP policy + env step."""
import numpy as np
import pytest
import torch

import configs
from gym_fixed_wing.rollout import MlpPolicy, RunningMeanStd, VecNormalizeDevice, collect_rollout
from gym_fixed_wing.vec_env import FixedWingVecEnv


def test_running_mean_std_matches_numpy():
    rng = np.random.default_rng(0)
    rms = RunningMeanStd((3,), epsilon=1e-8)
    chunks = [rng.normal(2.0, 3.0, size=(n, 3)).astype(np.float32) for n in (5, 17, 64)]
    for c in chunks:
        rms.update(torch.from_numpy(c))
    allx = np.concatenate(chunks)
    np.testing.assert_allclose(rms.mean.numpy(), allx.mean(axis=0), rtol=1e-4)
    np.testing.assert_allclose(rms.var.numpy(), allx.var(axis=0), rtol=1e-3)


def test_rollout_on_emulated_env():
    from emu.host_backend import HostBackend, build_emu
    cfg = configs.reference_like("examples")
    vec = FixedWingVecEnv(cfg, num_envs=6, config_kw={"steps_max": 10}, as_numpy=True, _backend=HostBackend(),
                          _lib_path=build_emu())
    obs = vec.reset()
    torch.manual_seed(0)
    policy = MlpPolicy(12)
    norm = VecNormalizeDevice((12,), 6)
    buf, last = collect_rollout(vec, policy, norm, 16, obs=torch.from_numpy(np.asarray(obs)))
    assert buf["obs"].shape == (16, 6, 12) and buf["actions"].shape == (16, 6, 3)
    assert torch.isfinite(buf["obs"]).all() and torch.isfinite(buf["rewards"]).all()
    assert buf["dones"].sum() == 6                     # every env hits steps_max=10 once in 16 steps
    assert float(buf["obs"].abs().max()) <= 10.0       # clipped normalised observations


@pytest.mark.gpu
def test_rollout_throughput_on_gpu():
    import time
    cfg = configs.reference_like("examples")
    n = 65536
    vec = FixedWingVecEnv(cfg, num_envs=n, device=0, derived_views=False)
    vec.reset()
    policy = MlpPolicy(12).cuda()
    norm = VecNormalizeDevice((12,), n, device="cuda")
    collect_rollout(vec, policy, norm, 16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    buf, _ = collect_rollout(vec, policy, norm, 128)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rate = 128 * n / dt
    print("C5 rollout: {:.3e} env-steps/s end to end (policy forward + normalisation + env step), {:.1f} us/step".format(rate, dt / 128 * 1e6))
    assert torch.isfinite(buf["rewards"]).all() and rate > 1e7
    # the same rollout as ONE replayed hipGraph
    from gym_fixed_wing.rollout import GraphedRollout
    g = GraphedRollout(vec, policy, norm, 128)
    g.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        buf = g.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("C5 rollout, hipGraph replay: {:.3e} env-steps/s, {:.1f} us/step".format(128 * n / dt, dt / 128 * 1e6))
    assert torch.isfinite(buf["rewards"]).all() and torch.isfinite(buf["obs"]).all()
    assert float(norm.obs_rms.var.min()) > 0 and torch.isfinite(norm.obs_rms.mean).all()
    steps = vec.get_state(["steps_count"])["steps_count"]
    assert steps.min() >= 0 and steps.max() <= vec.cfg["steps_max"]


@pytest.mark.gpu
def test_running_statistics_survive_graph_replay():
    """The running mean/variance must come out the same from a replayed hipGraph as from NumPy in float64 (torch's
    own dim-0 reductions do not on this stack from the second replay on, which is why RunningMeanStd avoids them)."""
    n, d, k = 65536, 12, 6
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    data = torch.randn((k, n, d), device="cuda", generator=gen) * torch.linspace(0.5, 6.0, d, device="cuda") + 20.0
    rms = RunningMeanStd((d,), device="cuda")
    ret = RunningMeanStd((), device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        rms.update(data[0]), ret.update(data[0, :, 0])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(k):
            rms.update(data[i])
            ret.update(data[i, :, 0])
    replays = 4
    for _ in range(replays):
        graph.replay()
    torch.cuda.synchronize()
    x = data.double().cpu().numpy()
    allx = np.concatenate([x[0]] + [x.reshape(-1, d)] * replays)
    np.testing.assert_allclose(rms.mean.cpu().numpy(), allx.mean(axis=0), rtol=2e-4)
    np.testing.assert_allclose(rms.var.cpu().numpy(), allx.var(axis=0), rtol=2e-3, atol=1e-4)
    np.testing.assert_allclose(float(ret.mean), allx[:, 0].mean(), rtol=2e-4)
    np.testing.assert_allclose(float(ret.var), allx[:, 0].var(), rtol=2e-3)
