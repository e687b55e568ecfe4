"""Profile target: fwg_gae on one PPO rollout of BASELINE configs[4] (65 536 envs x 128 steps) -- run under
`rocprofv3 --kernel-trace --stats`; also prints HIP-event timings.  Algorithmic bytes: 17 B per transition = 142.6 MB."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fixed-wing-gym_amd")]
import torch  # noqa: E402
from gym_fixed_wing import _native as nat  # noqa: E402
from gym_fixed_wing.ppo import gae  # noqa: E402
from gym_fixed_wing.vec_env import _TorchBackend  # noqa: E402

T, N = 128, 65536
mem, lib = _TorchBackend(0), nat.load_library()
g = torch.Generator(device="cuda"); g.manual_seed(0)
rew, val = torch.randn((T, N), device="cuda", generator=g), torch.randn((T, N), device="cuda", generator=g)
done = (torch.rand((T, N), device="cuda", generator=g) < 0.01).to(torch.uint8)
last = torch.randn((N,), device="cuda", generator=g)
adv, ret = torch.empty((T, N), device="cuda"), torch.empty((T, N), device="cuda")
for _ in range(5):
    gae(lib, mem, rew, val, done, last, 0.99, 0.95, adv, ret)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    gae(lib, mem, rew, val, done, last, 0.99, 0.95, adv, ret)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
alg = 17 * T * N
print("fwg_gae {} x {}: {:.1f} us per launch (HIP events, back to back), {:.1f} MB algorithmic -> {:.0f} GB/s = {:.2f} of 8 TB/s".format(
    T, N, us, alg / 1e6, alg / us / 1e3, alg / us / 1e3 / 8000.0))
