"""On-device rollout collection for PPO-style training on FixedWingVecEnv (BASELINE.json
configs[4]).  Replaces the data path of the reference's training script -- `VecNormalize(SubprocVecEnv(...))` feeding
`PPO2(MlpPolicy)` (examples/train_rl_controller.py:223-231) -- with device-resident pieces: running observation /
return normalisation (VecNormalize), a 64-64 tanh MLP policy + value net (the architecture of the shipped
models/mlp_controller), and a rollout buffer filled without leaving the GPU.  The optimiser step is out of scope."""
import math

import numpy as np
import torch
from torch import nn


class RunningMeanStd(object):
    """Batched parallel-variance update (what stable-baselines' VecNormalize keeps as obs_rms / ret_rms)."""

    def __init__(self, shape, device=None, epsilon=1e-4):
        self.mean = torch.zeros(shape, dtype=torch.float32, device=device)
        self.var = torch.ones(shape, dtype=torch.float32, device=device)
        self.count = torch.full((), epsilon, dtype=torch.float32, device=device)   # a tensor: graph-capturable
        self._ones = None

    def update(self, x):
        # Batch moments through a ones-row GEMM on the deviations from the running mean, not Tensor.mean/var: torch's
        # multi-block dim-0 reductions return garbage from the second replay of a captured hipGraph on this stack
        # (ROCm 7.x / torch 2.10; found with tests/test_rollout.py, negative running variances), and the centred
        # form is also the numerically better one in fp32.
        x = x.reshape(-1, max(1, self.mean.numel()))
        b_count = x.shape[0]
        if self._ones is None or self._ones.shape[1] != b_count:
            self._ones = torch.ones((1, b_count), dtype=torch.float32, device=x.device)
        d = x - self.mean.reshape(1, -1)
        s1 = (self._ones @ d).reshape(self.mean.shape) / b_count
        s2 = (self._ones @ (d * d)).reshape(self.mean.shape) / b_count
        b_mean, b_var = self.mean + s1, (s2 - s1 * s1).clamp_min(0.0)
        delta = b_mean - self.mean
        tot = self.count + b_count
        m2 = self.var * self.count + b_var * b_count + delta * delta * (self.count * b_count / tot)
        self.mean.add_(delta * (b_count / tot))      # in place: the buffers keep their addresses under graph replay
        self.var.copy_(m2 / tot)
        self.count.copy_(tot)


class VecNormalizeDevice(object):
    def __init__(self, obs_shape, num_envs, device=None, clip_obs=10.0, clip_reward=10.0, gamma=0.99, training=True):
        self.obs_rms = RunningMeanStd(obs_shape, device)
        self.ret_rms = RunningMeanStd((), device)
        self.ret = torch.zeros(num_envs, dtype=torch.float32, device=device)
        self.clip_obs, self.clip_reward, self.gamma, self.training = clip_obs, clip_reward, gamma, training

    def obs(self, obs):
        if self.training:
            self.obs_rms.update(obs)
        return ((obs - self.obs_rms.mean) / torch.sqrt(self.obs_rms.var + 1e-8)).clamp(-self.clip_obs, self.clip_obs)

    def reward(self, rew, done):
        self.ret.mul_(self.gamma).add_(rew)
        if self.training:
            self.ret_rms.update(self.ret)
        out = (rew / torch.sqrt(self.ret_rms.var + 1e-8)).clamp(-self.clip_reward, self.clip_reward)
        self.ret.mul_(1.0 - done.to(self.ret.dtype))
        return out


class MlpPolicy(nn.Module):
    """Separate 64-64 tanh networks for the Gaussian policy mean and the value (stable-baselines MlpPolicy layout of
    the shipped MLP controller: pi_fc0/pi_fc1/pi, vf_fc0/vf_fc1/vf), state-independent log-std."""

    def __init__(self, obs_dim, act_dim=3, hidden=64):
        super().__init__()
        self.pi = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(),
                                nn.Linear(hidden, act_dim))
        self.vf = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(),
                                nn.Linear(hidden, 1))
        self.log_std = nn.Parameter(torch.zeros(act_dim))

    @torch.no_grad()
    def act(self, obs, deterministic=False):
        mean = self.pi(obs)
        value = self.vf(obs).squeeze(-1)
        if deterministic:
            return mean, value, torch.zeros(obs.shape[0], device=obs.device)
        std = self.log_std.exp()
        noise = torch.randn_like(mean)
        action = mean + std * noise
        logp = (-0.5 * noise * noise - self.log_std - 0.5 * math.log(2 * math.pi)).sum(dim=-1)
        return action, value, logp


def collect_rollout(vec, policy, norm, n_steps, obs=None):
    """n_steps of policy inference + env step for all envs, everything on the device.  Returns the rollout buffer
    {obs, actions, values, logp, rewards, dones} ([n_steps, N, ...]) and the last normalised observation."""
    N = vec.num_envs
    raw = vec._obs.reshape(N, -1) if obs is None else obs
    raw = torch.as_tensor(raw)
    dev = raw.device
    buf = {"obs": torch.empty((n_steps, N, raw.shape[1]), device=dev), "actions": torch.empty((n_steps, N, 3), device=dev),
           "values": torch.empty((n_steps, N), device=dev), "logp": torch.empty((n_steps, N), device=dev),
           "rewards": torch.empty((n_steps, N), device=dev), "dones": torch.empty((n_steps, N), dtype=torch.uint8, device=dev)}
    cur = norm.obs(raw)
    for t in range(n_steps):
        action, value, logp = policy.act(cur)
        buf["obs"][t], buf["actions"][t], buf["values"][t], buf["logp"][t] = cur, action, value, logp
        if hasattr(vec, "step_device") and isinstance(vec._obs, torch.Tensor):
            o, r, d = vec.step_device(action.contiguous())
        else:
            o, r, d, _ = vec.step(action.cpu().numpy())
            o, r, d = torch.as_tensor(o).reshape(N, -1), torch.as_tensor(r), torch.as_tensor(d)
        buf["rewards"][t] = norm.reward(torch.as_tensor(r).to(dev), torch.as_tensor(d).to(dev))
        buf["dones"][t] = torch.as_tensor(d).to(dev)
        cur = norm.obs(torch.as_tensor(o).reshape(N, -1).to(dev))
    return buf, cur


class GraphedRollout(object):
    """The whole n-step rollout (policy forward, normalisation, fused env step, buffer writes) captured ONCE into a
    hipGraph (torch.cuda.CUDAGraph) and replayed: no per-kernel host launch cost.  Needs the env's graph mode (the
    global step counter lives on the device, see include/fwgym.h) and an even n_steps."""

    def __init__(self, vec, policy, norm, n_steps):
        assert n_steps % 2 == 0, "graph mode double-buffers the step counter: capture an even number of steps"
        self.vec, self.policy, self.norm, self.n_steps = vec, policy, norm, n_steps
        N, D = vec.num_envs, vec.obs_dim
        dev = vec._obs.device
        self.buf = {"obs": torch.empty((n_steps, N, D), device=dev), "actions": torch.empty((n_steps, N, 3), device=dev),
                    "values": torch.empty((n_steps, N), device=dev), "logp": torch.empty((n_steps, N), device=dev),
                    "rewards": torch.empty((n_steps, N), device=dev),
                    "dones": torch.empty((n_steps, N), dtype=torch.uint8, device=dev)}
        self.action = torch.zeros((N, 3), device=dev)
        # row-log envs: zero-copy windows only when the chunk is a whole number of window periods (they stay right under replay
        # then); any other length gets the gathered copy per step
        period = getattr(vec, "obs_window_period", 0)
        vec.set_graph_mode(True, obs="view" if (period and n_steps % period == 0) else "gather")
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):          # warm-up outside capture (allocations, lazy initialisation)
            self._body(2)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        self._parity = vec.capture_begin(n_steps)
        with torch.cuda.graph(self.graph):
            self._body(n_steps)
        vec.capture_end()

    def _body(self, n):
        vec, buf = self.vec, self.buf
        N = vec.num_envs
        cur = self.norm.obs(vec._obs.reshape(N, -1))
        for t in range(n):
            action, value, logp = self.policy.act(cur)
            self.action.copy_(action)
            buf["obs"][t].copy_(cur); buf["actions"][t].copy_(action); buf["values"][t].copy_(value); buf["logp"][t].copy_(logp)
            o, r, d = vec.step_device(self.action)
            buf["rewards"][t].copy_(self.norm.reward(r, d))
            buf["dones"][t].copy_(d)
            cur = self.norm.obs(o.reshape(N, -1))

    def run(self):
        self.vec.replay_check(self._parity)
        self.graph.replay()
        self.vec.note_replayed_steps(self.n_steps)
        return self.buf


# ----------------------------------------------------------------------------------------------------------------------
# The same rollout on the HIP rollout head (gym_fixed_wing.actor.DeviceActor): three launches per step
# ----------------------------------------------------------------------------------------------------------------------
class FusedRollout(object):
    """n-step rollouts of FixedWingVecEnv under a DeviceActor, filling {obs, actions, values, logp, rewards, dones}
    ([n_steps, N, ...], normalised observations/rewards as VecNormalize + PPO2 store them) plus `last_value` (the
    bootstrap value of the observation after the last step).  Per step: fwg_step (which also leaves the batch moments for
    the attached head) and fwg_actor_act --
    the act that follows step t writes the rollout slice t+1 directly, so nothing is copied except the carried-over
    slice 0.  `graph=True` captures the whole rollout into one hipGraph (n_steps must be even).

    `fused` (opt-in: fused=True, fused="auto" = wherever fwg_rollout_available, or FWGYM_ROLLOUT_FUSED=1): act t and env step t run as ONE launch (fwg_rollout_step) -- a rollout is
    then [env step 0] [act + step] x (n - 1) [act n]: n + 1 launches instead of 2 n, bit-identical buffers.  `tap(t, obs, rew,
    done)` (eager mode only) is called after every env step with the env's raw output tensors (tests, logging)."""

    def __init__(self, vec, actor, n_steps, graph=False, fused=None, tap=None):
        self.vec, self.actor, self.n_steps = vec, actor, int(n_steps)
        m, N, D, A = vec._mem, vec.num_envs, vec.obs_dim, actor.act_dim
        self.buf = {"obs": m.zeros((n_steps, N, D)), "actions": m.zeros((n_steps, N, A)), "values": m.zeros((n_steps, N)),
                    "logp": m.zeros((n_steps, N)), "rewards": m.zeros((n_steps, N)), "dones": m.zeros((n_steps, N), "u8")}
        # the head's outputs for the CURRENT observation, carried from one rollout to the next
        self.cur = {"obs": m.zeros((N, D)), "actions": m.zeros((N, A)), "values": m.zeros((N,)), "logp": m.zeros((N,))}
        self.last_value = self.cur["values"]
        self._primed = False
        self._graph = None
        # the env step kernel leaves the batch moments itself (two launches per step) -- except with row-log observations,
        # where the lagged rows never pass through the step kernel: then the head reads the window out of the log in place
        # (fwg_actor_set_obs_log: position read on the device, replay-safe) and takes the moments in a launch of its own
        self._attached = not getattr(vec, "obs_log_rows", 0)
        if self._attached:
            actor.attach(vec)
        else:
            actor.set_obs_log(vec)
        can_fuse = self._attached and hasattr(actor, "rollout_available") and actor.rollout_available(vec)
        if fused == "auto":   # the one-launch step wherever it is available
            fused = bool(can_fuse)
        if fused and not can_fuse:
            raise ValueError("FusedRollout(fused=True): fwg_rollout_step is not available for this env / head (needs a specialised "
                             "kernel, the dense observation batch and an attached head)")
        # OPT-IN (fused=True, or FWGYM_ROLLOUT_FUSED=1 for fused=None): tests/test_rollout.py::test_fused_launch_equals_two_launches_on_gpu
        # failed twice in ~25 runs of the whole GPU suite in round 4 and the cause was never found (2 000+ clean iterations since,
        # in isolation and in suite context: profiles/r05_soak.txt) -- until it is, the path whose every launch boundary is a
        # kernel boundary stays the default and the one-launch step is something a caller asks for (bench.py does, and says so)
        import os
        self.fused = (can_fuse and os.environ.get("FWGYM_ROLLOUT_FUSED", "0") == "1") if fused is None else bool(fused)
        self.tap = tap
        if graph:
            self._capture()

    def _prime(self):
        if not self._primed:   # first observation after reset(): no transition led here
            c = self.cur
            o = self.vec._obs if self._attached else self.vec._obs_buf
            self.actor.observe(o)
            self.actor.act(o, norm_obs=c["obs"], action=c["actions"], value=c["values"], logp=c["logp"])
            self._primed = True

    def _step(self, action, nxt, reward_out=None, done_out=None):
        """One env step under `action`, then the head on the new observation writing into the slices of `nxt`."""
        vec, actor = self.vec, self.actor
        o, r, d = vec.step_device(action, want_obs=self._attached)
        if not self._attached:
            o = vec._obs_buf    # the row log itself: the head windows it on the device
            actor.observe(o, r, d)
        actor.act(o, reward=r, done=d, norm_obs=nxt["obs"], action=nxt["actions"], value=nxt["values"], logp=nxt["logp"],
                  norm_reward=reward_out, done_out=done_out)

    def _body(self):
        buf, cur, n = self.buf, self.cur, self.n_steps
        vec, actor = self.vec, self.actor
        tap = self.tap if self._graph is None and not getattr(self, "_capturing", False) else None
        for k in ("obs", "actions", "values", "logp"):
            buf[k][0][...] = cur[k]
        if self.fused:
            # env step 0 under the carried-over actions; then [act t + env step t] in one launch each; then act n
            o, r, d = vec.step_device(buf["actions"][0])
            if tap is not None:
                tap(0, o, r, d)
            for t in range(1, n):
                o, r, d = actor.rollout_step(vec, norm_obs=buf["obs"][t], action=buf["actions"][t], value=buf["values"][t],
                                             logp=buf["logp"][t], norm_reward=buf["rewards"][t - 1], done_out=buf["dones"][t - 1])
                if tap is not None:
                    tap(t, o, r, d)
            actor.act(o, reward=r, done=d, norm_obs=cur["obs"], action=cur["actions"], value=cur["values"], logp=cur["logp"],
                      norm_reward=buf["rewards"][n - 1], done_out=buf["dones"][n - 1])
            return
        for t in range(n):
            nxt = cur if t == n - 1 else {k: buf[k][t + 1] for k in cur}
            self._step(buf["actions"][t], nxt, buf["rewards"][t], buf["dones"][t])
            if tap is not None:
                tap(t, self.vec._obs, self.vec._rew, self.vec._done)

    def _capture(self):
        import torch
        assert self.n_steps % 2 == 0, "graph mode double-buffers step counter and statistics: capture an even number of steps"
        vec = self.vec
        dev = vec._obs.device
        vec.set_graph_mode(True)
        self._prime()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):   # warm-up outside capture: two env steps, every kernel of the body launched once
            if self.fused:
                c = self.cur
                vec.step_device(c["actions"])
                o, r, d = self.actor.rollout_step(vec, norm_obs=c["obs"], action=c["actions"], value=c["values"], logp=c["logp"])
                self.actor.act(o, reward=r, done=d, norm_obs=c["obs"], action=c["actions"], value=c["values"], logp=c["logp"])
            else:
                self._step(self.cur["actions"], self.cur)
                self._step(self.cur["actions"], self.cur)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        self._parity = vec.capture_begin(self.n_steps)
        self._capturing = True
        with torch.cuda.graph(graph):
            self._body()
        self._capturing = False
        vec.capture_end()
        self._graph = graph

    def run(self):
        self._prime()
        if self._graph is not None:
            self.vec.replay_check(self._parity)
            self._graph.replay()
            self.vec.note_replayed_steps(self.n_steps)
        else:
            self._body()
        return self.buf
