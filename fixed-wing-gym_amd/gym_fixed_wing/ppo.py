"""PPO learner on top of the device-resident rollout (second half): what the reference's training
script does with `PPO2(MlpPolicy, VecNormalize(SubprocVecEnv(...))).learn(total_timesteps, callback=monitor_training)`
(examples/train_rl_controller.py:223-232; curriculum callback :80-87), with every env-side piece on the MI355X:

  rollout     FusedRollout: env step + VecNormalize + 64-64 MlpPolicy + sampling as HIP kernels (one or two launches per
              step), writing obs / actions / values / log-probs / normalised rewards / dones step-major in place;
  advantages  fwg_gae (HIP, one launch over the rollout buffers, 17 B per transition);
  update      the clipped-surrogate objective of stable-baselines' PPO2 with its defaults (gamma 0.99, lambda 0.95, clip 0.2 on
              policy AND value, entropy 0.01, value 0.5, lr 2.5e-4 Adam eps 1e-5, 4 epochs x 4 minibatches, gradient norm 0.5) --
              torch autograd on the 12-64-64 networks (the model side is not the hot path);
  curriculum  distributed.gather_success (one RCCL all-gather of 64 B per rank) + CurriculumSchedule after every rollout.

The torch policy is the master copy of the weights; after every update they are loaded into the HIP head
(DeviceActor.load_policy).  Multi-GPU: every rank collects its shard; gradients are averaged with one all-reduce per
minibatch step when torch.distributed is initialised (data-parallel PPO)."""
import math

import numpy as np
import torch
from torch import nn

from . import _native as nat
from .rollout import FusedRollout, MlpPolicy

# stable-baselines PPO2.__init__ defaults (the reference passes none: `PPO2(policy, env, verbose=1, tensorboard_log=...)`)
PPO2_DEFAULTS = dict(gamma=0.99, n_steps=128, ent_coef=0.01, learning_rate=2.5e-4, vf_coef=0.5, max_grad_norm=0.5, lam=0.95,
                     nminibatches=4, noptepochs=4, cliprange=0.2)


STAT_KEYS = ("pg_loss", "vf_loss", "entropy", "approx_kl", "clip_frac")


def sb_init_(policy):
    """stable-baselines' MlpPolicy initialisation: orthogonal, gain sqrt(2) on the hidden layers, 0.01 on the action mean,
    1 on the value output; zero biases; log-std 0 (common/policies.py mlp_extractor / linear(init_scale))."""
    for net, out_gain in ((policy.pi, 0.01), (policy.vf, 1.0)):
        lin = [m for m in net if isinstance(m, nn.Linear)]
        for i, m in enumerate(lin):
            nn.init.orthogonal_(m.weight, gain=out_gain if i == len(lin) - 1 else math.sqrt(2.0))
            nn.init.zeros_(m.bias)
    with torch.no_grad():
        policy.log_std.zero_()
    return policy


def gae(lib, mem, rewards, values, dones, last_value, gamma, lam, adv_out=None, ret_out=None):
    """fwg_gae on buffers of the env's memory backend ([T, N] step-major; device tensors on the GPU)."""
    import ctypes
    T, N = int(rewards.shape[0]), int(rewards.shape[1])
    adv_out = mem.zeros((T, N)) if adv_out is None else adv_out
    ret_out = mem.zeros((T, N)) if ret_out is None else ret_out
    nat.check(lib, lib.fwg_gae(T, N, mem.ptr(rewards), mem.ptr(values), mem.ptr(dones), mem.ptr(last_value),
                               ctypes.c_float(gamma), ctypes.c_float(lam), mem.ptr(adv_out), mem.ptr(ret_out), mem.stream()))
    return adv_out, ret_out


def ppo_loss(policy, obs, actions, old_values, old_logp, adv, returns, cliprange, ent_coef, vf_coef):
    """PPO2's loss on one minibatch (stable-baselines ppo2.py setup_model): advantages normalised per minibatch, clipped
    surrogate, value loss clipped around the old value with the SAME range, Gaussian entropy bonus."""
    adv = (adv - adv.mean()) / (adv.std(unbiased=False) + 1e-8)
    mean = policy.pi(obs)
    value = policy.vf(obs).squeeze(-1)
    log_std = policy.log_std
    neglogp = 0.5 * (((actions - mean) / log_std.exp()) ** 2).sum(dim=-1) + 0.5 * math.log(2.0 * math.pi) * actions.shape[-1] + log_std.sum()
    ratio = torch.exp(-old_logp - neglogp)
    pg_loss = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - cliprange, 1.0 + cliprange)).mean()
    v_clipped = old_values + torch.clamp(value - old_values, -cliprange, cliprange)
    vf_loss = 0.5 * torch.max((value - returns) ** 2, (v_clipped - returns) ** 2).mean()
    entropy = (log_std + 0.5 * math.log(2.0 * math.pi * math.e)).sum()
    loss = pg_loss - ent_coef * entropy + vf_coef * vf_loss
    with torch.no_grad():
        stats = {"pg_loss": pg_loss.detach(), "vf_loss": vf_loss.detach(), "entropy": entropy.detach(),
                 "approx_kl": 0.5 * ((neglogp + old_logp) ** 2).mean(), "clip_frac": ((ratio - 1.0).abs() > cliprange).float().mean()}
    return loss, stats


class PPO(object):
    """PPO2-style learner for a FixedWingVecEnv with vector observations (MlpPolicy).  `learn(total_timesteps)` alternates
    rollouts of n_steps x num_envs transitions with noptepochs x nminibatches gradient steps; `callback(self, info)` runs after
    every update (the reference's monitor_training)."""

    def __init__(self, vec, policy=None, seed=0, fused=None, graph=True, curriculum=None, group=None, precise=True,
                 graph_update=True, **kw):
        from .actor import DeviceActor
        hp = dict(PPO2_DEFAULTS)
        unknown = set(kw) - set(hp)
        if unknown:
            raise TypeError("unknown PPO hyper-parameters: {}".format(sorted(unknown)))
        hp.update(kw)
        self.hp, self.vec, self.group = hp, vec, group
        self.n_steps = int(hp["n_steps"])
        self._torch_dev = getattr(vec._mem, "device", torch.device("cpu"))
        torch.manual_seed(seed)
        self.policy = (sb_init_(MlpPolicy(vec.obs_dim)) if policy is None else policy).to(self._torch_dev)
        self.actor = DeviceActor.for_env(vec, seed=seed, gamma=hp["gamma"], precise=precise)
        self.actor.load_policy(self.policy)
        self._rollout_kw = dict(graph=bool(graph) and self._torch_dev.type == "cuda", fused=fused)
        self.rollout = FusedRollout(vec, self.actor, self.n_steps, **self._rollout_kw)
        self._spec_at_capture = vec.spec_index
        on_gpu = self._torch_dev.type == "cuda"
        # (GPU: the fused, capturable form -- ONE launch for all thirteen parameter tensors inside the captured minibatch step)
        self.opt = torch.optim.Adam(self.policy.parameters(), lr=hp["learning_rate"], eps=1e-5, **({"capturable": True, "fused": True} if on_gpu else {}))
        self._graph_update, self._step_graph = bool(graph_update) and on_gpu, None
        self.curriculum = curriculum
        m, N, T = vec._mem, vec.num_envs, self.n_steps
        self.adv, self.ret = m.zeros((T, N)), m.zeros((T, N))
        self.num_timesteps, self.updates = 0, 0
        self.history = []
        self._gen = torch.Generator(device=self._torch_dev)
        self._gen.manual_seed(seed + 1)
        self._world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self._world = torch.distributed.get_world_size(group)
            for p in self.policy.parameters():   # identical initial weights on every rank
                torch.distributed.broadcast(p.data, src=0, group=group)
            self.actor.load_policy(self.policy)

    def _t(self, x):
        return torch.as_tensor(x) if not isinstance(x, torch.Tensor) else x

    def collect(self):
        """One rollout + advantages.  Returns the flattened training batch (views of the rollout buffers, no copies)."""
        buf = self.rollout.run()
        gae(self.vec._lib, self.vec._mem, buf["rewards"], buf["values"], buf["dones"], self.rollout.last_value,
            self.hp["gamma"], self.hp["lam"], self.adv, self.ret)
        T, N = self.n_steps, self.vec.num_envs
        self.num_timesteps += T * N * self._world
        flat = lambda x, *s: self._t(x).reshape((T * N,) + s)
        return {"obs": flat(buf["obs"], self.vec.obs_dim), "actions": flat(buf["actions"], 3), "values": flat(buf["values"]),
                "logp": flat(buf["logp"]), "adv": flat(self.adv), "returns": flat(self.ret)}

    def _minibatch_step(self, mbatch, cliprange):
        hp = self.hp
        loss, stats = ppo_loss(self.policy, mbatch["obs"], mbatch["actions"], mbatch["values"], mbatch["logp"], mbatch["adv"],
                               mbatch["returns"], cliprange, hp["ent_coef"], hp["vf_coef"])
        self.opt.zero_grad(set_to_none=False)
        loss.backward()
        if self._world > 1:   # data-parallel PPO: one all-reduce of the (tiny) gradient per minibatch step
            flat = torch.cat([p.grad.reshape(-1) for p in self.policy.parameters()])
            torch.distributed.all_reduce(flat, group=self.group)
            flat /= self._world
            o = 0
            for p in self.policy.parameters():
                p.grad.copy_(flat[o:o + p.numel()].view_as(p))
                o += p.numel()
        nn.utils.clip_grad_norm_(self.policy.parameters(), hp["max_grad_norm"])
        self.opt.step()
        return torch.stack([stats[k] for k in STAT_KEYS])

    def _capture_step(self, batch, mb, cliprange):
        """One minibatch step -- gather by index, loss, backward, gradient clipping, Adam -- as ONE hipGraph: the 12-64-64
        networks make it ~60 launches of microseconds each, and issued from Python they cost 1.5 ms per step (99 % of a training
        run at 4 096 envs was the optimiser waiting for its own launches)."""
        dev = self._torch_dev
        g = {"idx": torch.zeros(mb, dtype=torch.long, device=dev), "acc": torch.zeros(len(STAT_KEYS), device=dev), "src": batch,
             "key": (tuple(int(batch[k].data_ptr()) for k in sorted(batch)), mb, float(cliprange))}
        snap_p = [p.detach().clone() for p in self.policy.parameters()]
        snap_o = [{k: v.clone() for k, v in st.items() if isinstance(v, torch.Tensor)} for st in self.opt.state.values()]

        # ONE gather per step: the six per-transition arrays side by side in a [n, D + 7] buffer, refreshed once per update
        D = batch["obs"].shape[1]
        g["packed"] = torch.empty((batch["obs"].shape[0], D + 7), device=dev)
        g["pack"] = lambda: torch.cat([batch["obs"], batch["actions"], batch["values"][:, None], batch["logp"][:, None],
                                       batch["adv"][:, None], batch["returns"][:, None]], dim=1, out=g["packed"])
        g["pack"]()

        def body():
            m = g["packed"].index_select(0, g["idx"])
            mbatch = {"obs": m[:, :D], "actions": m[:, D:D + 3], "values": m[:, D + 3], "logp": m[:, D + 4], "adv": m[:, D + 5],
                      "returns": m[:, D + 6]}
            g["acc"].add_(self._minibatch_step(mbatch, cliprange))

        g["idx"].copy_(torch.arange(mb, device=dev))
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        had_state = len(snap_o) > 0
        with torch.no_grad():   # the warm-up steps were real optimiser steps: put weights and moments back IN PLACE
            for p, q in zip(self.policy.parameters(), snap_p):
                p.copy_(q)
            for i, st in enumerate(self.opt.state.values()):
                for k, v in st.items():
                    if isinstance(v, torch.Tensor):
                        v.copy_(snap_o[i][k]) if had_state else v.zero_()
        g["graph"] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g["graph"]):
            body()
        g["acc"].zero_()
        return g

    def update(self, batch, lr=None, cliprange=None):
        hp = self.hp
        n = batch["obs"].shape[0]
        mb = n // int(hp["nminibatches"])
        cliprange = hp["cliprange"] if cliprange is None else cliprange
        if lr is not None:
            for g in self.opt.param_groups:
                g["lr"] = lr
        graphed = self._graph_update and self._world == 1 and lr is None
        if graphed:
            key = (tuple(int(batch[k].data_ptr()) for k in sorted(batch)), mb, float(cliprange))
            if self._step_graph is None or self._step_graph["key"] != key:
                self._step_graph = self._capture_step(batch, mb, cliprange)
            sg = self._step_graph
            sg["pack"]()
            sg["acc"].zero_()
        acc = torch.zeros(len(STAT_KEYS), device=self._torch_dev)
        for _ in range(int(hp["noptepochs"])):
            perm = torch.randperm(n, device=self._torch_dev, generator=self._gen)
            for k in range(int(hp["nminibatches"])):
                idx = perm[k * mb:(k + 1) * mb]
                if graphed:
                    sg["idx"].copy_(idx)
                    sg["graph"].replay()
                else:
                    acc += self._minibatch_step({key_: v[idx] for key_, v in batch.items()}, cliprange)
        if graphed:
            acc = sg["acc"].clone()
        steps = int(hp["noptepochs"]) * int(hp["nminibatches"])
        self.actor.load_policy(self.policy)
        self.updates += 1
        return {k: float(v) / steps for k, v in zip(STAT_KEYS, acc.tolist())}

    def learn(self, total_timesteps, callback=None, log=None):
        from . import distributed as fd
        while self.num_timesteps < total_timesteps:
            batch = self.collect()
            stats = self.update(batch)
            summary = fd.gather_success(self.vec, self.group)    # episodes finished during this rollout, all ranks
            info = dict(stats, timesteps=self.num_timesteps, update=self.updates, episodes=summary["episodes"],
                        level=None if self.curriculum is None else self.curriculum.level)
            if summary["episodes"] > 0:
                info["success"] = dict(summary["success"])
                info["control_variation"] = summary["control_variation"]["all"]
            if self.curriculum is not None:
                info["level"] = self.curriculum.update(self.vec, summary)
                if self._rollout_kw["graph"] and self.vec.spec_index != self._spec_at_capture:
                    # (a captured rollout holds a kernel INSTANCE; the curriculum's ranges are read from memory and do not move it,
                    # but a configuration update that changes a folded value would: capture anew, fwg_replay_check refuses otherwise)
                    self.rollout = FusedRollout(self.vec, self.actor, self.n_steps, **self._rollout_kw)
                    self._spec_at_capture = self.vec.spec_index
            self.history.append(info)
            if log is not None:
                log(info)
            if callback is not None and callback(self, info) is False:
                break
        return self

    def deterministic_policy(self):
        """obs [N, ...] (raw, device tensor) -> mean action: VecNormalize with the statistics FROZEN at this moment (clip 10) in
        front of the policy network -- what `model.predict(obs, deterministic=True)` on a VecNormalize(training=False) env
        computes in the reference's evaluation (examples/evaluate_controller.py:93-100, :139)."""
        st = self.actor.get_stats()
        dev = self._torch_dev
        mean = torch.as_tensor(st["obs_mean"], dtype=torch.float32, device=dev)
        std = torch.sqrt(torch.as_tensor(st["obs_var"], dtype=torch.float32, device=dev) + 1e-8)
        pi = self.policy.pi

        @torch.no_grad()
        def act(obs):
            x = torch.as_tensor(obs, dtype=torch.float32, device=dev)
            return pi(((x.reshape(x.shape[0], -1) - mean) / std).clamp(-10.0, 10.0))
        return act

    def save(self, path):
        """Weights + VecNormalize statistics (the reference's save_model: model.pkl + save_running_average)."""
        st = self.actor.get_stats()
        sd = {k: v.detach().cpu().numpy() for k, v in self.policy.state_dict().items()}
        np.savez(path, **sd, **{"stat_" + k: np.asarray(v) for k, v in st.items()})

    def load(self, path):
        z = np.load(path)
        self.policy.load_state_dict({k: torch.as_tensor(z[k]) for k in z.files if not k.startswith("stat_")})
        self.actor.load_policy(self.policy)
        self.actor.set_stats(z["stat_obs_mean"], z["stat_obs_var"], float(z["stat_obs_count"]), float(z["stat_ret_mean"]),
                             float(z["stat_ret_var"]), float(z["stat_ret_count"]))
        return self
