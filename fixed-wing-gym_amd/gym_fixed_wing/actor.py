"""The rollout head as HIP kernels: VecNormalize (running observation / return normalisation) + the stable-baselines
MlpPolicy (pi, vf: obs -> 64 tanh -> 64 tanh -> out) + action sampling, on device-resident batches.

Replaces what the reference's loops do between two env steps -- `VecNormalize(SubprocVecEnv(...))` around the env and
`model.predict` / `PPO2.step` on the normalised observation (examples/train_rl_controller.py:223-231,
examples/evaluate_controller.py:93-100) -- with fwg_actor_observe / fwg_actor_act of libfwgym (include/fwgym.h "Rollout
head"): one rollout step is three launches (env step, batch moments, statistics update + normalise + MLP on the matrix
cores + sample).  No CPU path: without the HIP library the constructor raises."""
import ctypes

import numpy as np

from . import _native as nat

_KEYS = ["pi_w0", "pi_b0", "pi_w1", "pi_b1", "pi_w2", "pi_b2", "vf_w0", "vf_b0", "vf_w1", "vf_b1", "vf_w2", "vf_b2", "log_std"]


def weights_from_module(policy):
    """torch MlpPolicy of gym_fixed_wing.rollout (pi / vf nn.Sequential + log_std) -> dict of float32 arrays."""
    out = {}
    for net in ("pi", "vf"):
        seq = getattr(policy, net)
        lin = [m for m in seq if hasattr(m, "weight")]
        assert len(lin) == 3, "MlpPolicy layout: three Linear layers per network"
        for i, m in enumerate(lin):
            out["{}_w{}".format(net, i)] = m.weight.detach().cpu().numpy().astype(np.float32)
            out["{}_b{}".format(net, i)] = m.bias.detach().cpu().numpy().astype(np.float32)
    out["log_std"] = policy.log_std.detach().cpu().numpy().astype(np.float32)
    return out


def weights_from_stable_baselines(params):
    """Parameter dict of a stable-baselines (TF1) MlpPolicy checkpoint -- kernels are [in][out] there (the layout of the
    shipped examples/models/mlp_controller, tests/golden/mlp_controller.json) -> torch layout [out][in]."""
    g = lambda k: np.asarray(params[k], dtype=np.float32)
    out = {}
    if "vf_fc0_w" not in params:   # policy-only export (evaluation): a zero value network
        params = dict(params)
        d = np.asarray(params["pi_fc0_w"]).shape[0]
        params.update({"vf_fc0_w": np.zeros((d, 64)), "vf_fc0_b": np.zeros(64), "vf_fc1_w": np.zeros((64, 64)),
                       "vf_fc1_b": np.zeros(64), "vf_w": np.zeros((64, 1)), "vf_b": np.zeros(1)})
    for net, last in (("pi", "pi"), ("vf", "vf")):
        out[net + "_w0"] = g(net + "_fc0_w").T.copy(); out[net + "_b0"] = g(net + "_fc0_b")
        out[net + "_w1"] = g(net + "_fc1_w").T.copy(); out[net + "_b1"] = g(net + "_fc1_b")
        out[net + "_w2"] = g(last + "_w").T.copy(); out[net + "_b2"] = g(last + "_b")
    out["log_std"] = g("logstd").reshape(-1) if "logstd" in params else np.zeros(out["pi_w2"].shape[0], np.float32)
    return out


class DeviceActor(object):
    def __init__(self, num_envs, obs_dim, act_dim=3, gamma=0.99, clip_obs=10.0, clip_reward=10.0, epsilon=1e-8, seed=0,
                 env_id_base=0, device=0, training=True, precise=True, _backend=None, _lib=None):
        self._lib = _lib if _lib is not None else nat.load_library()
        if _backend is None:
            from .vec_env import _TorchBackend
            _backend = _TorchBackend(device)
        self._mem = _backend
        self.num_envs, self.obs_dim, self.act_dim = int(num_envs), int(obs_dim), int(act_dim)
        h = ctypes.c_void_p()
        nat.check(self._lib, self._lib.fwg_actor_create(self._mem.index, self.num_envs, self.obs_dim, self.act_dim, gamma,
                                                        clip_obs, clip_reward, epsilon, ctypes.byref(h)))
        self._handle = h
        self.training, self.precise = bool(training), bool(precise)
        self._configure()
        self.seed(seed, env_id_base)
        self._observed = False     # batch moments of the current observation already accumulated?

    @classmethod
    def for_env(cls, vec, **kw):
        kw.setdefault("env_id_base", getattr(vec, "env_id_base", 0))
        return cls(vec.num_envs, vec.obs_dim, _backend=vec._mem, _lib=vec._lib, **kw)

    def close(self):
        if getattr(self, "_handle", None):
            try:
                self.detach()
                if getattr(self, "_log_env", None) is not None:
                    self._lib.fwg_actor_set_obs_log(self._handle, ctypes.c_void_p())
            except Exception:   # interpreter shutdown: module globals may already be gone
                pass
            self._lib.fwg_actor_destroy(self._handle)
            self._handle = None

    __del__ = close

    def _configure(self):
        nat.check(self._lib, self._lib.fwg_actor_configure(self._handle, int(self.training), int(self.precise)))

    def set_training(self, training):
        self.training = bool(training)
        self._configure()

    def seed(self, seed, env_id_base=0):
        nat.check(self._lib, self._lib.fwg_actor_seed(self._handle, int(seed) & (2 ** 64 - 1), int(env_id_base)))

    def load_policy(self, weights):
        """`weights`: the torch MlpPolicy module, or a dict with the keys pi_w0 .. vf_b2, log_std (torch Linear layout)."""
        w = weights if isinstance(weights, dict) else weights_from_module(weights)
        keep, st = [], nat.ActorWeights()
        shapes = {"pi_w0": (64, self.obs_dim), "pi_w1": (64, 64), "pi_w2": (self.act_dim, 64), "vf_w0": (64, self.obs_dim),
                  "vf_w1": (64, 64), "vf_w2": (1, 64), "pi_b0": (64,), "pi_b1": (64,), "pi_b2": (self.act_dim,),
                  "vf_b0": (64,), "vf_b1": (64,), "vf_b2": (1,), "log_std": (self.act_dim,)}
        for k in _KEYS:
            a = np.ascontiguousarray(np.asarray(w[k], dtype=np.float32))
            if a.shape != shapes[k]:
                raise ValueError("{}: shape {} but the 64-64 MlpPolicy needs {}".format(k, a.shape, shapes[k]))
            keep.append(a)
            setattr(st, k, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        nat.check(self._lib, self._lib.fwg_actor_set_weights(self._handle, ctypes.byref(st)))

    def set_stats(self, obs_mean, obs_var, obs_count=1e-4, ret_mean=0.0, ret_var=1.0, ret_count=1e-4):
        s = nat.ActorStats()
        m, v = np.zeros(64, np.float32), np.ones(64, np.float32)
        m[:self.obs_dim] = np.asarray(obs_mean, np.float32).reshape(-1)
        v[:self.obs_dim] = np.asarray(obs_var, np.float32).reshape(-1)
        s.obs_mean[:] = m.tolist(); s.obs_var[:] = v.tolist()
        s.obs_count, s.ret_mean, s.ret_var, s.ret_count = float(obs_count), float(ret_mean), float(ret_var), float(ret_count)
        nat.check(self._lib, self._lib.fwg_actor_set_stats(self._handle, ctypes.byref(s), self._mem.stream()))

    def get_stats(self):
        s = nat.ActorStats()
        nat.check(self._lib, self._lib.fwg_actor_get_stats(self._handle, ctypes.byref(s), self._mem.stream()))
        return {"obs_mean": np.array(s.obs_mean[:self.obs_dim], np.float32), "obs_var": np.array(s.obs_var[:self.obs_dim], np.float32),
                "obs_count": s.obs_count, "ret_mean": s.ret_mean, "ret_var": s.ret_var, "ret_count": s.ret_count}

    def attach(self, vec):
        """From now on vec.step()/step_device() does the observe() bookkeeping inside the env step kernel."""
        nat.check(self._lib, self._lib.fwg_attach_observer(vec._handle, self._handle))
        self._attached = vec

    def detach(self):
        vec = getattr(self, "_attached", None)
        if vec is not None and getattr(vec, "_handle", None):
            nat.check(self._lib, self._lib.fwg_attach_observer(vec._handle, ctypes.c_void_p()))
        self._attached = None

    def set_obs_log(self, vec):
        """Row-log observations: from now on the `obs` given to observe()/act() is `vec`'s observation row log
        (vec._obs_buf) and the head reads the current window out of it in place -- no dense copy, and safe under hipGraph
        replay because the window position is read on the device (fwg_actor_set_obs_log).  None: dense batches again."""
        nat.check(self._lib, self._lib.fwg_actor_set_obs_log(self._handle, vec._handle if vec is not None else ctypes.c_void_p()))
        self._log_env = vec

    def _p(self, t):
        return ctypes.c_void_p() if t is None else self._mem.ptr(t)

    @staticmethod
    def _dense(t):
        """Inputs must be dense: a row-log observation window (strided view) is materialised here."""
        if t is None:
            return None
        if hasattr(t, "is_contiguous"):
            return t if t.is_contiguous() else t.contiguous()
        return np.ascontiguousarray(t)

    def rollout_available(self, vec):
        """True when head + env step of `vec` can run as ONE launch (fwg_rollout_step): this head attached to it, a specialised
        kernel, the dense observation batch."""
        return bool(self._lib.fwg_rollout_available(vec._handle, self._handle))

    def rollout_step(self, vec, norm_obs=None, action=None, value=None, logp=None, norm_reward=None, done_out=None,
                     deterministic=False):
        """act() on the observation `vec` currently shows (and the reward / done flags of the step that led to it), then
        vec.step_device() under the sampled actions -- one launch (fwg_rollout_step; the body of the reference's training loop,
        examples/train_rl_controller.py:223-231).  Outputs as act(); afterwards vec._obs / _rew / _done hold the new step."""
        m = self._mem
        nat.check(self._lib, self._lib.fwg_rollout_step(
            vec._handle, self._handle, self._p(norm_obs), self._p(action), self._p(value), self._p(logp), self._p(norm_reward),
            self._p(done_out), m.ptr(vec._obs_buf), m.ptr(vec._rew), m.ptr(vec._done), m.ptr(vec._term), m.ptr(vec._term_obs),
            m.ptr(vec._metrics), int(bool(deterministic)), m.stream()))
        self._observed = False
        return vec._obs, vec._rew, vec._done

    def observe(self, obs, reward=None, done=None):
        """VecNormalize.step_wait bookkeeping for a new batch: moments of `obs` ([N, obs_dim] or [N, L, n]); with
        `reward`/`done` also the discounted returns.  The statistics change at the next act()."""
        obs, reward, done = self._dense(obs), self._dense(reward), self._dense(done)
        nat.check(self._lib, self._lib.fwg_actor_observe(self._handle, self._p(obs), self._p(reward), self._p(done),
                                                         self._mem.stream()))
        self._observed = True

    def act(self, obs, reward=None, done=None, norm_obs=None, action=None, value=None, logp=None, norm_reward=None,
            done_out=None, deterministic=False):
        """Normalised observation, action, value, log-probability for `obs`; `norm_reward` receives the normalised
        `reward` of the transition that led to `obs`.  Output arrays are allocated when not given."""
        m, N = self._mem, self.num_envs
        obs, reward, done = self._dense(obs), self._dense(reward), self._dense(done)
        norm_obs = m.zeros((N, self.obs_dim)) if norm_obs is None else norm_obs
        action = m.zeros((N, self.act_dim)) if action is None else action
        value = m.zeros((N,)) if value is None else value
        logp = m.zeros((N,)) if logp is None else logp
        if reward is not None and norm_reward is None:
            norm_reward = m.zeros((N,))
        nat.check(self._lib, self._lib.fwg_actor_act(self._handle, self._p(obs), self._p(reward), self._p(done),
                                                     self._p(norm_obs), self._p(action), self._p(value), self._p(logp),
                                                     self._p(norm_reward), self._p(done_out), int(bool(deterministic)),
                                                     m.stream()))
        self._observed = False
        return norm_obs, action, value, logp, norm_reward
