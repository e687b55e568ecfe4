"""Drop-in single-environment facade with the reference's public API
(reference gym_fixed_wing/fixed_wing.py: class FixedWingAircraft -- __init__ :14, seed :214, set_curriculum_level :224,
reset :287, step :338, render :572, save_history :654, get_metric :1095).

One instance is a 1-env slice of the batched HIP path (FixedWingVecEnv with auto-reset off), so a script written
against the reference (`env = FixedWingAircraft(config_path); obs = env.reset(); obs, r, done, info = env.step(a)`)
runs unchanged.  Throughput comes from FixedWingVecEnv; this class exists for API parity, evaluation and rendering.
"""
import os

import numpy as np

from . import _native as nat
from .spaces import Env
from .vec_env import FixedWingVecEnv

_RECORDED = ["roll", "pitch", "yaw", "omega_p", "omega_q", "omega_r", "position_n", "position_e", "position_d",
             "velocity_u", "velocity_v", "velocity_w", "Va", "alpha", "beta", "elevator", "aileron", "throttle"]


class _StateView(object):
    """simulator.state[name]: .value / .history (+ the limit attributes) as the reference reads them."""

    def __init__(self, owner, name, var):
        self._owner, self.name, self._var = owner, name, var
        self.history = None

    def __getattr__(self, item):
        return getattr(self._var, item)

    @property
    def value(self):
        return self._owner._values[self.name]


class _SimulatorView(object):
    def __init__(self, env):
        self._env = env
        ec = env._vec.env_config
        self.dt = ec.dt
        self.params = ec.params
        self.plots = []
        self._values = {}
        self.state = {n: _StateView(self, n, v) for n, v in ec.state.items()}

    def _refresh(self, record, reset=False):
        vals = self._env._vec.get_state(_RECORDED)
        self._values = {k: float(v[0]) for k, v in vals.items()}
        for n in _RECORDED:
            st = self.state[n]
            if n in ("elevator", "aileron", "throttle"):
                if reset:
                    st.history = {"value": [], "command": []}
                if record:
                    st.history["value"].append(self._values[n])
            else:
                if reset:
                    st.history = []
                if record:
                    st.history.append(self._values[n])

    def constrained_command(self, action):
        """What PyFly stores in simulator.state[actuator].history["command"] (read by the reference at
        fixed_wing.py:828 and :1110): the inputs after action scaling (fixed_wing.py:349-354,439-459), the value limits of
        elevator / aileron / throttle and of the two elevons, mapped back to elevator / aileron -- the same arithmetic the
        step kernel applies (csrc/fwgym_physics.h constrain_commands)."""
        ec = self._env._vec.env_config
        a = np.asarray(action, dtype=np.float64).reshape(3)
        if self._env.cfg["action"].get("scale_space", False):
            lo, hi = self._env.cfg["action"].get("scale_low", -1), self._env.cfg["action"].get("scale_high", 1)
            to_lo, to_hi = ec.action_scale_to_low, ec.action_scale_to_high
            a = (to_hi - to_lo) * (np.clip(a, lo, hi) - lo) / (hi - lo) + to_lo

        def lim(x, name):
            v = ec.state[name]
            if getattr(v, "value_min", None) is not None:
                x = max(x, v.value_min)
            if getattr(v, "value_max", None) is not None:
                x = min(x, v.value_max)
            return x
        e, al, t = lim(a[0], "elevator"), lim(a[1], "aileron"), lim(a[2], "throttle")
        er, el = lim(e - al, "elevon_right"), lim(e + al, "elevon_left")
        return {"elevator": 0.5 * (er + el), "aileron": 0.5 * (el - er), "throttle": t}

    def record_command(self, action):
        cmd = self.constrained_command(action)
        for n, v in cmd.items():
            self.state[n].history["command"].append(float(v))

    def get_states_vector(self, states, attribute="value"):
        return np.array([getattr(self.state[s], attribute) for s in states])

    def seed(self, seed=None):
        pass


class FixedWingAircraft(Env):
    def __init__(self, config_path=None, sampler=None, sim_config_path=None, sim_parameter_path=None, config_kw=None,
                 sim_config_kw=None, device=0, **vec_kw):
        if sampler is not None:
            raise NotImplementedError("sampler hook (reference fixed_wing.py:273-283) is not supported")
        self._vec = FixedWingVecEnv(config_path, num_envs=1, device=device, sim_config_path=sim_config_path,
                                    sim_parameter_path=sim_parameter_path, config_kw=config_kw,
                                    sim_config_kw=sim_config_kw, auto_reset=False, as_numpy=True, **vec_kw)
        self.cfg = self._vec.cfg
        self.simulator = _SimulatorView(self)
        self.observation_space = self._vec.observation_space
        self.action_space = self._vec.action_space
        self.steps_max = self.cfg["steps_max"]
        self.steps_count = None
        self.history = None
        self.target = None
        self.training = True
        self.render_on_reset = False
        self.render_on_reset_kw = {}
        self.save_on_reset = False
        self.save_on_reset_kw = {}
        self.np_random = np.random.RandomState()
        self.viewer = None
        self.goal_enabled = self._vec.env_config.goal_enabled
        self._last_metrics = None
        self._curriculum_level = 1

    # ------------------------------------------------------------------------------------------------------------------
    def seed(self, seed=None):
        seed = 0 if seed is None else int(seed)
        self.np_random = np.random.RandomState(seed % (2 ** 32))
        self._vec.seed(seed)
        return [seed]

    def set_curriculum_level(self, level):
        self._curriculum_level = level
        self._vec.set_curriculum_level(level)

    def _targets(self):
        tg = self._vec._mem.to_host(self._vec._target)[0]
        return {n: float(tg[k]) for k, n in enumerate(self._vec.target_names)}

    def reset(self, state=None, target=None, **sim_reset_kw):
        if self.render_on_reset:
            self.render(**self.render_on_reset_kw)
            self.render_on_reset, self.render_on_reset_kw = False, {}
        if self.save_on_reset:
            self.save_history(**self.save_on_reset_kw)
            self.save_on_reset, self.save_on_reset_kw = False, {}
        if sim_reset_kw:
            raise NotImplementedError("simulator reset keywords {}".format(list(sim_reset_kw)))
        obs = self._vec.reset(states=state, targets=target)[0]
        self.steps_count = 0
        vec = self._vec
        tg = vec.get_state(["target_" + n for n in vec.target_names])
        self.target = {n: float(tg["target_" + n][0]) for n in vec.target_names}
        self.simulator._refresh(record=True, reset=True)
        self.history = {"action": [], "reward": [], "observation": [obs],
                        "target": {k: [v] for k, v in self.target.items()},
                        "error": {k: [self._error(k)] for k in self.target}}
        if self.goal_enabled:
            self.history["goal"] = {k: [v] for k, v in self._goal_status().items()}
        return np.asarray(obs, dtype=np.float64)

    def _error(self, name):
        val, tgt = self.simulator.state[name].value, self.target[name]
        if self.simulator.state[name].wrap:
            d = (val - tgt + np.pi) % (2 * np.pi) - np.pi
            return d + 2 * np.pi if d < -np.pi else d
        return tgt - val

    def _goal_status(self):
        st = {}
        for t in self._vec.env_config.target_props_init["states"].items():
            name, props = t
            if props.get("bound", None) is not None:
                b = np.radians(props["bound"]) if props.get("convert_to_radians", False) else props["bound"]
                st[name] = bool(np.abs(self._error(name)) <= b)
        st["all"] = all(st.values())
        return st

    def step(self, action):
        action = np.asarray(action, dtype=np.float64)
        assert not np.any(np.isnan(action))
        self.history["action"].append(action)
        prev_target = dict(self.target)
        self.simulator.record_command(action)   # PyFly appends the constrained command before it integrates
        obs, rew, done, infos = self._vec.step(action.reshape(1, 3).astype(np.float32))
        self.steps_count += 1
        info = dict(infos[0])
        done = bool(done[0])
        failed = done and info.get("termination") not in ("steps", "success")
        self.simulator._refresh(record=not failed)
        if not failed:
            if self.goal_enabled:
                cur = self.target
                self.target = prev_target          # goal status is evaluated against the pre-update target
                for k, v in self._goal_status().items():
                    self.history["goal"][k].append(v)
                self.target = cur
            self.target = info["target"]
            for k, v in self.target.items():
                self.history["target"][k].append(v)
                self.history["error"][k].append(self._error(k))
            self.history["observation"].append(obs[0])
            self.history["reward"].append(float(rew[0]))
        if done:
            self._last_metrics = {m["name"]: info.get(m["name"]) for m in self.cfg.get("metrics", [])}
        info["target"] = self.target
        return np.asarray(obs[0], dtype=np.float64), float(rew[0]), done, info

    def get_metric(self, metric, **metric_kw):
        """Episodic metrics (fixed_wing.py:1095-1162) as computed on the device at the end of the episode."""
        if self._last_metrics is None or metric not in self._last_metrics:
            raise RuntimeError("metric {} is available after an episode has finished".format(metric))
        return self._last_metrics[metric]

    # ------------------------------------------------------------------------------------------------------------------
    def render(self, mode="plot", show=True, close=True, block=False, save_path=None):
        """Plots the recorded episode (reference fixed_wing.py:572-652); deferred to the next reset while training."""
        if self.training and not self.render_on_reset:
            self.render_on_reset = True
            self.render_on_reset_kw = {"mode": mode, "show": show, "block": block, "close": close, "save_path": save_path}
            return None
        if mode not in ("plot", "rgb_array"):
            if mode == "animation":
                raise NotImplementedError
            raise ValueError("Unexpected value {} for mode".format(mode))
        import matplotlib
        if not show:
            matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        import matplotlib.gridspec
        rcfg = self.cfg["render"]
        groups = [("roll", "pitch"), ("Va",), ("omega_p", "omega_q", "omega_r"), ("elevator", "aileron", "throttle")]
        extra = int(bool(rcfg["plot_action"])) + int(bool(rcfg["plot_reward"]))
        fig = plt.figure(figsize=(9, 16))
        gs = matplotlib.gridspec.GridSpec(len(groups) + extra, 1)
        for gi, names in enumerate(groups):
            ax = plt.subplot(gs[gi, 0])
            for n in names:
                h = self.simulator.state[n].history
                y = h["value"] if isinstance(h, dict) else h
                line, = ax.plot(range(len(y)), y, label=n)
                if rcfg["plot_target"] and n in self.history["target"]:
                    tgt = np.array(self.history["target"][n])
                    ax.plot(range(len(tgt)), tgt, "--", color=line.get_color(), label=n + " target")
                    if rcfg["plot_goal"] and self.goal_enabled and n in self.history["goal"]:
                        p = self._vec.env_config.target_props_init["states"][n]
                        b = np.radians(p["bound"]) if p.get("convert_to_radians", False) else p["bound"]
                        ok = np.array(self.history["goal"][n], dtype=bool)[:len(tgt)]
                        ax.fill_between(range(len(tgt)), tgt - b, tgt + b, where=ok, alpha=0.2, color=line.get_color())
            ax.legend(loc="upper right")
        if rcfg["plot_action"]:
            ax = plt.subplot(gs[-extra, 0], title="Actions")
            y = np.array(self.history["action"]).reshape(-1, 3)
            for i, a in enumerate(self.cfg["action"]["states"]):
                ax.plot(range(len(y)), y[:, i], label=a["name"])
            ax.legend()
        if rcfg["plot_reward"]:
            ax = plt.subplot(gs[-1, 0], title="Reward")
            ax.plot(range(len(self.history["reward"])), self.history["reward"])
        self.viewer = {"fig": fig, "gs": gs}
        if save_path is not None:
            d = os.path.dirname(save_path)
            if d and not os.path.isdir(d):
                os.makedirs(d)
            ext = os.path.splitext(save_path)[1]
            plt.savefig(save_path, bbox_inches="tight", **({"format": ext[1:]} if ext else {}))
        if mode == "rgb_array":
            return None
        if show:
            plt.show(block=block)
        if close:
            plt.close(fig)
            self.viewer = None
            return None
        return fig

    def save_history(self, path, states, save_targets=True):
        """reference fixed_wing.py:654-672: .npy dict of state histories (+ '<state>_target' entries)."""
        if self.training and not self.save_on_reset:
            self.save_on_reset = True
            self.save_on_reset_kw = {"path": path, "states": states, "save_targets": save_targets}
            return
        res = {}
        for s in ([states] if isinstance(states, str) else states):
            h = self.simulator.state[s].history
            res[s] = list(h["value"] if isinstance(h, dict) else h)
        if save_targets:
            for s in self.target:
                if s in res:
                    res[s + "_target"] = self.history["target"][s]
        np.save(path, res)

    def get_initial_state(self):
        """reference fixed_wing.py:848-862 (test-set record format)."""
        res = {"state": {}, "target": {}}
        for n in _RECORDED:
            h = self.simulator.state[n].history
            res["state"][n] = (h["value"] if isinstance(h, dict) else h)[0]
        for n in ("wind_n", "wind_e", "wind_d"):
            res["state"][n] = float(self._vec.get_state([n])[n][0])
        res["target"] = {s: h[0] for s, h in self.history["target"].items()}
        return res

    def get_simulator_parameters(self, normalize=True):
        """reference fixed_wing.py:872-888: the aircraft parameters sampled for this episode (simulator["model"])."""
        return list(self._vec.get_simulator_parameters(normalize)[0])

    def close(self):
        self._vec.close()
