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
from .spaces import Box, DictSpace, Env, GoalEnv
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


class FixedWingAircraftGoal(FixedWingAircraft, GoalEnv):
    """The reference's goal-conditioned variant (fixed_wing.py:1165-1277, gym.GoalEnv for hindsight replay): dict
    observations {observation, achieved_goal, desired_goal} with the goal states of cfg["observation"]["goals"] scaled by
    (x - mean) / var, get_goal_limits and compute_reward(achieved_goal, desired_goal, info).

    reset / step run on the device like the parent's.  compute_reward re-evaluates the reward function for SUBSTITUTED goal
    states and targets on transitions of the episode held in this object (the reference does it by temporarily overwriting its
    simulator / target / history attributes, :1218-1277); it is the relabelling call of a replay buffer, off the step path, and
    is evaluated here on the host in float64 from the same configuration (the reward of step() itself always comes from the
    kernel)."""

    def __init__(self, config_path=None, sampler=None, sim_config_path=None, sim_parameter_path=None, config_kw=None,
                 sim_config_kw=None, device=0, **vec_kw):
        super().__init__(config_path, sampler=sampler, sim_config_path=sim_config_path, sim_parameter_path=sim_parameter_path,
                         config_kw=config_kw, sim_config_kw=sim_config_kw, device=device, **vec_kw)
        goals = self.cfg["observation"]["goals"]
        self.goal_states = [g["name"] for g in goals]
        self.goal_means = np.array([g["mean"] for g in goals], dtype=np.float64)
        self.goal_vars = np.array([g["var"] for g in goals], dtype=np.float64)
        shape = self.observation_space.shape
        gshape = (len(self.goal_states),) if len(shape) == 1 else (shape[0], len(self.goal_states))
        self.observation_space = DictSpace(dict(desired_goal=Box(-np.inf, np.inf, shape=gshape, dtype="float32"),
                                                achieved_goal=Box(-np.inf, np.inf, shape=gshape, dtype="float32"),
                                                observation=self.observation_space))
        self.prev_shaping = {t["function_class"]: None for t in self.cfg["reward"]["terms"]}

    # -- observations -----------------------------------------------------------------------------------------------------
    def _goal_obs(self, obs):
        ach = (np.array([self.simulator.state[s].value for s in self.goal_states]) - self.goal_means) / self.goal_vars
        des = (np.array([self.target[s] for s in self.goal_states]) - self.goal_means) / self.goal_vars
        length = self.cfg["observation"]["length"]
        if length > 1:
            ach = np.repeat(ach[np.newaxis, :], length, axis=0)
            des = np.repeat(des[np.newaxis, :], length, axis=0)
        return dict(desired_goal=des, achieved_goal=ach, observation=obs)

    def reset(self, state=None, target=None):
        super().reset()          # (the reference resets twice, fixed_wing.py:1197-1199: one episode's draws are consumed)
        obs = super().reset(state, target)
        out = self._goal_obs(obs)
        self.history["observation"] = [out]
        return out

    def step(self, action):
        obs, rew, done, info = super().step(action)
        out = self._goal_obs(obs)
        if self.history["observation"] and self.history["observation"][-1] is obs:
            self.history["observation"][-1] = out
        return out, rew, done, info

    def get_goal_limits(self):
        low, high = [], []
        for i, name in enumerate(self.goal_states):
            g = [t for t in self.cfg["target"]["states"] if t["name"] == name][0]
            lo, hi = g["low"], g["high"]
            if g.get("convert_to_radians", False):
                lo, hi = np.radians(lo), np.radians(hi)
            low.append((lo - self.goal_means[i]) / self.goal_vars[i])
            high.append((hi - self.goal_means[i]) / self.goal_vars[i])
        return np.array(low), np.array(high)

    # -- reward for substituted goals -------------------------------------------------------------------------------------
    def _host_reward(self, values, targets, action, actions, steps_count, prev_shaping, potential):
        """get_reward (fixed_wing.py:674-774) with success = False, for the given state values / targets / raw-action history.
        Returns (reward, shaping values per function class)."""
        cfg = self.cfg

        def value(name):
            return values[name] if name in values else self.simulator.state[name].value

        def error(name):
            if self.simulator.state[name].wrap:
                d = (value(name) - targets[name] + np.pi) % (2 * np.pi) - np.pi
                return d + 2 * np.pi if d < -np.pi else d
            return targets[name] - value(name)

        def goal_status():
            st = {}
            for name, props in self._vec.env_config.target_props_init["states"].items():
                if props.get("bound", None) is not None:
                    b = np.radians(props["bound"]) if props.get("convert_to_radians", False) else props["bound"]
                    st[name] = bool(np.abs(error(name)) <= b)
            st["all"] = all(st.values())
            return st

        ec = self._vec.env_config
        terms = {t["function_class"]: {"val": 0.0, "weight": t["weight"], "val_shaping": 0.0} for t in cfg["reward"]["terms"]}
        for f in cfg["reward"]["factors"]:
            cls = f["class"]
            if cls == "action":
                if f["type"] == "value":
                    val = np.sum(np.abs(actions[-1]))
                elif f["type"] == "delta":
                    val = np.sum(np.abs(np.diff(actions[-f["window_size"]:], axis=0))) if steps_count > 1 else 0
                elif f["type"] == "bound":
                    hi, lo = ec.action_bounds_max, ec.action_bounds_min
                    val = 0 if action is None else float(np.sum(np.where(action > hi, action - hi, 0)) + np.sum(np.where(action < lo, lo - action, 0)))
                else:
                    raise ValueError("Unexpected type {} for reward class action".format(f["type"]))
            elif cls == "state":
                if f["type"] == "value":
                    val = value(f["name"])
                elif f["type"] == "error":
                    val = error(f["name"])
                else:
                    raise NotImplementedError("compute_reward with reward type {}".format(f["type"]))
            elif cls == "success":
                val = 0
            elif cls == "step":
                val = f["value"]
            elif cls == "goal":
                st = goal_status()
                if f["type"] == "per_state":
                    val = sum(f["value"] / len(targets) for k, ok in st.items() if k != "all" and ok)
                else:
                    val = f["value"] if st["all"] else 0
            else:
                raise ValueError("Unexpected reward component type {}".format(cls))
            if f["function_class"] == "linear":
                val = np.clip(np.abs(val) / f["scaling"], 0, f.get("max", None))
            else:
                val = val ** 2 / f["scaling"]
            terms[f["function_class"]]["val_shaping" if f.get("shaping", False) else "val"] += val * np.sign(f.get("sign", -1))
        reward, shaping = 0.0, {}
        for fc, t in terms.items():
            prev = prev_shaping.get(fc, None) if prev_shaping is not None else None
            if fc == "exponential":
                arg = t["val"] + ((t["val_shaping"] - prev) if prev is not None else 0.0) if potential else t["val"] + t["val_shaping"]
                val = -1 + np.exp(arg)
            else:
                val = t["val"]
                if potential:
                    if prev is not None:
                        val += t["val_shaping"] - prev
                else:
                    val += t["val_shaping"]
            shaping[fc] = t["val_shaping"]
            reward += t["weight"] * val
        return float(reward), shaping

    def compute_reward(self, achieved_goal, desired_goal, info):
        """fixed_wing.py:1218-1277.  info: {"step": index of the transition in this episode, "action": its raw action,
        "prev_state": the goal states before the transition (potential rewards)}."""
        achieved = np.asarray(achieved_goal, dtype=np.float64) * self.goal_vars + self.goal_means
        desired = np.asarray(desired_goal, dtype=np.float64) * self.goal_vars + self.goal_means
        action = np.asarray(info.get("action", np.zeros(len(self.cfg["action"]["states"]))), dtype=np.float64)
        actions = list(self.history["action"][:info["step"]]) + [action]
        steps_count = info["step"]
        targets = dict(self.target)
        for i, s in enumerate(self.goal_states):
            targets[s] = desired[i]
        potential = self.cfg["reward"]["form"] == "potential"
        prev_shaping = None
        if potential and info["step"] > 0:   # shaping value of the state before the transition, with the action before it
            prev_action = actions[-2] if len(actions) >= 2 else None
            _, prev_shaping = self._host_reward({s: info["prev_state"][i] for i, s in enumerate(self.goal_states)}, targets,
                                                prev_action, actions, steps_count, None, potential)
        reward, _ = self._host_reward({s: achieved[i] for i, s in enumerate(self.goal_states)}, targets, action, actions,
                                      steps_count, prev_shaping, potential)
        return reward
