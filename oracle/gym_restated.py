"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- float64 restatement of the gym-side half of the hot path.

Follows reference gym_fixed_wing/fixed_wing.py: __init__ :14-212, set_curriculum_level :224-285, reset :287-336,
step :338-437, linear_action_scaling :439-459, sample_target :461-521, get_reward :674-774, get_observation :776-846,
_get_error/_get_angle_dist :890-914, _get_goal_status :916-931, _get_next_target :933-991, get_metric :1095-1162.

It is written in the STREAMING formulation the HIP kernels use (bounded windows / running accumulators instead of the
reference's unbounded history lists), so that pinning it against the verbatim reference
(tests/test_oracle_vs_reference.py, build container only, and the committed vectors under tests/golden/) also
validates that reformulation.  The simulator underneath is oracle/pyfly_restated.PyFly (PARITY UNPINNED vs real PyFly).

Supported since rounds 3-4 (pinned against the verbatim reference like the rest): integrator observations and the int_error
reward factor (:708-711, :804-810, incl. the reset observation that reads the previous episode's history, :317-321), sampling of
the simulator keys `turbulence` / `turbulence_intensity` (:560-569), simulator["model"] (:532-559), reward.randomize_scaling
(:330-334).  FixedWingAircraftGoal (:1165-1277) is a host class of the product (gym_fixed_wing/fixed_wing.py) checked against
vectors recorded from the verbatim reference class (tests/golden/g4_goal_*.json), not restated here.
Unsupported (raise NotImplementedError, same as the product): target class attitude_angular (:474-478), the sampler hook
(:273-283), sampling of simulator keys other than states / model / the two turbulence keys.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import copy
import json
import math

import numpy as np

from .pyfly_restated import PyFly

F32MAX = float(np.finfo(np.float32).max)
METRIC_STATES = None


def _override(parent, kws):
    # config_kw semantics of fixed_wing.py:24-29 (dicts recurse; int keys address list items)
    for k, v in kws.items():
        if isinstance(v, dict) or isinstance(parent[k], list):
            _override(parent[k], v)
        else:
            parent[k] = v


class MTStream:
    """Draw source reproducing the reference's consumption order from a numpy RandomState."""

    def __init__(self, rs):
        self.rs = rs

    def target_uniform(self, k, low, high):
        return self.rs.uniform(low, high)

    def uniform(self, low=0.0, high=1.0):
        return self.rs.uniform(low, high)

    def init_noise(self, row):
        return self.rs.uniform(-1, 1)

    def obs_normal(self, idx, mean, std):
        return self.rs.normal(loc=mean, scale=std)

    def model_normal(self, i, loc, scale):
        return self.rs.normal(loc=loc, scale=scale)

    def reward_scale_uniform(self, i, low, high):
        return self.rs.uniform(low, high)

    def model_uniform(self, i, low, high):
        return self.rs.uniform(low=low, high=high)

    # simulator.<key> sampling (fixed_wing.py:560-569)
    def sim_choice(self, i, values, probs):
        return self.rs.choice(values, p=probs)

    def sim_uniform(self, i, low, high):
        return self.rs.uniform(low, high)

    def begin_obs(self, steps_count):
        pass

    def begin_target(self):
        pass

    def begin_episode(self, episode):
        pass


class PhiloxStream:
    """Draw source mirroring the DEVICE's counter-based streams (csrc/fwgym_env.h: reset_env, sample_targets,
    fix_padded_rows, add_obs_noise), so that sampled initial states/targets and noise can be compared exactly.
    ctr = (env_id, counter, sub, stream + 256*block), key = seed -- see oracle/physics.py rng_bits."""

    def __init__(self, seed, env_id):
        from . import physics as ph
        self.ph, self.seed, self.env_id = ph, int(seed), int(env_id)
        self.episode, self.resample, self.steps = 0, 0, 0
        self._tblock, self._tfield = None, 0

    def _bits(self, counter, sub, stream, block=0):
        return self.ph.rng_bits(self.seed, [self.env_id], counter, stream + 256 * block, sub=sub)[0]

    def begin_episode(self, episode):
        self.episode, self.resample = int(episode), 0

    def init_state_uniform(self, var_index, low, high):
        b = self._bits(self.episode, var_index // 4, self.ph.STREAM_RESET_STATE)
        return low + (high - low) * self.ph.u01(b[var_index % 4])

    def begin_target(self):
        self._resample_now = self.resample
        self.resample += 1

    def target_uniform(self, k, low, high):
        self._tblock = self._bits(self.episode, self._resample_now, self.ph.STREAM_RESET_TARGET, block=k)
        self._tfield = 1
        return low + (high - low) * self.ph.u01(self._tblock[0])

    def uniform(self, low=0.0, high=1.0):
        u = self.ph.u01(self._tblock[self._tfield])
        self._tfield += 1
        return low + (high - low) * u

    def begin_obs(self, steps_count):
        self.steps = int(steps_count)

    # simulator.model (csrc/fwgym.hip k_model_draw): listed parameter i of the episode, ctr = (env, episode, i, STREAM_MODEL)
    def model_normal(self, i, loc, scale):
        b = self._bits(self.episode, i, self.ph.STREAM_MODEL)
        z = np.sqrt(-2.0 * np.log(self.ph.u01(b[0]))) * np.cos(2.0 * np.pi * self.ph.u01(b[1]))
        return loc + scale * z

    def model_uniform(self, i, low, high):
        b = self._bits(self.episode, i, self.ph.STREAM_MODEL)
        return low + (high - low) * self.ph.u01(b[0])

    # simulator.<key> sampling (fixed_wing.py:560-569; csrc/fwgym_env.h draw_sim_keys): key i of the episode,
    # ctr = (env, episode, i, STREAM_SIM_KEY); a choice takes the first value whose cumulative probability exceeds u
    def sim_choice(self, i, values, probs):
        u = self.ph.u01(self._bits(self.episode, i, self.ph.STREAM_SIM_KEY)[0])
        p = np.full(len(values), 1.0 / len(values)) if probs is None else np.asarray(probs, dtype=np.float64)
        cum = np.cumsum(p)
        k = int(np.searchsorted(cum, u, side="right"))
        return values[min(k, len(values) - 1)]

    def sim_uniform(self, i, low, high):
        return low + (high - low) * self.ph.u01(self._bits(self.episode, i, self.ph.STREAM_SIM_KEY)[0])

    # reward.randomize_scaling (k_model_draw): factor i of the episode, ctr = (env, episode, i, STREAM_REWARD_SCALE)
    def reward_scale_uniform(self, i, low, high):
        b = self._bits(self.episode, i, self.ph.STREAM_REWARD_SCALE)
        return low + (high - low) * self.ph.u01(b[0])

    def init_noise(self, row):
        # after the reset's own draw (steps >= 1) row 0 is never a padding row: row r takes component r - 1, so that a window
        # of up to five rows costs the device ONE Philox block per step (csrc/fwgym_env.h init_noise_bits<true>)
        idx = row if self.steps == 0 else row - 1
        b = self._bits(self.steps, self.episode, self.ph.STREAM_INIT_NOISE, block=idx // 4)
        return 2.0 * self.ph.u01(b[idx % 4]) - 1.0

    def obs_normal(self, idx, mean, std):
        if std == 0 and mean == 0:
            return 0.0  # the device skips the noise pass entirely in this case
        key = (self.steps, self.episode, idx // 4)
        if getattr(self, "_nkey", None) != key:
            b = self._bits(self.steps, self.episode, self.ph.STREAM_OBS_NOISE, block=idx // 4)
            self._nkey, self._nval = key, self.ph.box_muller(b[None, :])[0]
        return mean + std * self._nval[idx % 4]


class FixedWingOracle:
    def __init__(self, config, sim_config_kw=None, config_kw=None, sim_config_path=None, sim_parameter_path=None):
        if isinstance(config, str):
            with open(config) as f:
                config = json.load(f)
        self.cfg = copy.deepcopy(config)
        if config_kw is not None:
            _override(self.cfg, config_kw)
        cfg = self.cfg
        sim_kw = {} if sim_config_kw is None else copy.deepcopy(sim_config_kw)
        sim_kw["actuation"] = {"inputs": [a["name"] for a in cfg["action"]["states"]]}
        sim_kw["turbulence_sim_length"] = cfg["steps_max"]
        kw = {"config_kw": sim_kw}
        if sim_config_path is not None:
            kw["config_path"] = sim_config_path
        if sim_parameter_path is not None:
            kw["parameter_path"] = sim_parameter_path
        self.simulator = PyFly(**kw)
        sim = self.simulator
        self.steps_max = cfg["steps_max"]
        self.integration_window = cfg.get("integration_window", 0)   # fixed_wing.py:53
        self.err_hist = None      # per target: the episode's error history (only kept for integrator entries / int_error)
        self._rew_factors_init = copy.deepcopy(cfg["reward"]["factors"])   # fixed_wing.py:62
        self.goal_achieved = False           # sticky for the env's lifetime (fixed_wing.py:51,381-382)
        self.steps_count = None
        self.steps_for_target = None
        self.rng = MTStream(np.random.RandomState())
        self.obs_norm = cfg["observation"].get("normalize", False)

        # ---- observation bounds and normalisation defaults (fixed_wing.py:62-132)
        lows, highs = [], []
        for ov in cfg["observation"]["states"]:
            var = sim.state[ov["name"]]
            hi = ov.get("high", None)
            if hi is None:
                hi = var.value_max if var.value_max is not None else (
                    var.constraint_max if var.constraint_max is not None else F32MAX)
            elif ov.get("convert_to_radians", False):
                hi = np.radians(hi)
            lo = ov.get("low", None)
            if lo is None:
                lo = var.value_min if var.value_min is not None else (
                    var.constraint_min if var.constraint_min is not None else -F32MAX)
            elif ov.get("convert_to_radians", False):
                lo = np.radians(lo)
            bounded = hi != F32MAX and lo != -F32MAX
            if ov["type"] == "target" and ov["value"] == "relative":
                highs.append(hi - lo if bounded else F32MAX)
                lows.append(lo - hi if bounded else -F32MAX)
            else:
                highs.append(hi)
                lows.append(lo)
            if self.obs_norm:
                if ov.get("mean", None) is None:
                    ov["mean"] = hi - lo if bounded else 0
                if ov.get("var", None) is None:
                    ov["var"] = (hi - lo) / (4 ** 2) if bounded else 1
        L = cfg["observation"]["length"]
        if L > 1:
            if cfg["observation"]["shape"] == "vector":
                lows, highs = lows * L, highs * L
            elif cfg["observation"]["shape"] == "matrix":
                lows, highs = [lows for _ in range(L)], [highs for _ in range(L)]
            else:
                raise ValueError
        self.obs_low, self.obs_high = np.array(lows), np.array(highs)

        # ---- action scaling / spaces (fixed_wing.py:136-191)
        a_lo, a_hi, sp_lo, sp_hi = [], [], [], []
        for av in cfg["action"]["states"]:
            var = sim.state[av["name"]]
            s_hi = var.value_max if var.value_max is not None else (
                var.constraint_max if var.constraint_max is not None else F32MAX)
            s_lo = var.value_min if var.value_min is not None else (
                var.constraint_min if var.constraint_min is not None else -F32MAX)
            h, l = av.get("high", None), av.get("low", None)
            sp_hi.append(F32MAX if h == "max" else (s_hi if h is None else h))
            sp_lo.append(-F32MAX if l == "max" else (s_lo if l is None else l))
            a_hi.append(s_hi)
            a_lo.append(s_lo)
        self.action_scale_to_low, self.action_scale_to_high = np.array(a_lo), np.array(a_hi)
        self.action_space_low, self.action_space_high = np.array(sp_lo), np.array(sp_hi)
        self.scale_actions = cfg["action"].get("scale_space", False)
        self.n_act = len(cfg["action"]["states"])
        if cfg["action"].get("bounds_multiplier", None) is not None:
            self.action_bounds_max = np.full(self.n_act, cfg["action"].get("scale_high", 1)) * \
                cfg["action"]["bounds_multiplier"]
            self.action_bounds_min = np.full(self.n_act, cfg["action"].get("scale_low", -1)) * \
                cfg["action"]["bounds_multiplier"]
        self.goal_enabled = cfg["target"]["success_streak_req"] > 0
        self.target = None
        self.tprops = None
        self.tprops_init = None
        self.prev_shaping = {}
        self.set_curriculum_level(1)

    # ------------------------------------------------------------------------------------------------------------------
    def seed(self, seed=None):
        seed = 0 if seed is None else int(seed)
        self.rng = MTStream(np.random.RandomState(seed % (2 ** 32)))
        self.simulator.seed(seed)
        return [seed]

    def set_curriculum_level(self, level):
        """fixed_wing.py:224-285."""
        assert 0 <= level <= 1
        self.curriculum_level = level
        for st in self.cfg["simulator"].get("states", []):
            st = dict(st)
            name = st.pop("name")
            rad = st.pop("convert_to_radians", False)
            for prop, val in st.items():
                if val is not None:
                    if "constraint" not in prop and ("min" in prop or "max" in prop):
                        mid = (st[prop[:-3] + "max"] + st[prop[:-3] + "min"]) / 2
                        val = mid - level * (mid - val)
                    if rad:
                        val = np.radians(val)
                setattr(self.simulator.state[name], prop, val)
        init = {"states": {}}
        for attr, val in self.cfg["target"].items():
            if attr == "states":
                for st in val:
                    d = {}
                    for k, v in st.items():
                        if k == "name":
                            continue
                        if k not in ("bound", "class") and v is not None and not isinstance(v, bool):
                            mid = (st["high"] + v) / 2 if k == "low" else ((v + st["low"]) / 2 if k == "high" else 0)
                            v = mid - level * (mid - v)
                        d[k] = v
                    init["states"][st["name"]] = d
            elif isinstance(val, list):
                init[attr] = val[round(len(val) * level)]
            else:
                init[attr] = val
        self.tprops_init = init

    # ------------------------------------------------------------------------------------------------------------------
    def _wrap(self, name):
        return getattr(self.simulator.state[name], "wrap", False)

    def _error(self, name):
        """fixed_wing.py:890-914: wrap states give (value - target) folded to [-pi, pi); others (target - value)."""
        val = self.simulator.state[name].value
        if self._wrap(name):
            d = (val - self.target[name] + np.pi) % (2 * np.pi) - np.pi
            if d < -np.pi:
                d += 2 * np.pi
            return d
        return self.target[name] - val

    def _goal_status(self):
        st = {}
        for name, props in self.tprops.items():
            b = props.get("bound", None)
            if b is not None:
                st[name] = np.abs(self._error(name)) <= b
        st["all"] = all(st.values())
        return st

    def sample_target(self):
        """fixed_wing.py:461-521."""
        self.steps_for_target = 0
        self.target, self.tprops = {}, {}
        self.rng.begin_target()
        for k, (name, props) in enumerate(self.tprops_init["states"].items()):
            cls = props.get("class", "constant")
            if cls == "attitude_angular":
                raise NotImplementedError("target class attitude_angular")
            vp = {"class": cls}
            delta = props.get("delta", None)
            rad = props.get("convert_to_radians", False)
            low, high = props["low"], props["high"]
            if rad:
                low, high = np.radians(low), np.radians(high)
                delta = np.radians(delta) if delta is not None else None
            if delta is not None:
                x = self.simulator.state[name].value
                low = max(low, x - delta)
                high = max(min(high, x + delta), low)
            init = self.rng.target_uniform(k, low, high)
            if cls in "linear":
                vp["slope"] = self.rng.uniform(props["slope_low"], props["slope_high"])
                if self.rng.uniform() < 0.5:
                    vp["slope"] *= -1
                if rad:
                    vp["slope"] = np.radians(vp["slope"])
                vp["intercept"] = init
            elif cls == "sinusoidal":
                vp["amplitude"] = self.rng.uniform(props["amplitude_low"], props["amplitude_high"])
                if rad:
                    vp["amplitude"] = np.radians(vp["amplitude"])
                vp["period"] = self.rng.uniform(props.get("period_low", 250), props.get("period_high", 500))
                vp["phase"] = self.rng.uniform(0, 2 * np.pi) / (2 * np.pi / vp["period"])
                vp["bias"] = init - vp["amplitude"] * np.sin(2 * np.pi / vp["period"] *
                                                               (self.steps_count + vp["phase"]))
            b = props.get("bound", None)
            if b is not None:
                vp["bound"] = b if not rad else np.radians(b)
            self.target[name] = init
            self.tprops[name] = vp

    def get_simulator_parameters(self, normalize=True):
        """fixed_wing.py:872-888."""
        res = []
        model = self.cfg["simulator"].get("model", {})
        for param in model.get("parameters", []):
            val = self.simulator.params[param["name"]]
            if normalize:
                var = param.get("var", model["var"])
                if model.get("var_type", "relative") == "relative":
                    original_value = param.get("original", self.simulator.params[param["name"]])
                    if original_value == 0:
                        continue
                    var *= original_value
                val = (val - param.get("original", 0)) / var
            res.append(val)
        return res

    def _sample_sim_attrs(self):
        # fixed_wing.py:523-570: keys other than "states" are sampled and set on the simulator at each reset
        for key, value in self.cfg["simulator"].items():
            if key == "states":
                continue
            if key == "model":   # fixed_wing.py:532-559: the aircraft parameter table, one draw per listed parameter
                dist_type = value.get("distribution", "gaussian")
                n_drawn = 0
                for pa in value["parameters"]:
                    orig = pa.get("original", None)
                    if orig is None:
                        orig = self.simulator.params[pa["name"]]
                        pa["original"] = orig
                    if orig == 0:
                        continue
                    var = pa.get("var", value["var"])
                    if value["var_type"] == "relative":
                        var *= np.abs(orig)
                    if dist_type == "gaussian":
                        x = self.rng.model_normal(n_drawn, orig, var)
                        clip = pa.get("clip", value.get("clip", None))
                        if clip is not None:
                            if value["var_type"] == "relative":
                                clip *= orig
                            x = min(max(x, orig - clip), orig + clip)   # np.clip's order of operations
                    elif dist_type == "uniform":
                        x = self.rng.model_uniform(n_drawn, orig - var, orig + var)
                    else:
                        raise ValueError("Unexpected distribution type {}".format(dist_type))
                    n_drawn += 1
                    self.simulator.params[pa["name"]] = x
                continue
            k_idx = [k for k in self.cfg["simulator"] if k not in ("states", "model")].index(key)
            if "values" in value:
                probs = value.get("probabilities", None)
                val = self.rng.sim_choice(k_idx, value["values"], None if probs is None else np.array(probs))
            else:
                val = self.rng.sim_uniform(k_idx, value["low"], value["high"])
                if isinstance(value["low"], bool):
                    val = bool(val)
            setattr(self.simulator, key, val)

    # ------------------------------------------------------------------------------------------------------------------
    def reset(self, state=None, target=None, **sim_reset_kw):
        """fixed_wing.py:287-336."""
        cfg = self.cfg
        self.steps_count = 0
        self.rng.begin_episode(self.simulator.episode + 1)
        if isinstance(self.rng, PhiloxStream):
            sim_reset_kw = dict(sim_reset_kw, draw=self.rng.init_state_uniform)
        self.simulator.reset(state, **sim_reset_kw)
        self._sample_sim_attrs()
        self.sample_target()
        if target is not None:
            for k, v in target.items():
                if self.tprops[k]["class"] not in ("constant", "compensate"):
                    self.tprops[k]["class"] = "constant"
                self.target[k] = v
        names = list(self.target.keys())
        self.tnames = names
        # windows / accumulators (streaming form of the reference's history lists)
        self.raw_actions = []            # bounded to max window below
        self.commands = []
        self.max_window = max([ov.get("window_size", 1) for ov in cfg["observation"]["states"]
                               if ov["type"] == "action"] +
                              [f.get("window_size", 1) for f in cfg["reward"]["factors"]
                               if f["class"] == "action" and f["type"] == "delta"] + [1])
        L, step = cfg["observation"]["length"], cfg["observation"].get("step", 1)
        offs = list(range(1, (L + (1 if step == 1 else 0)) * step, step))
        self.row_offsets = offs
        self.lag_rows = {}               # time index -> un-normalised row-0 vector (kept for the last max(offs) steps)
        err0 = {k: self._error(k) for k in names}
        # integrator observations at reset (fixed_wing.py:317-321 builds the observation BEFORE it re-creates the histories, so
        # :804-810 reads the PREVIOUS episode's error history, or the "history is None" branch on the very first reset)
        self._err_hist_prev = self.err_hist
        self.err_hist = {k: [err0[k]] for k in names}
        self.err_last = dict(err0)
        self.m = {k: {"e0": err0[k], "sum": err0[k], "sum_abs": abs(err0[k]), "min": err0[k], "max": err0[k],
                      "n": 1, "last50": [err0[k]], "rise_lo": math.nan, "rise_hi": math.nan} for k in names}
        obs = self._observation(ok=True)
        self.goal_bits = {}
        self.goal_count = {}
        self.goal_n = 0
        self.settle = {}
        if self.goal_enabled:
            for s, flag in self._goal_status().items():
                self.goal_bits[s] = [bool(flag)]
                self.goal_count[s] = int(bool(flag))
                self.settle[s] = math.nan
            self.goal_n = 1
            self._update_settle()
        for term in cfg["reward"]["terms"]:
            self.prev_shaping[term["function_class"]] = None
        self.cmd_prev = None
        self.cmd_var_sum = 0.0
        self.n_cmds = 0
        if cfg["reward"].get("randomize_scaling", False):   # fixed_wing.py:330-334
            for i, rew_factor in enumerate(self._rew_factors_init):
                if isinstance(rew_factor["scaling"], list):
                    low, high = rew_factor["scaling"]
                    cfg["reward"]["factors"][i]["scaling"] = self.rng.reward_scale_uniform(i, low, high)
        return obs

    def _update_settle(self):
        req = self.cfg["target"]["success_streak_req"]
        frac = self.cfg["target"]["success_streak_fraction"]
        for s, bits in self.goal_bits.items():
            if len(bits) > req:
                del bits[0]
            if math.isnan(self.settle[s]) and len(bits) == req and np.mean(bits) >= frac:
                self.settle[s] = self.goal_n - 1

    def _scale_action(self, a, backward=False):
        lo, hi = self.cfg["action"].get("scale_low"), self.cfg["action"].get("scale_high")
        if not backward:
            return np.array(self.action_scale_to_high - self.action_scale_to_low) * (a - lo) / (hi - lo) + \
                self.action_scale_to_low
        return np.array(hi - lo) * (a - self.action_scale_to_low) / \
            (self.action_scale_to_high - self.action_scale_to_low) + lo

    # ------------------------------------------------------------------------------------------------------------------
    def step(self, action):
        """fixed_wing.py:338-437."""
        cfg = self.cfg
        raw = action
        self.raw_actions.append(raw)
        if len(self.raw_actions) > self.max_window + max(self.row_offsets):
            del self.raw_actions[0]
        assert not np.any(np.isnan(action))
        if self.scale_actions:
            action = self._scale_action(np.clip(action, cfg["action"].get("scale_low"), cfg["action"].get("scale_high")))
        ok, sim_info = self.simulator.step(list(action))
        cmd = np.array([self.simulator.state[a["name"]].command for a in cfg["action"]["states"]])
        self.commands.append(cmd)
        if len(self.commands) > self.max_window + max(self.row_offsets):
            del self.commands[0]
        if self.cmd_prev is not None:
            self.cmd_var_sum += float(np.sum(np.abs(cmd - self.cmd_prev)))
        self.cmd_prev = cmd
        self.n_cmds += 1
        self.steps_count += 1
        self.steps_for_target += 1
        info, done = {}, False
        if self.steps_count >= self.steps_max > 0:
            done = True
            info["termination"] = "steps"
        if ok:
            resample = False
            achieved_now = False
            if self.goal_enabled:
                for s, flag in self._goal_status().items():
                    self.goal_bits[s].append(bool(flag))
                    self.goal_count[s] += int(bool(flag))
                self.goal_n += 1
                self._update_settle()
                req = cfg["target"]["success_streak_req"]
                if self.steps_for_target >= req and \
                        np.mean(self.goal_bits["all"][-req:]) >= cfg["target"]["success_streak_fraction"]:
                    achieved_now = not self.goal_achieved
                    self.goal_achieved = True
                    mode = cfg["target"]["on_success"]
                    if mode == "done":
                        done = True
                        info["termination"] = "success"
                    elif mode == "new":
                        resample = True
                    elif mode != "none":
                        raise ValueError("Unexpected goal action")
            reward = self._reward(raw, achieved_now, cfg["reward"].get("form", "absolute") == "potential")
            every = cfg["target"].get("resample_every", 0)
            if resample or (every and self.steps_for_target >= every):
                self.sample_target()
            for k, v in self._next_target().items():
                self.target[k] = v
                e = self._error(k)
                m = self.m[k]
                # streaming rise-time crossing test (earliest downward crossing, fixed_wing.py:1130-1147)
                prev_abs, cur_abs = abs(self.err_last[k]), abs(e)
                lo_lim, hi_lim = self._rise_limits(k)
                idx_prev = m["n"] - 1
                if math.isnan(m["rise_lo"]) and prev_abs >= lo_lim and cur_abs < lo_lim:
                    m["rise_lo"] = idx_prev
                if math.isnan(m["rise_hi"]) and prev_abs >= hi_lim and cur_abs < hi_lim:
                    m["rise_hi"] = idx_prev
                m["sum"] += e
                m["sum_abs"] += abs(e)
                m["min"], m["max"] = min(m["min"], e), max(m["max"], e)
                m["n"] += 1
                self.err_hist[k].append(e)
                m["last50"].append(e)
                if len(m["last50"]) > 50:
                    del m["last50"][0]
                self.err_last[k] = e
            obs = self._observation(ok=True)
        else:
            done = True
            fail = cfg["reward"].get("step_fail", 0)
            reward = self.steps_count - self.steps_max if fail == "timesteps" else fail
            info["termination"] = sim_info["termination"]
            obs = self._observation(ok=False)
        if done:
            for metric in cfg.get("metrics", []):
                info[metric["name"]] = self.get_metric(metric["name"], **metric)
        info["target"] = self.target
        return obs, reward, done, info

    def _rise_limits(self, k):
        lo, hi = 0.1, 0.9
        for metric in self.cfg.get("metrics", []):
            if metric["name"] == "rise_time":
                lo, hi = metric.get("low", 0.1), metric.get("high", 0.9)
        e0 = self.m[k]["e0"]
        return np.abs(lo * e0), np.abs(hi * e0)

    # ------------------------------------------------------------------------------------------------------------------
    def _reward(self, raw, success, potential):
        """fixed_wing.py:674-774."""
        cfg = self.cfg
        terms = {t["function_class"]: {"val": 0, "weight": t["weight"], "shaping": 0} for t in cfg["reward"]["terms"]}
        for f in cfg["reward"]["factors"]:
            c, t = f["class"], f.get("type", None)
            if c == "action":
                if t == "value":
                    val = np.sum(np.abs(raw))
                elif t == "delta":
                    if self.steps_count > 1:
                        val = np.sum(np.abs(np.diff(self.raw_actions[-f["window_size"]:], axis=0)))
                    else:
                        val = 0
                elif t == "bound":
                    hi = np.where(raw > self.action_bounds_max, raw - self.action_bounds_max, 0)
                    lo = np.where(raw < self.action_bounds_min, raw - self.action_bounds_min, 0)
                    val = np.sum(np.abs(hi)) + np.sum(np.abs(lo))
                else:
                    raise ValueError
            elif c == "state":
                if t == "value":
                    val = self.simulator.state[f["name"]].value
                elif t == "error":
                    val = self._error(f["name"])
                elif t == "int_error":   # fixed_wing.py:708-711 (the history holds the errors up to the previous step here)
                    W, h = self.integration_window, self.err_hist[f["name"]]
                    val = float(np.sum(h[-W:])) if W > 0 else float(np.sum(h))
                    if self.steps_count < W:
                        val += (W - self.steps_count) * h[0]
                else:
                    raise NotImplementedError("reward type {}".format(t))
            elif c == "success":
                val = ((self.steps_max - self.steps_count) if f["value"] == "timesteps" else f["value"]) \
                    if success else 0
            elif c == "step":
                val = f["value"]
            elif c == "goal":
                val = 0
                st = self._goal_status()
                if t == "per_state":
                    for s, flag in st.items():
                        if s != "all":
                            val += f["value"] / len(self.target) if flag else 0
                elif t == "all":
                    val += f["value"] if st["all"] else 0
                else:
                    raise ValueError
            else:
                raise ValueError
            fc = f["function_class"]
            if fc == "linear":
                val = np.clip(np.abs(val) / f["scaling"], 0, f.get("max", None))
            elif fc in ("exponential", "quadratic"):
                val = val ** 2 / f["scaling"]
            else:
                raise ValueError
            terms[fc]["shaping" if f.get("shaping", False) else "val"] += val * np.sign(f.get("sign", -1))
        reward = 0
        for fc, ti in terms.items():
            prev = self.prev_shaping[fc]
            if fc == "exponential":
                if potential:
                    val = -1 + np.exp(ti["val"] + (ti["shaping"] - prev)) if prev is not None else -1 + np.exp(ti["val"])
                else:
                    val = -1 + np.exp(ti["val"] + ti["shaping"])
            else:
                val = ti["val"]
                if potential:
                    if prev is not None:
                        val += ti["shaping"] - prev
                else:
                    val += ti["shaping"]
            self.prev_shaping[fc] = ti["shaping"]
            reward += ti["weight"] * val
        return reward

    def _next_target(self):
        """fixed_wing.py:933-991."""
        res = {}
        dt = self.simulator.dt
        for name, props in self.tprops.items():
            cls = props.get("class", "constant")
            if cls == "constant":
                res[name] = self.target[name]
            elif cls == "compensate":
                if name != "Va":
                    raise NotImplementedError
                pcls = self.tprops["pitch"]["class"]
                if pcls in ("constant", "linear"):
                    pt = self.target["pitch"]
                elif pcls == "sinusoidal":
                    pt = self.tprops["pitch"]["bias"]
                else:
                    raise ValueError
                va = self.target["Va"]
                if pt <= np.radians(-2.5):
                    va_end = 28.434 - 40.0841 * pt
                    if va <= va_end:
                        slope = 7 * max(0, 1 if va < va_end * 0.95 else 1 - va / (va_end * 1.5))
                    else:
                        slope = 0
                    res[name] = va + (slope * (-self.target["pitch"]) - 0.25) * dt
                elif pt >= np.radians(5):
                    va_end = 26.27 - 41.2529 * pt
                    if va > va_end:
                        res[name] = va + (va_end - va) * 1 / 150 if self.steps_for_target < 750 else va_end
                    else:
                        res[name] = va
                else:
                    res[name] = va
            elif cls == "linear":
                res[name] = self.target[name] + props["slope"] * dt
            elif cls == "sinusoidal":
                res[name] = props["amplitude"] * np.sin(2 * np.pi / props["period"] *
                                                        (self.steps_count + props["phase"])) + props["bias"]
            else:
                raise ValueError
            if self._wrap(name) and np.abs(res[name]) > np.pi:
                res[name] = np.sign(res[name]) * (np.abs(res[name]) % np.pi - np.pi)
        return res

    # ------------------------------------------------------------------------------------------------------------------
    def _row0(self, reset_quirk=False):
        """Un-normalised, noise-free newest observation row; action entries hold the diff-sum when available else
        None (filled at assembly time with the current actuator value)."""
        cfg = self.cfg
        row = []
        anames = [a["name"] for a in cfg["action"]["states"]]
        for ov in cfg["observation"]["states"]:
            if ov["type"] == "state":
                row.append(self.simulator.state[ov["name"]].value)
            elif ov["type"] == "target":
                if ov["value"] == "integrator":
                    row.append(self._integrator_obs(ov["name"], reset_quirk))
                else:
                    row.append(self._error(ov["name"]) if ov["value"] == "relative" else self.target[ov["name"]])
            else:
                row.append(None)
        return row

    def _integrator_obs(self, name, reset_quirk):
        """fixed_wing.py:804-810 for the newest row (i = 1).  The windowed sum runs over history["error"][-W-1:-1], i.e. it
        ends BEFORE the newest error, padded with the initial error while the episode is younger than the window.
        reset_quirk: the value the reference returns in the reset observation (previous episode's history, see reset)."""
        W, t = self.integration_window, self.steps_count
        if reset_quirk:
            old = self._err_hist_prev
            if old is None:
                return self._error(name) * W
            h = old[name]
            return float(np.sum(h[-W - 1:-1])) + (W + 1) * h[0]    # steps_count - i = -1 < W always
        h = self.err_hist[name]                                    # t + 1 entries (the newest error included)
        val = float(np.sum(h[-W - 1:-1]))
        if t - 1 < W:
            val += (W - (t - 1)) * h[0]
        return val

    def _action_obs(self, ov, i):
        """Entry of type "action" for row offset i (fixed_wing.py:813-828)."""
        cfg = self.cfg
        anames = [a["name"] for a in cfg["action"]["states"]]
        ai = anames.index(ov["name"])
        if self.steps_count - i < 0:
            val = self.simulator.state[ov["name"]].value
            if self.scale_actions:
                a = np.zeros(shape=(sum(1 for o in cfg["observation"]["states"] if o["type"] == "action")))
                a[ai] = val
                val = self._scale_action(a, backward=True)[ai]
            return val
        w = ov.get("window_size", 1)
        hist = self.raw_actions if self.scale_actions else self.commands
        lo, hi = -w - i + 1, (None if i == 1 else -(i - 1))
        return np.sum(np.abs(np.diff([a[ai] for a in hist[lo:hi]])), dtype=np.float32)

    def _observation(self, ok):
        """fixed_wing.py:776-846 via the lag-row window: row k equals row 0 of k*step steps ago."""
        cfg = self.cfg
        L = cfg["observation"]["length"]
        noise = cfg["observation"].get("noise", None)
        t = self.steps_count
        # time index of the newest valid simulator/target record: t on success (and at reset), t-1 after a failed step
        newest = t if ok else t - 1
        if ok:
            self.lag_rows[newest] = self._row0()
            for old in [k for k in self.lag_rows if k < newest - max(self.row_offsets)]:
                del self.lag_rows[old]
        at_reset = ok and t == 0 and self.integration_window and any(
            ov["type"] == "target" and ov["value"] == "integrator" for ov in cfg["observation"]["states"])
        reset_row = self._row0(reset_quirk=True) if at_reset else None
        self.rng.begin_obs(t)
        obs = []
        n_idx = 0
        for r, i in enumerate(self.row_offsets):
            init_noise = None
            if i > t:
                i = t + 1
                if L > 1:
                    init_noise = self.rng.init_noise(r) * self.simulator.dt
            src = self.lag_rows[max(newest - (i - 1), 0)]
            if reset_row is not None:
                src = reset_row
            if r == 0 and not ok:
                # row 0 after a failed step: error against the (un-advanced) target with the last valid state
                src = self._row0()
            row = []
            for j, ov in enumerate(cfg["observation"]["states"]):
                val = self._action_obs(ov, i) if ov["type"] == "action" else src[j]
                if init_noise is not None:
                    val += init_noise
                if self.obs_norm and ov.get("norm", True):
                    val -= ov["mean"]
                    val /= ov["var"]
                if noise is not None:
                    val += self.rng.obs_normal(n_idx, noise["mean"], noise["var"])
                n_idx += 1
                row.append(val)
            if cfg["observation"]["shape"] == "vector":
                obs.extend(row)
            elif cfg["observation"]["shape"] == "matrix":
                obs.append(row)
            else:
                raise ValueError
        return np.array(obs)

    # ------------------------------------------------------------------------------------------------------------------
    def get_metric(self, metric, **kw):
        """fixed_wing.py:1095-1162 from the streaming accumulators."""
        res = {}
        m = self.m
        if metric == "avg_error":
            res = {k: np.abs((v["sum"] / v["n"]) / v["e0"]) if np.abs(v["e0"]) >= 0.01 else np.nan
                   for k, v in m.items()}
        if metric == "total_error":
            res = {k: v["sum_abs"] for k, v in m.items()}
        if metric == "end_error":
            res = {k: np.abs(np.mean(v["last50"])) for k, v in m.items()}
        if metric == "control_variation":
            with np.errstate(invalid="ignore", divide="ignore"):
                res["all"] = np.float64(self.cmd_var_sum) / (3 * self.simulator.dt * (self.n_cmds - 1))
        if metric in ("success", "settling_time"):
            for s in self.goal_bits:
                res[s] = (not math.isnan(self.settle[s])) if metric == "success" else self.settle[s]
        if metric == "rise_time":
            for k, v in m.items():
                res[k] = v["rise_lo"] - v["rise_hi"]
        if metric == "overshoot":
            for k, v in m.items():
                ext = v["min"] if v["e0"] > 0 else v["max"]
                res[k] = np.nan if np.sign(ext) == np.sign(v["e0"]) else np.abs(ext / v["e0"])
        if metric == "success_time_frac":
            res = {s: self.goal_count[s] / self.goal_n for s in self.goal_bits}
        return res
