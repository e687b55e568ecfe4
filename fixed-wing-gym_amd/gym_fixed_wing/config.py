"""Host-side config compiler: fixed_wing_config.json (+ simulator config + aircraft parameter table) -> the flat
fwg_config consumed by libfwgym.so.

Mirrors what the reference does at construction time and in set_curriculum_level
(gym_fixed_wing/fixed_wing.py:14-212 and :224-285) -- same JSON schema, same config_kw / sim_config_kw override
semantics -- but the result is a table of numbers uploaded once to the GPU instead of Python objects consulted
every step.
"""
import copy
import json
import math
import os

import numpy as np

from . import _native as nat

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_ENV_CONFIG = os.path.join(_HERE, "fixed_wing_config.json")
DEFAULT_SIM_CONFIG = os.path.join(_HERE, "sim_config.json")
DEFAULT_PARAMETERS = os.path.join(_HERE, "x8_param.json")
F32MAX = float(np.finfo(np.float32).max)


def set_config_attrs(parent, kws):
    """config_kw override (reference fixed_wing.py:24-29): dicts recurse, int keys address list items."""
    for attr, val in kws.items():
        if isinstance(val, dict) or isinstance(parent[attr], list):
            set_config_attrs(parent[attr], val)
        else:
            parent[attr] = val


def _deep_update(base, kw):
    for k, v in kw.items():
        if isinstance(v, dict) and isinstance(base.get(k, None), dict):
            _deep_update(base[k], v)
        else:
            base[k] = v


class SimVariable(object):
    """Attribute view of one simulator variable (radians): the surface the reference reads from PyFly Variables
    (fixed_wing.py:69-87,143-166,245,282-283,897)."""
    _PROPS = ("value_min", "value_max", "init_min", "init_max", "constraint_min", "constraint_max")

    def __init__(self, name, cfg):
        self.name = name
        unit = cfg.get("unit", "")
        conv = math.radians if unit in ("degrees", "degrees/s") else (lambda x: x)
        for p in self._PROPS:
            v = cfg.get(p, None)
            setattr(self, p, conv(float(v)) if v is not None else None)
        self.wrap = bool(cfg.get("wrap", False))
        self.order = cfg.get("order", None)
        self.tau = cfg.get("tau", None)
        self.omega_0 = cfg.get("omega_0", None)
        self.zeta = cfg.get("zeta", None)
        self.dot_max = conv(float(cfg["dot_max"])) if cfg.get("dot_max", None) is not None else None


def _expm(a):
    """Matrix exponential by scaling-and-squaring with a Taylor series (small dense matrices, float64)."""
    a = np.asarray(a, dtype=np.float64)
    nrm = np.linalg.norm(a, 1)
    s = max(0, int(math.ceil(math.log2(nrm))) + 4) if nrm > 0 else 0
    a = a / (2.0 ** s)
    e = np.eye(a.shape[0])
    term = np.eye(a.shape[0])
    for k in range(1, 24):
        term = term @ a / k
        e = e + term
    for _ in range(s):
        e = e @ e
    return e


def dryden_matrices(b, dt, h=100.0, va=25.0, intensity="light"):
    """MIL-F-8785C low-altitude Dryden model at nominal altitude h [m] and airspeed va [m/s] for wingspan b [m],
    as one joint 8-state discrete system x' = A x + B n (n ~ N(0,1)^4, the sqrt(pi/dt) white-noise scaling folded
    into B), gust = C x with outputs (u,v,w [m/s], p,q,r [rad/s]).  State order: u | v(2),r | w(2),q | p."""
    m2f, kn2ms = 3.281, 0.5144
    f2m = 1.0 / m2f
    h, va, b = h * m2f, va * m2f, b * m2f
    if intensity not in ("light", "moderate", "severe"):
        raise ValueError("turbulence_intensity must be light, moderate or severe")
    w20 = {"light": 15.0, "moderate": 30.0, "severe": 45.0}[intensity] * kn2ms * m2f
    sw = 0.1 * w20
    su = sw / (0.177 + 0.000823 * h) ** 0.4
    sv = su
    lu = h / (0.177 + 0.000823 * h) ** 1.2
    lv, lw = lu, h
    ku = su * math.sqrt(2 * lu / (math.pi * va))
    kv = sv * math.sqrt(lv / (math.pi * va))
    kw = sw * math.sqrt(lw / (math.pi * va))
    tu, tv1, tv2, tw1, tw2 = lu / va, math.sqrt(3.0) * lv / va, lv / va, math.sqrt(3.0) * lw / va, lw / va
    kp = sw * math.sqrt(0.8 / va) * (math.pi / (4 * b)) ** (1.0 / 6.0) / lw ** (1.0 / 3.0)
    kq = kr = 1.0 / va
    tp = 4 * b / (math.pi * va)
    tq, tr = tp, 3 * b / (math.pi * va)
    A = np.zeros((8, 8))
    B = np.zeros((8, 4))
    Cm = np.zeros((6, 8))
    A[0, 0], B[0, 0], Cm[0, 0] = -1 / tu, 1 / tu, f2m * ku

    def second_order(i0, noise, k, t1, t2, ta, ka, lin_row, ang_row, sign):
        A[i0, i0], B[i0, noise] = -1 / t2, 1 / t2
        A[i0 + 1, i0], A[i0 + 1, i0 + 1] = 1 / t2, -1 / t2
        f = np.zeros(8)
        f[i0], f[i0 + 1] = k * t1 / t2, k * (1 - t1 / t2)
        A[i0 + 2, :] = f / ta
        A[i0 + 2, i0 + 2] -= 1 / ta
        Cm[lin_row, :] = f2m * f
        Cm[ang_row, :] = sign * ka * f / ta
        Cm[ang_row, i0 + 2] -= sign * ka / ta

    second_order(1, 1, kv, tv1, tv2, tr, kr, 1, 5, +1.0)
    second_order(4, 2, kw, tw1, tw2, tq, kq, 2, 4, -1.0)
    A[7, 7], B[7, 3], Cm[3, 7] = -1 / tp, 1 / tp, kp
    aug = np.zeros((12, 12))
    aug[:8, :8], aug[:8, 8:] = A, B
    E = _expm(aug * dt)
    return E[:8, :8], E[:8, 8:] * math.sqrt(math.pi / dt), Cm


class EnvConfig(object):
    def __init__(self, config_path=None, sim_config_path=None, sim_parameter_path=None, config_kw=None,
                 sim_config_kw=None):
        config_path = DEFAULT_ENV_CONFIG if config_path is None else config_path
        if isinstance(config_path, dict):
            self.cfg = copy.deepcopy(config_path)
        else:
            with open(config_path) as f:
                self.cfg = json.load(f)
        if config_kw is not None:
            set_config_attrs(self.cfg, config_kw)
        cfg = self.cfg
        # simulator config: defaults <- file <- sim_config_kw <- what the gym forces (fixed_wing.py:37-41)
        with open(DEFAULT_SIM_CONFIG if sim_config_path is None else sim_config_path) as f:
            self.sim_cfg = json.load(f)
        kw = {} if sim_config_kw is None else copy.deepcopy(sim_config_kw)
        kw["actuation"] = {"inputs": [a["name"] for a in cfg["action"]["states"]]}
        kw["turbulence_sim_length"] = cfg["steps_max"]
        _deep_update(self.sim_cfg, kw)
        ppath = DEFAULT_PARAMETERS if sim_parameter_path is None else sim_parameter_path
        if ppath.endswith(".mat"):
            import scipy.io
            raw = scipy.io.loadmat(ppath, squeeze_me=True)
            self.params = {k: float(v) for k, v in raw.items() if not k.startswith("__") and np.ndim(v) == 0}
        else:
            with open(ppath) as f:
                self.params = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        missing = [p for p in nat.PARAMS if p not in self.params]
        if missing:
            raise KeyError("aircraft parameter file lacks {}".format(missing))
        self.dt = float(self.sim_cfg["dt"])
        self.turbulence = bool(self.sim_cfg.get("turbulence", False))
        self.turbulence_intensity = self.sim_cfg.get("turbulence_intensity", "light")
        if self.turbulence_intensity == "none":
            self.turbulence = False
            self.turbulence_intensity = "light"
        # "increment": the gust sample is the first difference of the Dryden filter outputs -- the reading of PyFly 0.1.2's
        # turbulence that the reference's published evaluation traces support; "filter": MIL-F-8785C
        self.turbulence_output = self.sim_cfg.get("turbulence_output", "increment")
        if self.turbulence_output not in ("increment", "filter"):
            raise ValueError("turbulence_output must be 'increment' or 'filter'")
        if self.sim_cfg["actuation"]["inputs"] != ["elevator", "aileron", "throttle"]:
            raise NotImplementedError("action.states must be [elevator, aileron, throttle]")
        if self.sim_cfg["actuation"].get("dynamics", None) != ["elevon_right", "elevon_left", "throttle"]:
            raise NotImplementedError("actuation.dynamics must be [elevon_right, elevon_left, throttle]")
        self.state = {}
        for st in self.sim_cfg["states"]:
            if st["name"] not in nat.VAR_ID:
                raise KeyError("unknown simulator state {}".format(st["name"]))
            self.state[st["name"]] = SimVariable(st["name"], st)
        for name in nat.VARS:
            if name not in self.state:
                self.state[name] = SimVariable(name, {})
        self._check_supported()

        self.steps_max = cfg["steps_max"]
        self.obs_norm = cfg["observation"].get("normalize", False)
        self.obs_norm_mean_mask = []
        self.obs_module_indices = {"pi": [], "vf": []}
        lows, highs = [], []
        for i, ov in enumerate(cfg["observation"]["states"]):   # fixed_wing.py:64-118
            self.obs_norm_mean_mask.append(ov.get("mask_mean", False))
            var = self.state[ov["name"]]
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
            if ov.get("module", "all") != "all":
                self.obs_module_indices[ov["module"]].append(i)
            else:
                self.obs_module_indices["pi"].append(i)
                self.obs_module_indices["vf"].append(i)
        self.obs_exclusive_states = self.obs_module_indices["pi"] != self.obs_module_indices["vf"]
        L = cfg["observation"]["length"]
        if L > 1:
            if cfg["observation"]["shape"] == "vector":
                lows, highs = lows * L, highs * L
                self.obs_norm_mean_mask = self.obs_norm_mean_mask * L
            elif cfg["observation"]["shape"] == "matrix":
                lows, highs = [lows for _ in range(L)], [highs for _ in range(L)]
                self.obs_norm_mean_mask = [self.obs_norm_mean_mask for _ in range(L)]
            else:
                raise ValueError
        self.obs_norm_mean_mask = np.array(self.obs_norm_mean_mask)
        self.obs_low, self.obs_high = np.array(lows), np.array(highs)
        self.obs_shape = self.obs_low.shape

        a_lo, a_hi, sp_lo, sp_hi = [], [], [], []
        for av in cfg["action"]["states"]:   # fixed_wing.py:140-174
            var = self.state[av["name"]]
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
        self.action_bounds_max = self.action_bounds_min = None
        if cfg["action"].get("bounds_multiplier", None) is not None:
            self.action_bounds_max = np.full(3, cfg["action"].get("scale_high", 1)) * cfg["action"]["bounds_multiplier"]
            self.action_bounds_min = np.full(3, cfg["action"].get("scale_low", -1)) * cfg["action"]["bounds_multiplier"]
        self.goal_enabled = cfg["target"]["success_streak_req"] > 0
        self.target_names = [t["name"] for t in cfg["target"]["states"]]
        self.curriculum_level = None
        self.target_props_init = None
        self.set_curriculum_level(1)

    # ------------------------------------------------------------------------------------------------------------------
    W20_KNOTS = {"light": 15.0, "moderate": 30.0, "severe": 45.0}

    def _lower_sim_keys(self, c):
        """simulator.turbulence / simulator.turbulence_intensity sampled at every reset (fixed_wing.py:560-569) -> the tables
        of the per-env gust gain.  With either key sampled the kernels run with turbulence compiled in."""
        keys = [k for k in self.cfg["simulator"] if k not in ("states", "model")]
        c.sk_n_intensity = c.sk_n_turbulence = 0
        c.sk_base_gain = 1.0
        if not keys:
            return
        base = self.W20_KNOTS[self.turbulence_intensity]

        def cum(value, n):
            p = value.get("probabilities", None)
            p = np.full(n, 1.0 / n) if p is None else np.asarray(p, dtype=np.float64)
            return np.cumsum(p)

        on_configured = 1.0 if self.turbulence else 0.0
        if "turbulence_intensity" in keys:
            v = self.cfg["simulator"]["turbulence_intensity"]
            n = len(v["values"])
            c.sk_n_intensity, c.sk_index_intensity = n, keys.index("turbulence_intensity")
            for i, (name, cp) in enumerate(zip(v["values"], cum(v, n))):
                c.sk_cum_intensity[i], c.sk_gain_intensity[i] = cp, self.W20_KNOTS[name] / base
        if "turbulence" in keys:
            v = self.cfg["simulator"]["turbulence"]
            c.sk_index_turbulence = keys.index("turbulence")
            if "values" in v:
                n = len(v["values"])
                c.sk_n_turbulence = n
                for i, (val, cp) in enumerate(zip(v["values"], cum(v, n))):
                    c.sk_cum_turbulence[i], c.sk_on_turbulence[i] = cp, 1.0 if val else 0.0
            else:   # bool(uniform(low, high)): True unless the draw is exactly 0
                c.sk_n_turbulence = 1
                c.sk_cum_turbulence[0], c.sk_on_turbulence[0] = 1.0, 1.0 if (v["high"] or v["low"]) else 0.0
        else:
            c.sk_base_gain = on_configured
        c.turbulence = 1

    def _check_supported(self):
        cfg = self.cfg
        if cfg.get("integration_window", 0) > 49:   # the windowed sums are differences over the 51-slot cumulative error ring
            raise NotImplementedError("integration_window > 49")
        for key, value in cfg["simulator"].items():
            if key in ("states", "model"):
                continue
            # fixed_wing.py:560-569 samples ANY attribute of the simulator object; here the two turbulence keys (the intensity
            # is an output gain of the Dryden filters, so a per-env choice costs one multiplication)
            if key not in ("turbulence", "turbulence_intensity"):
                raise NotImplementedError("simulator.{} sampling (supported: turbulence, turbulence_intensity)".format(key))
            if "values" in value:
                if len(value["values"]) > (4 if key == "turbulence_intensity" else 2):
                    raise NotImplementedError("simulator.{}: too many values".format(key))
            elif not ("low" in value and "high" in value and key == "turbulence"):
                raise ValueError("simulator.{} needs values[, probabilities] (or low/high for turbulence)".format(key))
        if "model" in cfg["simulator"]:
            m = cfg["simulator"]["model"]
            if m.get("distribution", "gaussian") not in ("gaussian", "uniform"):
                raise ValueError("Unexpected distribution type {}".format(m.get("distribution")))   # fixed_wing.py:557
            for pa in m["parameters"]:
                if pa["name"] not in nat.PARAMS:
                    raise NotImplementedError("simulator.model parameter {} is not part of the force/moment model".format(pa["name"]))
        for t in cfg["target"]["states"]:
            if t.get("class", "constant") not in ("constant", "compensate", "linear", "sinusoidal"):
                raise NotImplementedError("target class {}".format(t.get("class")))
        for ov in cfg["observation"]["states"]:
            if ov["type"] == "target" and ov["value"] not in ("relative", "absolute", "integrator"):
                raise NotImplementedError("observation value {}".format(ov["value"]))
        for f in cfg["reward"]["factors"]:
            if f["class"] == "state" and f["type"] not in ("value", "error", "int_error"):
                raise NotImplementedError("reward type {}".format(f["type"]))

    def set_curriculum_level(self, level):
        """fixed_wing.py:224-285: init ranges and target ranges shrink towards their midpoints."""
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
                setattr(self.state[name], prop, val)
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
        self.target_props_init = init

    # ------------------------------------------------------------------------------------------------------------------
    def compile(self, auto_reset=True, store_derived=True, obs_log_rows=0):
        """-> _native.Config"""
        cfg, c = self.cfg, nat.Config()
        c.abi_version, c.struct_bytes = nat.FWG_ABI_VERSION, nat.C.sizeof(nat.Config)
        c.dt, c.rho, c.g = self.dt, float(self.sim_cfg["rho"]), float(self.sim_cfg["g"])
        integ = self.sim_cfg.get("integrator", {})
        if integ.get("method", "rk4") != "rk4":
            raise NotImplementedError("integrator method {}".format(integ.get("method")))
        c.n_substeps = int(integ.get("substeps", 1))
        c.actuator_microsteps = int(integ.get("actuator_microsteps", 16))
        c.turbulence = int(self.turbulence)
        self._lower_sim_keys(c)
        c.turbulence_output = {"filter": nat.TURB_FILTER, "increment": nat.TURB_INCREMENT}[self.turbulence_output]
        for i, p in enumerate(nat.PARAMS):
            c.param[i] = float(self.params[p])
        for i, name in enumerate(nat.VARS):
            v = self.state[name]
            c.con_min[i] = -math.inf if v.constraint_min is None else v.constraint_min
            c.con_max[i] = math.inf if v.constraint_max is None else v.constraint_max
            c.val_min[i] = -math.inf if v.value_min is None else v.value_min
            c.val_max[i] = math.inf if v.value_max is None else v.value_max
            c.init_min[i] = math.nan if v.init_min is None else v.init_min
            c.init_max[i] = math.nan if v.init_max is None else v.init_max
        for i in range(nat.N_RESET_VARS):
            if nat.VARS[i] in ("Va", "alpha", "beta"):
                continue
            if math.isnan(c.init_min[i]) or math.isnan(c.init_max[i]):
                raise ValueError("simulator state {} needs init_min/init_max".format(nat.VARS[i]))
        for i, name in enumerate(("elevon_right", "elevon_left")):
            v = self.state[name]
            if v.order != 2:
                raise NotImplementedError("elevons must be second-order actuators")
            c.elevon_omega0[i], c.elevon_zeta[i] = float(v.omega_0), float(v.zeta)
            c.elevon_dot_max[i] = math.inf if v.dot_max is None else v.dot_max
        thr = self.state["throttle"]
        if thr.order != 1:
            raise NotImplementedError("throttle must be a first-order actuator")
        c.throttle_tau = float(thr.tau)
        A, B, Cm = dryden_matrices(float(self.params["b"]), self.dt, float(self.sim_cfg.get("turbulence_nominal_altitude", 100.0)),
                                   float(self.sim_cfg.get("turbulence_nominal_airspeed", 25.0)), self.turbulence_intensity)
        for i, v in enumerate(A.ravel()):
            c.dryden_A[i] = v
        for i, v in enumerate(B.ravel()):
            c.dryden_B[i] = v
        for i, v in enumerate(Cm.ravel()):
            c.dryden_C[i] = v

        c.steps_max = int(self.steps_max)
        ocfg = cfg["observation"]
        c.obs_length, c.obs_step = int(ocfg["length"]), int(ocfg.get("step", 1))
        c.n_obs = len(ocfg["states"])
        c.obs_normalize = int(bool(self.obs_norm))
        noise = ocfg.get("noise", None)
        c.obs_noise = int(noise is not None and (noise["var"] != 0 or noise["mean"] != 0))
        c.obs_noise_mean = float(noise["mean"]) if noise is not None else 0.0
        c.obs_noise_std = float(noise["var"]) if noise is not None else 0.0   # the reference passes "var" as scale
        anames = [a["name"] for a in cfg["action"]["states"]]
        if c.n_obs > nat.MAX_OBS:
            raise ValueError("more than {} observation states".format(nat.MAX_OBS))
        for j, ov in enumerate(ocfg["states"]):
            d = c.obs[j]
            if ov["type"] == "state":
                d.type, d.src = nat.OBS_STATE, nat.VAR_ID[ov["name"]]
            elif ov["type"] == "target":
                d.type = {"relative": nat.OBS_TARGET_RELATIVE, "absolute": nat.OBS_TARGET_ABSOLUTE,
                          "integrator": nat.OBS_TARGET_INTEGRATOR}[ov["value"]]
                d.src = self.target_names.index(ov["name"])
            elif ov["type"] == "action":
                d.type, d.src = nat.OBS_ACTION, anames.index(ov["name"])
                d.window = int(ov.get("window_size", 1))
            else:
                raise Exception("Unexpected observation variable type: {}".format(ov["type"]))
            d.norm = int(bool(self.obs_norm and ov.get("norm", True)))
            d.mean = float(ov.get("mean", 0) or 0) if d.norm else 0.0
            d.var = float(ov.get("var", 1) or 1) if d.norm else 1.0

        acfg = cfg["action"]
        c.n_actions, c.scale_actions = len(anames), int(bool(self.scale_actions))
        c.scale_low, c.scale_high = float(acfg.get("scale_low", -1) or 0), float(acfg.get("scale_high", 1) or 0)
        for i in range(3):
            c.act_to_low[i], c.act_to_high[i] = self.action_scale_to_low[i], self.action_scale_to_high[i]
        c.has_action_bounds = int(self.action_bounds_max is not None)
        if c.has_action_bounds:
            for i in range(3):
                c.act_bound_min[i], c.act_bound_max[i] = self.action_bounds_min[i], self.action_bounds_max[i]

        tinit = self.target_props_init
        c.n_targets = len(self.target_names)
        if c.n_targets > nat.MAX_TARGETS:
            raise ValueError("more than {} target states".format(nat.MAX_TARGETS))
        c.resample_every = int(tinit.get("resample_every", 0) or 0)
        c.streak_req = int(tinit["success_streak_req"])
        c.streak_fraction = float(tinit.get("success_streak_fraction", 1.0))
        if tinit.get("on_success", "none") not in nat.ON_SUCCESS:
            raise ValueError("Unexpected goal action {}".format(tinit.get("on_success")))
        c.on_success = nat.ON_SUCCESS[tinit.get("on_success", "none")]
        cls_id = {"constant": nat.TGT_CONSTANT, "compensate": nat.TGT_COMPENSATE, "linear": nat.TGT_LINEAR,
                  "sinusoidal": nat.TGT_SINUSOIDAL}
        for k, name in enumerate(self.target_names):
            p, t = tinit["states"][name], c.target[k]
            rad = p.get("convert_to_radians", False)
            conv = (lambda x: float(np.radians(x))) if rad else float
            t.var, t.cls = nat.VAR_ID[name], cls_id[p.get("class", "constant")]
            t.wrap = int(self.state[name].wrap)
            t.low, t.high = conv(p["low"]), conv(p["high"])
            t.has_delta = int(p.get("delta", None) is not None)
            t.delta = conv(p["delta"]) if t.has_delta else 0.0
            t.has_bound = int(p.get("bound", None) is not None)
            t.bound = conv(p["bound"]) if t.has_bound else 0.0
            if t.cls == nat.TGT_LINEAR:
                t.slope_low, t.slope_high = conv(p["slope_low"]), conv(p["slope_high"])
            if t.cls == nat.TGT_SINUSOIDAL:
                t.amplitude_low, t.amplitude_high = conv(p["amplitude_low"]), conv(p["amplitude_high"])
                t.period_low, t.period_high = float(p.get("period_low", 250)), float(p.get("period_high", 500))
            if t.cls == nat.TGT_COMPENSATE and "pitch" not in self.target_names:
                raise ValueError("target class compensate needs a pitch target")

        rcfg = cfg["reward"]
        c.reward_potential = int(rcfg.get("form", "absolute") == "potential")
        fail = rcfg.get("step_fail", 0)
        c.step_fail_timesteps = int(fail == "timesteps")
        c.step_fail_value = 0.0 if fail == "timesteps" else float(fail)
        fc_id = {"linear": nat.FC_LINEAR, "quadratic": nat.FC_QUADRATIC, "exponential": nat.FC_EXPONENTIAL}
        for term in rcfg["terms"]:
            c.term_present[fc_id[term["function_class"]]] = 1
            c.term_weight[fc_id[term["function_class"]]] = float(term["weight"])
        c.n_factors = len(rcfg["factors"])
        if c.n_factors > nat.MAX_FACTORS:
            raise ValueError("more than {} reward factors".format(nat.MAX_FACTORS))
        for i, f in enumerate(rcfg["factors"]):
            d = c.factor[i]
            cls, typ = f["class"], f.get("type", None)
            if cls == "action":
                d.cls = nat.RC_ACTION
                d.type = {"value": nat.RT_VALUE, "delta": nat.RT_DELTA, "bound": nat.RT_BOUND}[typ]
                d.window = int(f.get("window_size", 1))
                if typ == "delta" and f["name"] != "action":
                    raise NotImplementedError("action delta factor over history '{}'".format(f["name"]))
            elif cls == "state":
                d.cls = nat.RC_STATE
                if typ == "value":
                    d.type, d.src = nat.RT_VALUE, nat.VAR_ID[f["name"]]
                else:
                    d.type = nat.RT_INT_ERROR if typ == "int_error" else nat.RT_ERROR
                    d.src = self.target_names.index(f["name"])
            elif cls == "success":
                d.cls = nat.RC_SUCCESS
                d.value_is_timesteps = int(f["value"] == "timesteps")
                d.value = 0.0 if d.value_is_timesteps else float(f["value"])
            elif cls == "step":
                d.cls, d.value = nat.RC_STEP, float(f["value"])
            elif cls == "goal":
                d.cls = nat.RC_GOAL
                d.type = {"per_state": nat.RT_PER_STATE, "all": nat.RT_ALL}[typ]
                d.value = float(f["value"])
            else:
                raise ValueError("Unexpected reward component type {}".format(cls))
            if f["function_class"] not in fc_id:
                raise ValueError("Unexpected function class {} for {}".format(f["function_class"], f.get("name")))
            d.fclass = fc_id[f["function_class"]]
            if not c.term_present[d.fclass]:
                raise KeyError(f["function_class"])   # the reference fails the same way (terms[...] lookup)
            d.shaping = int(bool(f.get("shaping", False)))
            d.sign = float(np.sign(f.get("sign", -1)))
            # reward.randomize_scaling (fixed_wing.py:330-334): a scaling given as [low, high] is drawn per env at every reset
            c.randomize_scaling = int(bool(rcfg.get("randomize_scaling", False)))
            if isinstance(f["scaling"], (list, tuple)):
                if not c.randomize_scaling:
                    raise ValueError("reward factor {}: a [low, high] scaling needs reward.randomize_scaling".format(f.get("name")))
                lo, hi = float(f["scaling"][0]), float(f["scaling"][1])
            else:
                lo = hi = float(f["scaling"])
            d.scaling = lo
            c.factor_scaling_low[i], c.factor_scaling_high[i] = lo, hi
            d.has_max = int(f.get("max", None) is not None)
            d.max = float(f["max"]) if d.has_max else 0.0

        metrics = cfg.get("metrics", [])
        c.integration_window = int(cfg.get("integration_window", 0) or 0)
        uses_window = any(ov["type"] == "target" and ov["value"] == "integrator" for ov in cfg["observation"]["states"]) or \
            any(f["class"] == "state" and f["type"] == "int_error" for f in cfg["reward"]["factors"])
        if not uses_window:
            c.integration_window = 0
        # (the windowed sums live on the episode's cumulative error sums, which the metric accumulators maintain)
        c.metrics = int(len(metrics) > 0 or c.integration_window > 0)
        c.rise_low, c.rise_high = 0.1, 0.9
        for m in metrics:
            if m["name"] == "rise_time":
                c.rise_low, c.rise_high = float(m.get("low", 0.1)), float(m.get("high", 0.9))
        c.auto_reset = int(bool(auto_reset))
        c.store_derived = int(bool(store_derived))
        c.obs_log_rows = int(obs_log_rows)
        self._compile_model(c)
        return c

    def _compile_model(self, c):
        """simulator["model"] (sample_simulator_parameters, fixed_wing.py:532-559) -> absolute spreads and clip intervals per
        listed parameter.  `original` is the value of the parameter file (fixed_wing.py:540-543); parameters whose original
        is 0 are never sampled (fixed_wing.py:544-545)."""
        c.model_n = 0
        model = self.cfg["simulator"].get("model", None)
        if model is None:
            return
        relative = model["var_type"] == "relative"
        c.model_dist = {"gaussian": 0, "uniform": 1}[model.get("distribution", "gaussian")]
        n = 0
        for pa in model["parameters"]:
            orig = pa.get("original", None)
            if orig is None:
                orig = self.params[pa["name"]]
            orig = float(orig)
            if orig == 0:
                continue
            if abs(orig - float(self.params[pa["name"]])) > 1e-12 * max(1.0, abs(orig)):
                raise NotImplementedError("simulator.model: `original` of {} differs from the parameter file".format(pa["name"]))
            var = float(pa.get("var", model["var"]))
            if relative:
                var *= abs(orig)
            clip = pa.get("clip", model.get("clip", None))
            lo, hi = -math.inf, math.inf
            if clip is not None and c.model_dist == 0:
                clip = float(clip) * (orig if relative else 1.0)   # signed, as the reference computes it
                lo, hi = orig - clip, orig + clip
            c.model_idx[n] = nat.PARAMS.index(pa["name"])
            c.model_var[n], c.model_clip_lo[n], c.model_clip_hi[n] = var, lo, hi
            n += 1
        c.model_n = n
