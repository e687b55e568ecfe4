"""Named configuration presets: the feature mixes of the reference's shipped config variants expressed as edits of the
package default (reference files: gym_fixed_wing/fixed_wing_config.json, fixed_wing_config_dev.json,
examples/fixed_wing_config.json, examples/models/{mlp,cnn}_controller/fixed_wing_config.json), plus the two benchmark
workloads of BASELINE.json.  Used by the build to freeze kernel specialisations and by tests/bench."""
import copy
import json

from .config import DEFAULT_ENV_CONFIG


def default():
    with open(DEFAULT_ENV_CONFIG) as f:
        return json.load(f)


def preset(kind):
    if kind == "cnn_model16":   # the cnn variant with randomised aircraft (simulator["model"])
        cfg = preset("cnn")
        cfg["simulator"]["model"] = copy.deepcopy(MODEL_16)
        return cfg
    cfg = default()
    if kind == "default":
        return cfg
    if kind in ("examples", "mlp", "cnn"):
        obs = cfg["observation"]
        del obs["noise"]
        states = []
        for ov in obs["states"]:
            if ov["name"] in ("alpha", "beta"):
                continue
            ov = {k: v for k, v in ov.items() if k not in ("mean", "var")}
            if ov["type"] == "target":
                ov["value"] = "relative"
            states.append(ov)
        obs["states"] = states
        for f in cfg["reward"]["factors"]:
            if f["class"] == "state":
                f.pop("max", None)
            if f["class"] == "action" and f["type"] == "delta":
                f["scaling"] = 45
        for st in cfg["simulator"]["states"]:
            if st["name"].startswith("omega"):
                st["constraint_min"], st["constraint_max"] = -360, 360
        if kind == "mlp":
            cfg["action"]["scale_space"] = False
            cfg["target"]["states"][0]["bound"] = 3
            cfg["target"]["states"][1]["bound"] = 3
        if kind in ("mlp", "cnn"):
            for f in cfg["reward"]["factors"]:
                if f["class"] == "state":
                    f["max"] = 0.3
                if f["class"] == "action" and f["type"] == "delta":
                    f["scaling"] = 60
        if kind == "cnn":
            cfg.pop("integration_window", None)
            obs["length"], obs["shape"] = 5, "matrix"
            obs.pop("normalize", None)
            for ov in obs["states"]:
                if ov["name"] == "Va" and ov["type"] == "state":
                    ov.pop("low", None)
                    ov["high"] = 60
                if ov["type"] == "action":
                    ov.pop("norm", None)
            for a in cfg["action"]["states"]:
                a["low"], a["high"] = None, None
            for st in cfg["simulator"]["states"]:
                if st["name"].startswith("omega"):
                    st["constraint_min"], st["constraint_max"] = -720, 720
        return cfg
    if kind == "dev":
        cfg["observation"]["noise"]["var"] = 0.1
        cfg["action"]["scale_space"] = False
        cfg["target"]["states"][0]["bound"] = 3
        cfg["target"]["states"][1]["bound"] = 3
        cfg["simulator"]["states"] = [s for s in cfg["simulator"]["states"] if s["name"] != "Va"]
        for st in cfg["simulator"]["states"]:
            if st["name"].startswith("omega"):
                st["constraint_min"], st["constraint_max"] = None, None
        # the shipped dev file lists the body velocities a second time; entries are applied in order, the later ones win
        cfg["simulator"]["states"] += [{"name": "velocity_u", "init_min": 11, "init_max": 28},
                                       {"name": "velocity_v", "init_min": -5, "init_max": 5},
                                       {"name": "velocity_w", "init_min": -5, "init_max": 5}]
        cfg["render"]["plot_action"] = False
        return cfg
    raise KeyError(kind)


# (name, preset kind, config_kw, sim_config_kw): configurations whose kernels are frozen at build time
TURB_MODERATE = {"turbulence": True, "turbulence_intensity": "moderate"}
INTEGRATOR_4X64 = {"method": "rk4", "substeps": 4, "actuator_microsteps": 64}
# simulator["model"] (fixed_wing.py:532-559): every env flies its own aircraft, 16 parameters re-sampled at every reset
MODEL_16 = {"var_type": "relative", "var": 0.1, "clip": 0.2, "distribution": "gaussian",
            "parameters": [{"name": n} for n in ("mass", "Jx", "Jz", "C_L_alpha", "C_L_0", "C_D_p", "C_m_alpha", "C_m_q", "C_m_delta_e",
                                                 "C_Y_beta", "C_l_p", "C_l_delta_a", "C_n_beta", "C_n_r", "k_motor", "S_prop")]}
SPECIALISED = [
    ("c2_default", "default", None, None),                                   # BASELINE configs[1]
    ("c3_cnn_step2_dryden", "cnn", {"observation": {"step": 2}}, TURB_MODERATE),  # BASELINE configs[2]/[3]
    ("c5_examples", "examples", None, None),                                 # BASELINE configs[4]
    # "_lean": the same configurations with derived_views=False (no per-step roll/pitch/yaw/Va/alpha/beta rows)
    ("c2_default_lean", "default", None, None),
    ("c3_cnn_step2_dryden_lean", "cnn", {"observation": {"step": 2}}, TURB_MODERATE),
    ("c5_examples_lean", "examples", None, None),
    # "_log": observation history as a row log + zero-copy window (FixedWingVecEnv(obs_log_rows=OBS_LOG_ROWS))
    ("c3_cnn_step2_dryden_lean_log", "cnn", {"observation": {"step": 2}}, TURB_MODERATE),
    ("c3_cnn_step2_dryden_log", "cnn", {"observation": {"step": 2}}, TURB_MODERATE),
    # the c3 workload with 4 RK4 sub-steps and 64 actuator micro-steps per env step (bench.py side figure `integrator_4x64`)
    ("c3_hi_lean_log", "cnn", {"observation": {"step": 2}}, dict(TURB_MODERATE, integrator=INTEGRATOR_4X64)),
    # the c3 workload with randomised aircraft (bench.py side figure `randomised_aircraft`)
    ("c3_model16_lean_log", "cnn_model16", {"observation": {"step": 2}}, TURB_MODERATE),
    # the other configuration files the reference ships, as FixedWingVecEnv(config) constructs them by default (derived views on,
    # row log where it applies): examples/models/mlp_controller, examples/models/cnn_controller, fixed_wing_config_dev.json --
    # with these every shipped configuration runs a specialised kernel on a machine without hipcc (the interpreting kernel is 30x
    # slower; other configurations are compiled at run time, gym_fixed_wing/jit.py)
    ("ship_mlp", "mlp", None, None),
    ("ship_cnn_log", "cnn", None, None),
    ("ship_dev", "dev", None, None),
]
# rows per parity of the observation row log: depth 36 - (5 - 1) = 32 for the 5-row matrix observations, so that the window's
# position repeats every obs_step x 32 steps and captured chunks of 64 / 128 / 256 steps replay with the views handed out at
# capture time still right (FixedWingVecEnv.set_graph_mode)
OBS_LOG_ROWS = 36


def workload(name):
    """BASELINE.json workloads -> (config dict, config_kw, sim_config_kw, envs per GPU, description)."""
    if name == "c3":
        return (preset("cnn"), {"observation": {"step": 2}}, copy.deepcopy(TURB_MODERATE), 65536,
                "C3: 65536 envs/GPU, Dryden turbulence moderate, obs 5x12 lag step 2, auto-reset, metrics on")
    if name == "c2":
        return (preset("default"), None, None, 4096,
                "C2: 4096 envs/GPU, turbulence off, obs 14-vector, auto-reset, metrics on")
    if name == "c5":
        return (preset("examples"), None, None, 65536,
                "C5: 65536 envs/GPU, examples config (obs 12-vector), turbulence off")
    raise KeyError(name)
