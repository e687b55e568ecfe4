"""ctypes binding of libfwgym.so (include/fwgym.h).  The HIP library is the ONLY compute backend: importing this
module without a built library raises, there is no CPU fallback."""
import ctypes as C
import os

FWG_ABI_VERSION = 20
N_VARS = 23
N_RESET_VARS = 21
N_PARAMS = 49
MAX_OBS = 32
MAX_ROWS = 8
MAX_FACTORS = 16
MAX_TARGETS = 3
MAX_WINDOW = 8
MAX_STREAK = 128
END_WINDOW = 50
N_DRYDEN = 8
N_METRICS = 28
N_REDUCE = 16

VARS = ["roll", "pitch", "yaw", "omega_p", "omega_q", "omega_r", "position_n", "position_e", "position_d",
        "velocity_u", "velocity_v", "velocity_w", "Va", "alpha", "beta", "elevator", "aileron", "throttle",
        "wind_n", "wind_e", "wind_d", "elevon_right", "elevon_left"]
VAR_ID = {n: i for i, n in enumerate(VARS)}
PARAMS = ["mass", "Jx", "Jy", "Jz", "Jxz", "S_wing", "b", "c", "S_prop", "C_prop", "k_motor", "k_T_P", "k_Omega",
          "e", "ar", "M", "a_0", "C_L_0", "C_L_alpha", "C_L_q", "C_L_delta_e", "C_D_p", "C_D_beta1", "C_D_beta2",
          "C_D_q", "C_D_delta_e", "C_m_0", "C_m_alpha", "C_m_q", "C_m_delta_e", "C_m_fp", "C_Y_0", "C_Y_beta",
          "C_Y_p", "C_Y_r", "C_Y_delta_a", "C_Y_delta_r", "C_l_0", "C_l_beta", "C_l_p", "C_l_r", "C_l_delta_a",
          "C_l_delta_r", "C_n_0", "C_n_beta", "C_n_p", "C_n_r", "C_n_delta_a", "C_n_delta_r"]
assert len(PARAMS) == N_PARAMS and len(VARS) == N_VARS

TERM_NONE, TERM_STEPS, TERM_SUCCESS, TERM_VAR0, TERM_NAN = 0, 1, 2, 16, 255
OBS_STATE, OBS_TARGET_RELATIVE, OBS_TARGET_ABSOLUTE, OBS_ACTION, OBS_TARGET_INTEGRATOR = 0, 1, 2, 3, 4
TGT_CONSTANT, TGT_COMPENSATE, TGT_LINEAR, TGT_SINUSOIDAL = 0, 1, 2, 3
RC_STATE, RC_ACTION, RC_SUCCESS, RC_STEP, RC_GOAL = 0, 1, 2, 3, 4
RT_VALUE, RT_ERROR, RT_DELTA, RT_BOUND, RT_PER_STATE, RT_ALL, RT_INT_ERROR = 0, 1, 2, 3, 4, 5, 6
FC_LINEAR, FC_QUADRATIC, FC_EXPONENTIAL = 0, 1, 2
ON_SUCCESS = {"none": 0, "done": 1, "new": 2}
TURB_FILTER, TURB_INCREMENT = 0, 1

# rows of the metrics block
M_RISE_TIME, M_SETTLING_TIME, M_OVERSHOOT, M_TOTAL_ERROR, M_AVG_ERROR = 0, 3, 7, 10, 13
M_CONTROL_VARIATION, M_SUCCESS, M_SUCCESS_TIME_FRAC, M_END_ERROR = 16, 17, 21, 25


def term_name(code):
    """info["termination"] string of a termination code (reference fixed_wing.py:368,385,416)."""
    code = int(code)
    if code == TERM_STEPS:
        return "steps"
    if code == TERM_SUCCESS:
        return "success"
    if code == TERM_NAN:
        return "nan"
    if TERM_VAR0 <= code < TERM_VAR0 + N_VARS:
        return VARS[code - TERM_VAR0]
    return None


class ObsDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("src", C.c_int32), ("window", C.c_int32), ("norm", C.c_int32),
                ("mean", C.c_double), ("var", C.c_double)]


class TargetDesc(C.Structure):
    _fields_ = [("var", C.c_int32), ("cls", C.c_int32), ("wrap", C.c_int32), ("has_delta", C.c_int32),
                ("has_bound", C.c_int32), ("pad_", C.c_int32),
                ("low", C.c_double), ("high", C.c_double), ("delta", C.c_double), ("bound", C.c_double),
                ("slope_low", C.c_double), ("slope_high", C.c_double),
                ("amplitude_low", C.c_double), ("amplitude_high", C.c_double),
                ("period_low", C.c_double), ("period_high", C.c_double)]


class FactorDesc(C.Structure):
    _fields_ = [("cls", C.c_int32), ("type", C.c_int32), ("src", C.c_int32), ("fclass", C.c_int32),
                ("shaping", C.c_int32), ("window", C.c_int32), ("has_max", C.c_int32),
                ("value_is_timesteps", C.c_int32),
                ("sign", C.c_double), ("scaling", C.c_double), ("max", C.c_double), ("value", C.c_double)]


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("struct_bytes", C.c_uint32),
        ("dt", C.c_double), ("rho", C.c_double), ("g", C.c_double),
        ("n_substeps", C.c_int32), ("actuator_microsteps", C.c_int32), ("turbulence", C.c_int32), ("turbulence_output", C.c_int32),
        ("param", C.c_double * N_PARAMS),
        ("con_min", C.c_double * N_VARS), ("con_max", C.c_double * N_VARS),
        ("val_min", C.c_double * N_VARS), ("val_max", C.c_double * N_VARS),
        ("init_min", C.c_double * N_VARS), ("init_max", C.c_double * N_VARS),
        ("elevon_omega0", C.c_double * 2), ("elevon_zeta", C.c_double * 2), ("elevon_dot_max", C.c_double * 2),
        ("throttle_tau", C.c_double),
        ("dryden_A", C.c_double * (N_DRYDEN * N_DRYDEN)), ("dryden_B", C.c_double * (N_DRYDEN * 4)),
        ("dryden_C", C.c_double * (6 * N_DRYDEN)),
        ("steps_max", C.c_int32), ("obs_length", C.c_int32), ("obs_step", C.c_int32), ("n_obs", C.c_int32),
        ("obs_normalize", C.c_int32), ("obs_noise", C.c_int32),
        ("obs_noise_mean", C.c_double), ("obs_noise_std", C.c_double),
        ("obs", ObsDesc * MAX_OBS),
        ("n_actions", C.c_int32), ("scale_actions", C.c_int32),
        ("scale_low", C.c_double), ("scale_high", C.c_double),
        ("act_to_low", C.c_double * 3), ("act_to_high", C.c_double * 3),
        ("has_action_bounds", C.c_int32), ("pad0_", C.c_int32),
        ("act_bound_min", C.c_double * 3), ("act_bound_max", C.c_double * 3),
        ("n_targets", C.c_int32), ("resample_every", C.c_int32), ("streak_req", C.c_int32), ("on_success", C.c_int32),
        ("streak_fraction", C.c_double),
        ("target", TargetDesc * MAX_TARGETS),
        ("reward_potential", C.c_int32), ("step_fail_timesteps", C.c_int32),
        ("step_fail_value", C.c_double),
        ("term_present", C.c_int32 * 3), ("n_factors", C.c_int32),
        ("term_weight", C.c_double * 3),
        ("factor", FactorDesc * MAX_FACTORS),
        ("metrics", C.c_int32), ("auto_reset", C.c_int32), ("store_derived", C.c_int32), ("obs_log_rows", C.c_int32),
        ("rise_low", C.c_double), ("rise_high", C.c_double),
        ("model_n", C.c_int32), ("model_dist", C.c_int32), ("model_idx", C.c_int32 * N_PARAMS), ("pad_model_", C.c_int32),
        ("model_var", C.c_double * N_PARAMS), ("model_clip_lo", C.c_double * N_PARAMS), ("model_clip_hi", C.c_double * N_PARAMS),
        ("randomize_scaling", C.c_int32), ("integration_window", C.c_int32),
        ("sk_n_intensity", C.c_int32), ("sk_n_turbulence", C.c_int32), ("sk_index_intensity", C.c_int32), ("sk_index_turbulence", C.c_int32),
        ("sk_cum_intensity", C.c_double * 4), ("sk_gain_intensity", C.c_double * 4),
        ("sk_cum_turbulence", C.c_double * 2), ("sk_on_turbulence", C.c_double * 2), ("sk_base_gain", C.c_double),
        ("factor_scaling_low", C.c_double * MAX_FACTORS), ("factor_scaling_high", C.c_double * MAX_FACTORS),
    ]


class Layout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ["rows", "sim", "cold", "derived", "gym", "tprop", "goal", "act_ring", "cmd_ring", "end_ring", "lag_ring",
                 "window", "lag_depth", "lag_groups", "draw", "aero", "aero_next", "fscale", "fscale_next", "model_raw", "model_raw_next", "fin"]]


class ActorWeights(C.Structure):
    """fwg_actor_weights: float32 host arrays in torch.nn.Linear layout."""
    _fields_ = [(n, C.POINTER(C.c_float)) for n in
                ["pi_w0", "pi_b0", "pi_w1", "pi_b1", "pi_w2", "pi_b2", "vf_w0", "vf_b0", "vf_w1", "vf_b1", "vf_w2", "vf_b2",
                 "log_std"]]


class ActorStats(C.Structure):
    """fwg_actor_stats: VecNormalize's obs_rms / ret_rms."""
    _fields_ = [("obs_mean", C.c_float * 64), ("obs_var", C.c_float * 64), ("obs_count", C.c_float),
                ("ret_mean", C.c_float), ("ret_var", C.c_float), ("ret_count", C.c_float)]


class NativeError(RuntimeError):
    pass


INSTANCE_SHAPE = 1000       # include/fwgym.h FWG_INSTANCE_SHAPE: fwg_spec_index / fwg_config_instance values from here on are shape instances
INSTANCE_GENERIC = 100000   # FWG_INSTANCE_GENERIC (fwg_config_instance only)


_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libfwgym.so")
EXPORTS = ["fwg_abi_version", "fwg_get_layout", "fwg_create", "fwg_destroy", "fwg_update_config", "fwg_seed",
           "fwg_reset", "fwg_step", "fwg_check_actions", "fwg_reduce_success", "fwg_global_step", "fwg_last_error",
           "fwg_dump_spec", "fwg_num_specs", "fwg_spec_index", "fwg_config_instance", "fwg_set_graph_mode", "fwg_note_replayed_steps",
           "fwg_capture_begin", "fwg_capture_end", "fwg_capture_parity", "fwg_replay_check", "fwg_finish_episodes", "fwg_actor_create", "fwg_actor_destroy", "fwg_actor_set_weights",
           "fwg_actor_set_stats", "fwg_actor_get_stats", "fwg_actor_configure", "fwg_actor_seed", "fwg_actor_observe",
           "fwg_actor_act", "fwg_attach_observer", "fwg_obs_log_floats", "fwg_obs_window", "fwg_reduce_success_device",
           "fwg_obs_gather", "fwg_actor_set_obs_log", "fwg_selftest_philox", "fwg_rollout_available", "fwg_rollout_step", "fwg_gae"]
_libs = {}


def load_library(path=None):
    """Loads libfwgym.so and declares the prototypes.  Raises NativeError when the library is missing: build it with
    `python __graft_entry__.py` (or `make -C fixed-wing-gym_amd`)."""
    path = os.path.abspath(path or os.environ.get("FWGYM_LIB", DEFAULT_LIB))
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise NativeError("HIP library {} not found: build it first (python -c 'import __graft_entry__ as g; "
                          "g.build()'); there is no CPU fallback".format(path))
    try:
        # PyTorch-ROCm bundles its own HIP runtime; it must be the one already mapped when libfwgym.so resolves
        # libamdhip64, otherwise the process ends up with two runtimes and no visible device
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise NativeError("{} does not export {}".format(path, name))
    vp, i64, u64 = C.c_void_p, C.c_int64, C.c_uint64
    lib.fwg_abi_version.restype = C.c_int
    lib.fwg_get_layout.argtypes = [C.POINTER(Config), C.POINTER(Layout)]
    lib.fwg_create.argtypes = [C.POINTER(Config), i64, C.c_int, vp, i64, C.POINTER(vp)]
    lib.fwg_destroy.argtypes = [vp]
    lib.fwg_update_config.argtypes = [vp, C.POINTER(Config)]
    lib.fwg_seed.argtypes = [vp, u64]
    lib.fwg_reset.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.fwg_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.fwg_check_actions.argtypes = [vp, vp, vp]
    lib.fwg_reduce_success.argtypes = [vp, C.POINTER(C.c_float), vp]
    lib.fwg_reduce_success_device.argtypes = [vp, vp, vp]
    lib.fwg_reduce_success_device.restype = C.c_int
    lib.fwg_global_step.argtypes = [vp]
    lib.fwg_global_step.restype = i64
    lib.fwg_last_error.restype = C.c_char_p
    lib.fwg_dump_spec.argtypes = [C.POINTER(Config), C.POINTER(C.c_uint32), i64]
    lib.fwg_dump_spec.restype = C.c_int
    lib.fwg_num_specs.restype = C.c_int
    lib.fwg_config_instance.argtypes = [C.POINTER(Config)]
    lib.fwg_config_instance.restype = C.c_int
    lib.fwg_spec_index.argtypes = [vp]
    lib.fwg_spec_index.restype = C.c_int
    lib.fwg_set_graph_mode.argtypes = [vp, C.c_int, vp]
    lib.fwg_set_graph_mode.restype = C.c_int
    lib.fwg_note_replayed_steps.argtypes = [vp, i64]
    lib.fwg_note_replayed_steps.restype = C.c_int
    lib.fwg_capture_begin.argtypes = [vp]
    lib.fwg_capture_begin.restype = C.c_int
    lib.fwg_capture_end.argtypes = [vp]
    lib.fwg_capture_end.restype = C.c_int
    lib.fwg_capture_parity.argtypes = [vp]
    lib.fwg_capture_parity.restype = C.c_int
    lib.fwg_replay_check.argtypes = [vp, C.c_int]
    lib.fwg_replay_check.restype = C.c_int
    lib.fwg_finish_episodes.argtypes = [vp, vp, vp]
    lib.fwg_finish_episodes.restype = C.c_int
    f32 = C.c_float
    lib.fwg_actor_create.argtypes = [C.c_int, i64, C.c_int, C.c_int, f32, f32, f32, f32, C.POINTER(vp)]
    lib.fwg_actor_destroy.argtypes = [vp]
    lib.fwg_actor_destroy.restype = None
    lib.fwg_actor_set_weights.argtypes = [vp, C.POINTER(ActorWeights)]
    lib.fwg_actor_set_stats.argtypes = [vp, C.POINTER(ActorStats), vp]
    lib.fwg_actor_get_stats.argtypes = [vp, C.POINTER(ActorStats), vp]
    lib.fwg_actor_configure.argtypes = [vp, C.c_int, C.c_int]
    lib.fwg_actor_seed.argtypes = [vp, u64, i64]
    lib.fwg_actor_observe.argtypes = [vp, vp, vp, vp, vp]
    lib.fwg_obs_log_floats.argtypes = [C.POINTER(Config), i64]
    lib.fwg_obs_log_floats.restype = i64
    lib.fwg_obs_window.argtypes = [vp, C.POINTER(i64)]
    lib.fwg_obs_window.restype = C.c_int
    lib.fwg_selftest_philox.argtypes = [vp, vp, i64, vp]
    lib.fwg_selftest_philox.restype = C.c_int
    lib.fwg_obs_gather.argtypes = [vp, vp, vp, vp]
    lib.fwg_obs_gather.restype = C.c_int
    lib.fwg_actor_set_obs_log.argtypes = [vp, vp]
    lib.fwg_actor_set_obs_log.restype = C.c_int
    lib.fwg_attach_observer.argtypes = [vp, vp]
    lib.fwg_attach_observer.restype = C.c_int
    lib.fwg_actor_act.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp]
    lib.fwg_rollout_available.argtypes = [vp, vp]
    lib.fwg_rollout_available.restype = C.c_int
    lib.fwg_rollout_step.argtypes = [vp, vp] + [vp] * 12 + [C.c_int, vp]
    lib.fwg_rollout_step.restype = C.c_int
    lib.fwg_gae.argtypes = [i64, i64, vp, vp, vp, vp, f32, f32, vp, vp, vp]
    lib.fwg_gae.restype = C.c_int
    for name in ("fwg_actor_create", "fwg_actor_set_weights", "fwg_actor_set_stats", "fwg_actor_get_stats",
                 "fwg_actor_configure", "fwg_actor_seed", "fwg_actor_observe", "fwg_actor_act"):
        getattr(lib, name).restype = C.c_int
    for name in ("fwg_get_layout", "fwg_create", "fwg_destroy", "fwg_update_config", "fwg_seed", "fwg_reset",
                 "fwg_step", "fwg_check_actions", "fwg_reduce_success"):
        getattr(lib, name).restype = C.c_int
    if lib.fwg_abi_version() != FWG_ABI_VERSION:
        raise NativeError("ABI mismatch: library {} vs binding {}".format(lib.fwg_abi_version(), FWG_ABI_VERSION))
    _libs[path] = lib
    return lib


def check(lib, status):
    if status != 0:
        msg = lib.fwg_last_error().decode("utf-8", "replace")
        if status == -4:
            raise AssertionError(msg)  # NaN action: the reference asserts (fixed_wing.py:347)
        raise NativeError("libfwgym error {}: {}".format(status, msg))
