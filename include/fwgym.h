/*
 * fwgym.h -- C ABI of libfwgym.so, the MI355X-native batched replacement for the hot path
 *            FixedWingAircraft.step()/reset() of eivindeb/fixed-wing-gym.
 *
 * The reference has no FFI: its hot path is Python calling Python (gym_fixed_wing/fixed_wing.py calling
 * pyfly.pyfly.PyFly).  Each entry point below names the reference interface it replaces for a whole batch of
 * N independent environments ("one wavefront lane = one aircraft").  All pointers are DEVICE pointers unless the
 * parameter name ends in _host.  No torch types cross this boundary; `stream` is a hipStream_t passed as void*.
 *
 * Conventions
 *   - every function returns 0 on success or a negative fwg_status; fwg_last_error() gives the message
 *     (thread-local); nothing throws across the ABI.
 *   - per-environment SIMULATION failures are data, not API errors: done=1 and term_code=FWG_TERM_VAR0+var_id,
 *     mirroring fixed_wing.py:409-416.
 *   - the caller owns every buffer, including the persistent state arena (rows x N 32-bit words in 16-byte
 *     groups [rows/4][env][4]); the library allocates only its small device-side constant block and reduction scratch in
 *     fwg_create.  No allocation and no synchronisation happens in fwg_step/fwg_reset.
 *   - a handle is not thread-safe; use one handle per GPU/stream.
 */
#ifndef FWGYM_H
#define FWGYM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FWG_ABI_VERSION 20

#define FWG_N_VARS 23        /* simulator variables, see fwg_var */
#define FWG_N_RESET_VARS 21  /* the keys of reset(state=...) records (fixed_wing.py:287,308; test-set format) */
#define FWG_N_PARAMS 49      /* aircraft parameters, see fwg_param */
#define FWG_MAX_OBS 32       /* observation.states entries per row (fixed_wing.py:796) */
#define FWG_MAX_ROWS 8       /* observation.length (fixed_wing.py:790) */
#define FWG_MAX_FACTORS 16   /* reward.factors (fixed_wing.py:683) */
#define FWG_MAX_TARGETS 3    /* target.states (fixed_wing.py:471) */
#define FWG_MAX_WINDOW 8     /* action window_size (fixed_wing.py:689,822) */
#define FWG_MAX_STREAK 128   /* target.success_streak_req (fixed_wing.py:377) */
#define FWG_END_WINDOW 50    /* end_error window (fixed_wing.py:1107) */
#define FWG_END_RING (FWG_END_WINDOW + 1)   /* slots of the cumulative-error ring: the window and the record just before it */
#define FWG_N_DRYDEN 8       /* Dryden filter states (joint realisation u | v,r | w,q | p) */
#define FWG_N_METRICS 28     /* rows of the metrics block written by fwg_step, see fwg_metric_row */
#define FWG_N_REDUCE 16      /* floats accumulated for fwg_reduce_success */

typedef enum fwg_status {
    FWG_OK = 0,
    FWG_ERR_INVALID = -1,     /* bad argument / unsupported configuration */
    FWG_ERR_ABI = -2,         /* struct size or version mismatch */
    FWG_ERR_HIP = -3,         /* a HIP runtime call failed */
    FWG_ERR_NAN_ACTION = -4   /* fwg_check_actions found NaN (reference: AssertionError, fixed_wing.py:347) */
} fwg_status;

/* simulator variable ids; 0..20 are exactly the keys of the reference's test-set "state" records */
typedef enum fwg_var {
    FWG_V_ROLL = 0, FWG_V_PITCH, FWG_V_YAW, FWG_V_OMEGA_P, FWG_V_OMEGA_Q, FWG_V_OMEGA_R,
    FWG_V_POS_N, FWG_V_POS_E, FWG_V_POS_D, FWG_V_VEL_U, FWG_V_VEL_V, FWG_V_VEL_W,
    FWG_V_VA, FWG_V_ALPHA, FWG_V_BETA, FWG_V_ELEVATOR, FWG_V_AILERON, FWG_V_THROTTLE,
    FWG_V_WIND_N, FWG_V_WIND_E, FWG_V_WIND_D, FWG_V_ELEVON_RIGHT, FWG_V_ELEVON_LEFT
} fwg_var;

/* aircraft parameter ids (names of the X8 parameter table, simulator.params[name] at fixed_wing.py:539) */
typedef enum fwg_param {
    FWG_P_MASS = 0, FWG_P_JX, FWG_P_JY, FWG_P_JZ, FWG_P_JXZ, FWG_P_S_WING, FWG_P_B, FWG_P_C, FWG_P_S_PROP,
    FWG_P_C_PROP, FWG_P_K_MOTOR, FWG_P_K_T_P, FWG_P_K_OMEGA, FWG_P_E, FWG_P_AR, FWG_P_M, FWG_P_A_0,
    FWG_P_C_LIFT_0, FWG_P_C_LIFT_ALPHA, FWG_P_C_LIFT_Q, FWG_P_C_LIFT_DELTA_E, FWG_P_C_D_P, FWG_P_C_D_BETA1, FWG_P_C_D_BETA2,
    FWG_P_C_D_Q, FWG_P_C_D_DELTA_E, FWG_P_C_M_0, FWG_P_C_M_ALPHA, FWG_P_C_M_Q, FWG_P_C_M_DELTA_E, FWG_P_C_M_FP,
    FWG_P_C_Y_0, FWG_P_C_Y_BETA, FWG_P_C_Y_P, FWG_P_C_Y_R, FWG_P_C_Y_DELTA_A, FWG_P_C_Y_DELTA_R,
    FWG_P_C_ROLL_0, FWG_P_C_ROLL_BETA, FWG_P_C_ROLL_P, FWG_P_C_ROLL_R, FWG_P_C_ROLL_DELTA_A, FWG_P_C_ROLL_DELTA_R,
    FWG_P_C_N_0, FWG_P_C_N_BETA, FWG_P_C_N_P, FWG_P_C_N_R, FWG_P_C_N_DELTA_A, FWG_P_C_N_DELTA_R
} fwg_param;

/* termination codes written to term_code_out (info["termination"], fixed_wing.py:368,385,416) */
enum {
    FWG_TERM_NONE = 0,
    FWG_TERM_STEPS = 1,      /* "steps"   */
    FWG_TERM_SUCCESS = 2,    /* "success" */
    FWG_TERM_VAR0 = 16,      /* FWG_TERM_VAR0 + fwg_var = name of the violated simulator variable */
    FWG_TERM_NAN = 255       /* a state became non-finite */
};

/* observation.states[i] (fixed_wing.py:796-838) */
enum { FWG_OBS_STATE = 0, FWG_OBS_TARGET_RELATIVE = 1, FWG_OBS_TARGET_ABSOLUTE = 2, FWG_OBS_ACTION = 3,
       FWG_OBS_TARGET_INTEGRATOR = 4 /* windowed error sum over integration_window steps (fixed_wing.py:804-810) */ };
typedef struct fwg_obs_desc {
    int32_t type;      /* FWG_OBS_* */
    int32_t src;       /* STATE: fwg_var; TARGET_*: index into target.states; ACTION: index into action.states */
    int32_t window;    /* ACTION: window_size (fixed_wing.py:822) */
    int32_t norm;      /* apply (val-mean)/var (fixed_wing.py:833-835) */
    double mean;
    double var;        /* the reference divides by "var" itself */
} fwg_obs_desc;

/* target.states[i] (fixed_wing.py:461-521, 933-991) */
enum { FWG_TGT_CONSTANT = 0, FWG_TGT_COMPENSATE = 1, FWG_TGT_LINEAR = 2, FWG_TGT_SINUSOIDAL = 3 };
typedef struct fwg_target_desc {
    int32_t var;       /* fwg_var */
    int32_t cls;       /* FWG_TGT_* */
    int32_t wrap;      /* simulator.state[name].wrap (fixed_wing.py:897,988) */
    int32_t has_delta;
    int32_t has_bound;
    int32_t pad_;
    double low, high;  /* radians where convert_to_radians; AFTER curriculum scaling (fixed_wing.py:256-265) */
    double delta;
    double bound;
    double slope_low, slope_high;          /* LINEAR (radians/s where flagged) */
    double amplitude_low, amplitude_high;  /* SINUSOIDAL */
    double period_low, period_high;
} fwg_target_desc;

/* reward.factors[i] (fixed_wing.py:683-751) */
enum { FWG_RC_STATE = 0, FWG_RC_ACTION = 1, FWG_RC_SUCCESS = 2, FWG_RC_STEP = 3, FWG_RC_GOAL = 4 };
enum { FWG_RT_VALUE = 0, FWG_RT_ERROR = 1, FWG_RT_DELTA = 2, FWG_RT_BOUND = 3, FWG_RT_PER_STATE = 4, FWG_RT_ALL = 5,
       FWG_RT_INT_ERROR = 6 /* fixed_wing.py:708-711 */ };
enum { FWG_FC_LINEAR = 0, FWG_FC_QUADRATIC = 1, FWG_FC_EXPONENTIAL = 2 };
typedef struct fwg_factor_desc {
    int32_t cls;       /* FWG_RC_* */
    int32_t type;      /* FWG_RT_* */
    int32_t src;       /* STATE/VALUE: fwg_var; STATE/ERROR: target index */
    int32_t fclass;    /* FWG_FC_* */
    int32_t shaping;
    int32_t window;    /* ACTION/DELTA */
    int32_t has_max;
    int32_t value_is_timesteps;  /* SUCCESS with value "timesteps" (fixed_wing.py:716) */
    double sign;       /* np.sign(sign) */
    double scaling;
    double max;
    double value;
} fwg_factor_desc;

enum { FWG_TURB_FILTER = 0, FWG_TURB_INCREMENT = 1 };
enum { FWG_ON_SUCCESS_NONE = 0, FWG_ON_SUCCESS_DONE = 1, FWG_ON_SUCCESS_NEW = 2 };

/* Flat, host-side "compiled" form of fixed_wing_config.json + the simulator config + the aircraft parameter table.
 * Angles are radians.  +-INFINITY encodes an absent min/max. */
typedef struct fwg_config {
    uint32_t abi_version;   /* FWG_ABI_VERSION */
    uint32_t struct_bytes;  /* sizeof(fwg_config) as seen by the caller */

    /* ---- simulator (replaces the PyFly object built at fixed_wing.py:41-46) */
    double dt, rho, g;
    int32_t n_substeps;     /* RK4 steps of the 13 rigid-body states per env step */
    int32_t actuator_microsteps; /* exact actuator micro-steps per env step (multiple of 2*n_substeps) */
    int32_t turbulence;     /* sim_config_kw["turbulence"] (examples/evaluate_controller.py:78) */
    int32_t turbulence_output;  /* FWG_TURB_INCREMENT (default of sim_config.json) | FWG_TURB_FILTER: what enters the airspeed and
                                 * body rates -- the first difference of the Dryden filter outputs (PyFly 0.1.2 as observed in
                                 * the reference's published traces) or the MIL-F-8785C outputs themselves */
    double param[FWG_N_PARAMS];
    double con_min[FWG_N_VARS], con_max[FWG_N_VARS];    /* Variable.constraint_min/max  */
    double val_min[FWG_N_VARS], val_max[FWG_N_VARS];    /* Variable.value_min/max       */
    double init_min[FWG_N_VARS], init_max[FWG_N_VARS];  /* Variable.init_min/max AFTER curriculum (fixed_wing.py:233-245) */
    double elevon_omega0[2], elevon_zeta[2], elevon_dot_max[2];  /* [right, left] */
    double throttle_tau;
    double dryden_A[FWG_N_DRYDEN * FWG_N_DRYDEN];  /* discrete x' = A x + B n, row-major */
    double dryden_B[FWG_N_DRYDEN * 4];             /* includes the sqrt(pi/dt) white-noise scaling */
    double dryden_C[6 * FWG_N_DRYDEN];             /* gust (u,v,w,p,q,r) = C x */

    /* ---- gym side */
    int32_t steps_max;                  /* fixed_wing.py:49 */
    int32_t obs_length, obs_step;       /* fixed_wing.py:786-790 */
    int32_t n_obs;                      /* len(observation.states) */
    int32_t obs_normalize;              /* fixed_wing.py:59 */
    int32_t obs_noise;                  /* observation.noise present with var != 0 or mean != 0 */
    double obs_noise_mean, obs_noise_std;
    fwg_obs_desc obs[FWG_MAX_OBS];

    int32_t n_actions;                  /* must be 3: elevator, aileron, throttle */
    int32_t scale_actions;              /* action.scale_space (fixed_wing.py:185,349) */
    double scale_low, scale_high;
    double act_to_low[3], act_to_high[3];      /* action_scale_to_low/high (fixed_wing.py:177-178) */
    int32_t has_action_bounds, pad0_;
    double act_bound_min[3], act_bound_max[3]; /* fixed_wing.py:187-191 */

    int32_t n_targets, resample_every, streak_req, on_success;  /* fixed_wing.py:377-398 */
    double streak_fraction;
    fwg_target_desc target[FWG_MAX_TARGETS];

    int32_t reward_potential;           /* reward.form == "potential" */
    int32_t step_fail_timesteps;        /* reward.step_fail == "timesteps" (fixed_wing.py:411-415) */
    double step_fail_value;
    int32_t term_present[3];            /* reward.terms by FWG_FC_* */
    int32_t n_factors;
    double term_weight[3];
    fwg_factor_desc factor[FWG_MAX_FACTORS];

    int32_t metrics;                    /* any entry in cfg["metrics"] (fixed_wing.py:419-421) */
    int32_t auto_reset;                 /* VecEnv semantics: a done env restarts inside the same fwg_step */
    int32_t store_derived;              /* keep roll/pitch/yaw/Va/alpha/beta of the committed state in the arena (host views) */
    int32_t obs_log_rows;   /* 0: fwg_step writes the dense [N][obs_dim] batch.  L > 0 (matrix observations without
                             * observation noise): observation history kept once, as a row log -- see fwg_obs_window */
    double rise_low, rise_high;         /* metrics[rise_time].low/high (fixed_wing.py:1131) */

    /* ---- simulator["model"]: aircraft parameters re-sampled for every env at every reset (sample_simulator_parameters,
     * fixed_wing.py:532-559).  Listed parameter i has id model_idx[i]; its nominal value is param[id].  Gaussian:
     * N(nominal, model_var[i]) then min(max(x, model_clip_lo[i]), model_clip_hi[i]) (-/+INFINITY = no clip; numpy's clip
     * order, so an interval given upside down -- relative clip of a negative nominal -- collapses to its upper end).
     * Uniform: U(nominal - model_var[i], nominal + model_var[i]).  Entries with nominal == 0 are skipped by the caller. */
    int32_t model_n;                    /* 0 = off */
    int32_t model_dist;                 /* 0 gaussian, 1 uniform */
    int32_t model_idx[FWG_N_PARAMS];
    int32_t pad_model_;
    double model_var[FWG_N_PARAMS], model_clip_lo[FWG_N_PARAMS], model_clip_hi[FWG_N_PARAMS];

    /* ---- reward["randomize_scaling"] (fixed_wing.py:330-334): factors whose scaling is given as [low, high] get a
     * scaling drawn U(low, high) for every env at every reset; low == high: the fixed factor[i].scaling */
    int32_t randomize_scaling;
    int32_t integration_window;  /* fixed_wing.py:53: window of the integrator observations / int_error factors (0: none; <= FWG_END_WINDOW - 1) */
    /* ---- simulator.<key> sampling at every reset (fixed_wing.py:560-569) for the two turbulence keys.  Per env and episode the
     * gust is scaled by on(turbulence) * W20(turbulence_intensity) / W20(the intensity dryden_C was built for): the intensity
     * only enters the MIL-F-8785C filters as an output gain.  n = 0: the key is not sampled (the configuration's value holds). */
    int32_t sk_n_intensity, sk_n_turbulence;
    int32_t sk_index_intensity, sk_index_turbulence;   /* position among the sampled keys (RNG sub-stream) */
    double sk_cum_intensity[4], sk_gain_intensity[4];  /* cumulative probabilities / gain of each listed value */
    double sk_cum_turbulence[2], sk_on_turbulence[2];
    double sk_base_gain;                               /* gain of the keys that are not sampled (0 when turbulence is configured off) */
    double factor_scaling_low[FWG_MAX_FACTORS], factor_scaling_high[FWG_MAX_FACTORS];
} fwg_config;

/* The caller-owned state arena is an array of 16-byte GROUPS [rows/4][N] of 32-bit words: word w of env e lives at
 * ((w >> 2) * N + e) * 4 + (w & 3).  All offsets below are in words and multiples of 4. */
typedef struct fwg_layout {
    int32_t rows;        /* total words per env (multiple of 4); arena = rows * N words */
    int32_t sim;         /* 28: e0 e1 e2 e3 | p q r pn | pe pd u v | w elevon_r elevon_l throttle | elevon_r_dot elevon_l_dot dryden0 dryden1 | dryden2..5 | dryden6 dryden7 pad pad */
    int32_t cold;        /* 8: wind_n wind_e wind_d episode | e0(3) pad -- per-episode constants, written by reset only */
    int32_t derived;     /* 8: roll pitch yaw Va | alpha beta pad pad of the committed state (written when store_derived) */
    int32_t gym;         /* 36: tgt0 tgt1 tgt2 steps_count|steps_for_target<<16 | flags window_counts goal_counts(2) | prev_cmd(3) sum_dcmd | settle(2) rise0 rise1 | rise2 sum_e(3) | sum_abs_e(3) prev_err0 | min_e(3) prev_err1 | max_e(3) prev_err2 | prev_shaping(3) pad */
    int32_t tprop;       /* 12: per target slope|amplitude, period, phase, bias (linear/sinusoidal targets) */
    int32_t goal;        /* 16 plain word rows [word][N] (NOT grouped): goal-window ring, 8 positions x 4 flags per word */
    int32_t act_ring;    /* window*4: raw actions (a0 a1 a2 pad) per slot, slot = global_step % window */
    int32_t cmd_ring;    /* window*4: constrained commands (only when observations need them) */
    int32_t end_ring;    /* 51*4: cumulative error sums of the episode (S0 S1 S2 pad) per slot, slot = global_step % 51 */
    int32_t lag_ring;    /* lag_depth*lag_groups*4: normalised observation records, slot = global_step % lag_depth */
    int32_t window;      /* action window depth */
    int32_t lag_depth;   /* (length-1)*step+1 */
    int32_t lag_groups;  /* ceil(n_obs/4) */
    int32_t draw;        /* 44 (+12 with linear/sinusoidal targets): the NEXT episode's reset draw, prepared ahead of time (cold) */
    int32_t aero;        /* 52 (model randomisation only): this episode's 49 force/moment constants of the env | - - - */
    int32_t aero_next;   /* 52: the next episode's | episode it is for, configuration generation, - */
    int32_t fscale;      /* 16 (reward.randomize_scaling only): 1 / scaling of every reward factor, this episode */
    int32_t fscale_next; /* 20: the next episode's | episode it is for, configuration generation, - - */
    int32_t model_raw;      /* model_n rounded up to 4: the sampled values of the listed parameters, this episode (get_simulator_parameters, fixed_wing.py:872-888) */
    int32_t model_raw_next; /* the same for the next episode (validity: the tag of aero_next) */
    int32_t fin;         /* 28 (metrics only): the raw accumulators of the env's last finished episode, turned into the metrics
                          * block and the success sums by fwg_finish_episodes / fwg_reduce_success* (flag bit 7 of the flags word) */
} fwg_layout;

/* rows of the metrics block (float32 [FWG_N_METRICS][N], valid where done) -- get_metric, fixed_wing.py:1095-1162 */
typedef enum fwg_metric_row {
    FWG_M_RISE_TIME = 0,          /* 3 */
    FWG_M_SETTLING_TIME = 3,      /* 4: target0..2, all */
    FWG_M_OVERSHOOT = 7,          /* 3 */
    FWG_M_TOTAL_ERROR = 10,       /* 3 */
    FWG_M_AVG_ERROR = 13,         /* 3 */
    FWG_M_CONTROL_VARIATION = 16, /* 1 */
    FWG_M_SUCCESS = 17,           /* 4 */
    FWG_M_SUCCESS_TIME_FRAC = 21, /* 4 */
    FWG_M_END_ERROR = 25          /* 3 */
} fwg_metric_row;

typedef struct fwg_handle fwg_handle;

/* Library/ABI version check. */
int fwg_abi_version(void);

/* Computes the arena layout for a configuration (pure host function). */
int fwg_get_layout(const fwg_config* cfg_host, fwg_layout* out_host);

/* Replaces FixedWingAircraft.__init__ (fixed_wing.py:14-212) for n_envs environments on HIP device `device`.
 * `state_arena` must hold layout.rows * n_envs 32-bit words (16-byte aligned, zero-initialised) and stay alive until
 * fwg_destroy.
 * `env_id_base` is the global index of this handle's first env (multi-GPU sharding: RNG streams depend only on the
 * global env index, so results do not depend on how envs are split over GPUs). */
int fwg_create(const fwg_config* cfg_host, int64_t n_envs, int device, void* state_arena, int64_t env_id_base,
               fwg_handle** out);
int fwg_destroy(fwg_handle* h);

/* Re-uploads the constants after a host-side change that keeps the layout (set_curriculum_level, fixed_wing.py:224-285;
 * setattr(simulator, key, val), fixed_wing.py:570). */
int fwg_update_config(fwg_handle* h, const fwg_config* cfg_host);

/* FixedWingAircraft.seed (fixed_wing.py:214-222): key of the counter-based device RNG. */
int fwg_seed(fwg_handle* h, uint64_t seed);

/* FixedWingAircraft.reset (fixed_wing.py:287-336) for the envs selected by `mask` (NULL = all).
 *   init_state : NULL or float32 [FWG_N_RESET_VARS][N]; NaN entries are sampled from init_min/init_max
 *                (reset(state=...) semantics, fixed_wing.py:308)
 *   init_target: NULL or float32 [n_targets][N]; NaN entries are sampled (reset(target=...), fixed_wing.py:311-315)
 *   obs_out    : float32 [N][obs_length*n_obs]; rows of unselected envs are left untouched */
int fwg_reset(fwg_handle* h, const uint8_t* mask, const float* init_state, const float* init_target,
              float* obs_out, void* stream);

/* FixedWingAircraft.step (fixed_wing.py:338-437) for all N envs, one fused launch.
 *   actions          : float32 [N][3] raw policy actions
 *   obs_out          : float32 [N][obs_length*n_obs]
 *   reward_out       : float32 [N]
 *   done_out         : uint8 [N]
 *   term_code_out    : uint8 [N]  FWG_TERM_*
 *   terminal_obs_out : NULL or float32 [N][obs_dim]; written for done envs only (VecEnv "terminal_observation")
 *   metrics_out      : NULL or float32 [FWG_N_METRICS][N]; written for done envs, by the NEXT fwg_finish_episodes /
 *                      fwg_reduce_success* call (the step itself only records the episode's accumulators).  LIFETIME: the
 *                      pointer of the LAST fwg_step is remembered and written through by fwg_reduce_success* -- keep the
 *                      buffer alive (or pass the same one every step) until the next collection; readers on the device
 *                      must order themselves after fwg_finish_episodes, not after fwg_step.  An env that ends two episodes
 *                      between collections keeps only the later one's column (both count in the success sums)
 *   target_out       : NULL or float32 [N][n_targets] = info["target"] (fixed_wing.py:435) after the step */
int fwg_step(fwg_handle* h, const float* actions, float* obs_out, float* reward_out, uint8_t* done_out,
             uint8_t* term_code_out, float* terminal_obs_out, float* metrics_out, float* target_out, void* stream);

/* Row-log observations (fwg_config.obs_log_rows = L > 0).  A lagged observation matrix is the newest record on top of
 * records the env already produced obs_step, 2 obs_step, ... steps ago; instead of copying those rows into a dense
 * batch every step (80 % of the observation traffic), fwg_step appends the new record to a log laid out as
 * float32 [obs_step][L][N][n_obs] -- the buffer passed as obs_out to fwg_reset / fwg_step, fwg_obs_log_floats() long --
 * in DESCENDING row order, so that the current observation of env e is the strided window
 *     obs[e][i][k] = log[((*plane + i) * N + e) * n_obs + k],  i < obs_length
 * (a zero-copy view: torch `log.view(-1, N, n_obs)[plane : plane + length].permute(1, 0, 2)`), valid until the next
 * fwg_step / fwg_reset.  Values are identical to the dense batch.  `plane` refers to the last completed step. */
int64_t fwg_obs_log_floats(const fwg_config* cfg_host, int64_t n_envs);
int fwg_obs_window(const fwg_handle* h, int64_t* plane);
/* Dense copy [N][obs_length * n_obs] of the current window of `obs_log` (the buffer fwg_step writes), for consumers that
 * need contiguous rows.  Stream-ordered; in graph mode the window position is read on the DEVICE, so the call may be
 * captured and replayed (a host-side view from fwg_obs_window is only valid for direct calls).  n_obs % 4 == 0. */
int fwg_obs_gather(const fwg_handle* h, const float* obs_log, float* obs_out, void* stream);

/* Known-answer hook: runs the device's Philox4x32-10 (the generator behind every sampled reset state, target, noise
 * and turbulence sample) on n inputs {counter[4], key[2]} (uint32 [n][6], device) -> uint32 [n][4] (device), so that
 * tests can check it against the published Random123 vectors.  Not used by the product path. */
int fwg_selftest_philox(const uint32_t* ctr_key_dev, uint32_t* out_dev, int64_t n, void* stream);

/* Debug-mode check for NaN actions (fixed_wing.py:347); synchronises the stream. */
int fwg_check_actions(fwg_handle* h, const float* actions, void* stream);

/* Local sums over the episodes finished since the last call: out_host[0]=episodes, [1..4]=success target0..2/all,
 * [5]=sum control_variation, [6..8]=sum end_error, [9..11]=sum total_error, [12..15]=sum success_time_frac.
 * These are the per-GPU contributions to the curriculum/logging reduction of
 * examples/train_rl_controller.py:51-66,80-85; the caller all-gathers them over RCCL.  Synchronises the stream and
 * clears the accumulators. */
int fwg_reduce_success(fwg_handle* h, float* out_host, void* stream);
/* Episode ends are recorded by fwg_step as a compact per-env record; THIS call (one small launch, stream-ordered, no
 * synchronisation) turns the records not yet collected into the metrics block (metrics_out: NULL or float32
 * [FWG_N_METRICS][N], written for the envs collected) and adds them to the success sums.  fwg_reduce_success and
 * fwg_reduce_success_device collect first (with the metrics_out of the last fwg_step).  Call it before reading the metrics of
 * the envs a step reported done; an env that ends a second episode before any collection folds the first record itself. */
int fwg_finish_episodes(fwg_handle* h, float* metrics_out, void* stream);
/* The same sums into a DEVICE buffer (16 floats), stream-ordered and without synchronising: the form to hand to the
 * RCCL all-gather directly, so that the rollout never drains the GPU for the success reduction. */
int fwg_reduce_success_device(fwg_handle* h, float* out_dev, void* stream);

/* Build-time specialisation support: writes the lowered STATIC configuration as 32-bit words (pure host function; the
 * build freezes such word lists into constexpr objects, see csrc/fwgym.hip "Specialisation").  Returns the number of
 * words, or a negative fwg_status. */
int fwg_dump_spec(const fwg_config* cfg_host, uint32_t* words_out_host, int64_t capacity);
/* Number of frozen configurations compiled into this library, and the kernel instance a handle runs: -1 = the generic
 * kernel (configuration interpreted at run time), i in [0, fwg_num_specs()) = frozen configuration i, FWG_INSTANCE_SHAPE + i =
 * the SHAPE instance of frozen configuration i -- the same structure (every count, type, source, flag: the integer members of
 * the lowered configuration) with the configuration's own VALUES (reward scalings, normalisation, constraints, aircraft
 * constants, noise levels, the time limit ...) read from memory: what a configuration that differs from a frozen one in values
 * only runs without any run-time compilation, ~1.1x the frozen kernel's step time (the generic kernel: ~35x). */
int fwg_num_specs(void);
int fwg_spec_index(const fwg_handle* h);
#define FWG_INSTANCE_SHAPE 1000
#define FWG_INSTANCE_GENERIC 100000
/* The instance fwg_create would pick for this configuration (pure host function): as fwg_spec_index, but FWG_INSTANCE_GENERIC
 * for the generic kernel, so that negative values stay fwg_status codes.  Replaces nothing in the reference (its env is
 * interpreted Python throughout); lets a host decide whether compiling a specialised kernel at run time is worth it. */
int fwg_config_instance(const fwg_config* cfg_host);

/* hipGraph support.  The ring positions of a launch depend on the global step counter; in graph mode that counter
 * lives on the device (each step launch publishes counter+1), so a captured sequence of fwg_step launches can be
 * replayed any number of times.  Rules: enable before capturing; capture an EVEN number of fwg_step calls per graph
 * (the counter is double-buffered by parity); after every replay tell the host how many steps ran
 * (fwg_note_replayed_steps) before issuing further direct calls.  Both functions synchronise `stream`/are host-only. */
int fwg_set_graph_mode(fwg_handle* h, int enable, void* stream);
int fwg_capture_begin(fwg_handle* h);   /* bracket the fwg_step calls issued under stream capture (they do not execute) */
int fwg_capture_end(fwg_handle* h);
int fwg_note_replayed_steps(fwg_handle* h, int64_t n_steps);
/* A captured sequence has the double-buffer copy its first launch reads baked in: it may only be replayed when the step
 * counter has the parity it had at fwg_capture_begin.  fwg_capture_parity: that parity (0 / 1) for the capture just
 * bracketed; fwg_replay_check: FWG_ERR_INVALID unless the handle's current step count has parity `capture_parity` -- call
 * it before every replay (host-only, no synchronisation).  With simulator.model / reward.randomize_scaling the captured
 * sequence is also tied to the configuration generation: fwg_capture_begin fails while every per-env parameter set is stale
 * (right after fwg_create / fwg_update_config / fwg_seed: issue fwg_reset or two direct steps first) and fwg_replay_check
 * fails for a sequence captured before the last fwg_update_config / fwg_seed (capture again). */
int fwg_capture_parity(const fwg_handle* h);
int fwg_replay_check(const fwg_handle* h, int capture_parity);


/* ---------------------------------------------------------------------------------------------------------------------
 * Rollout head ("actor"): what sits between two env steps in the reference's training/evaluation loops --
 * VecNormalize (running observation / return normalisation) around the env and the stable-baselines MlpPolicy
 * (pi and vf: obs -> 64 tanh -> 64 tanh -> act_dim / 1, state-independent log-std) acting on the normalised
 * observation (examples/train_rl_controller.py:223-231, examples/evaluate_controller.py:93-100) -- as two HIP kernels
 * on device-resident batches, so that one rollout step is three launches (fwg_step, fwg_actor_observe, fwg_actor_act).
 * The MLP runs on the matrix cores (bf16 MFMA, each fp32 operand split into bf16 hi + lo: three products per tile,
 * error ~1e-5 of the fp32 result), weights resident in LDS.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct fwg_actor fwg_actor;

/* Row-major float32 host arrays in torch.nn.Linear layout (weight [out][in], bias [out]); hidden width is 64. */
typedef struct fwg_actor_weights {
    const float *pi_w0, *pi_b0, *pi_w1, *pi_b1, *pi_w2, *pi_b2;   /* [64][obs_dim],[64],[64][64],[64],[act_dim][64],[act_dim] */
    const float *vf_w0, *vf_b0, *vf_w1, *vf_b1, *vf_w2, *vf_b2;   /* ... [1][64],[1] */
    const float *log_std;                                          /* [act_dim] */
} fwg_actor_weights;

/* Running statistics as VecNormalize keeps them (obs_rms, ret_rms). */
typedef struct fwg_actor_stats {
    float obs_mean[64], obs_var[64];
    float obs_count, ret_mean, ret_var, ret_count;
} fwg_actor_stats;

/* obs_dim <= 64 (a matrix observation is taken flattened), act_dim <= 4.  gamma/clip/epsilon: VecNormalize arguments
 * (defaults of the reference's scripts: 0.99, 10, 10, 1e-8). */
int fwg_actor_create(int device, int64_t n_envs, int obs_dim, int act_dim, float gamma, float clip_obs,
                     float clip_reward, float epsilon, fwg_actor** out);
void fwg_actor_destroy(fwg_actor* a);
int fwg_actor_set_weights(fwg_actor* a, const fwg_actor_weights* w_host);
int fwg_actor_set_stats(fwg_actor* a, const fwg_actor_stats* s_host, void* stream);
int fwg_actor_get_stats(fwg_actor* a, fwg_actor_stats* s_host, void* stream);   /* synchronises the stream */
/* training != 0: fwg_actor_act folds the batches seen by fwg_actor_observe into the running statistics (VecNormalize
 * training mode); 0: statistics frozen (evaluation).  precise != 0 (default): split-bf16 products; 0: plain bf16. */
int fwg_actor_configure(fwg_actor* a, int training, int precise);
/* Sampling noise: Philox stream (seed, env_id_base + env, act counter). */
int fwg_actor_seed(fwg_actor* a, uint64_t seed, int64_t env_id_base);
/* Attaches the head to an env (NULL detaches): from then on every fwg_step also leaves the batch moments of the
 * observations it writes and advances the discounted returns with the rewards it writes -- what fwg_actor_observe
 * would do in a launch of its own -- so that a rollout step is two launches (fwg_step, fwg_actor_act).  Same n_envs,
 * obs_dim and device required.  The env keeps a plain pointer: detach (or destroy the env) before fwg_actor_destroy. */
int fwg_attach_observer(fwg_handle* h, fwg_actor* a);
/* Row-log observations: after this call the `obs` argument of fwg_actor_observe / fwg_actor_act is `env`'s observation
 * row log and the head reads the window of the env's last completed step out of it (strided, no dense copy; position
 * taken from the host count for direct calls and from the device-resident positions in graph mode, so captured
 * sequences replay correctly).  NULL switches back to dense [N][obs_dim] batches.  The head keeps a plain pointer. */
int fwg_actor_set_obs_log(fwg_actor* a, const fwg_handle* env);
/* Accumulates the batch moments of `obs` ([N][obs_dim]) and, when `reward` is not NULL, advances the discounted
 * returns (ret = ret * gamma + reward, zeroed where `done`) and accumulates their moments (VecNormalize.step_wait). */
int fwg_actor_observe(fwg_actor* a, const float* obs, const float* reward, const uint8_t* done, void* stream);
/* Folds the accumulated moments into the running statistics (training mode), normalises `obs`, evaluates pi and vf,
 * samples the action (or takes the mean when deterministic != 0).  Outputs (each may be NULL):
 *   norm_obs_out [N][obs_dim], action_out [N][act_dim], value_out [N], logp_out [N],
 *   norm_reward_out [N] = clip(reward / sqrt(ret_var + eps)) for the `reward` given (the transition that led here),
 *   done_out [N] = copy of `done` (so that a rollout buffer is filled without extra copy launches).
 * Launches under stream capture must come in EVEN numbers per graph (the statistics are double-buffered by parity). */
int fwg_actor_act(fwg_actor* a, const float* obs, const float* reward, const uint8_t* done, float* norm_obs_out,
                  float* action_out, float* value_out, float* logp_out, float* norm_reward_out, uint8_t* done_out,
                  int deterministic, void* stream);

/* One rollout step in ONE launch: the head on the observation the env currently shows, then the env step under the actions
 * just sampled -- the body of the reference's training loop, VecNormalize(SubprocVecEnv).step_wait inside PPO2's runner
 * (examples/train_rl_controller.py:223-231), without a launch boundary between policy and env.  Equivalent to
 *     fwg_actor_act(head, obs_io, reward_io, done_io, norm_obs_out, action_out, value_out, logp_out, norm_reward_out,
 *                   done_prev_out, deterministic, stream);
 *     fwg_step(env, action_out, obs_io, reward_io, done_io, term_code_out, terminal_obs_out, metrics_out, NULL, stream);
 * -- the same arithmetic in the same order: the two paths agree bit for bit in every run of tests/test_rollout.py and of the soak
 * loops (tests/soak_rollout.py, tests/soak_suite_context.py: profiles/r05_soak.txt), with ONE unexplained exception on record: two
 * runs of the whole GPU suite in round 4 saw a difference that never reproduced.  gym_fixed_wing.rollout.FusedRollout therefore
 * uses this call only when asked to (fused=True / "auto").  obs_io / reward_io / done_io are IN-OUT: on entry what the env's previous step (or fwg_reset, with
 * reward_io / done_io zeroed) left there, on return this step's results; the four head outputs describe the observation on
 * entry, norm_reward_out / done_prev_out the transition that led to it (each may be NULL).  Needs `head` attached to `env`
 * (fwg_attach_observer: the step phase leaves the batch moments for the NEXT head) and a configuration
 * fwg_rollout_available() accepts: a build-time / run-time specialised kernel, dense observation batch (no row log), no
 * per-env aircraft parameters; callers fall back to the two calls above otherwise.  Counts as one fwg_step and one
 * fwg_actor_act for the hipGraph rules (even numbers per captured sequence). */
int fwg_rollout_available(const fwg_handle* env, const fwg_actor* head);
int fwg_rollout_step(fwg_handle* env, fwg_actor* head, float* norm_obs_out, float* action_out, float* value_out, float* logp_out,
                     float* norm_reward_out, uint8_t* done_prev_out, float* obs_io, float* reward_io, uint8_t* done_io,
                     uint8_t* term_code_out, float* terminal_obs_out, float* metrics_out, int deterministic, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Learner side of the rollout.  fwg_gae: generalised advantage estimation over the rollout the calls above
 * filled in place -- replaces the backward loop of PPO2's runner (stable-baselines ppo2.py Runner.run) behind
 * `PPO2(policy, env).learn(...)`, reference gym_fixed_wing/examples/train_rl_controller.py:231-232.  All buffers are device
 * pointers, step-major: rewards / values / adv_out / ret_out float [n_steps][n_envs], dones uint8 [n_steps][n_envs] (the flag
 * the env returned FOR step t: the value behind it belongs to the next episode), last_value float [n_envs] (value of the
 * observation after the last step).  adv_t = delta_t + gamma lam (1 - done_t) adv_(t+1), delta_t = r_t + gamma V_(t+1)
 * (1 - done_t) - V_t; ret = adv + V.  One launch on `stream`, no synchronisation, no allocation. */
int fwg_gae(int64_t n_steps, int64_t n_envs, const float* rewards, const float* values, const uint8_t* dones,
            const float* last_value, float gamma, float lam, float* adv_out, float* ret_out, void* stream);

/* Global step counter driving the ring slots (diagnostics/tests). */
int64_t fwg_global_step(const fwg_handle* h);

const char* fwg_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* FWGYM_H */
