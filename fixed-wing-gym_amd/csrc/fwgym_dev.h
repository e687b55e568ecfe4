// fwgym_dev.h -- device-side constant block and small math helpers for the gfx950 kernels (internal, not ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fwgym.h"

#define FWG_WAVE 64
#define FWG_TILE_STRIDE 65  // LDS row stride (words) of the [entry][lane] tiles: (k + lane) % 32 banks, conflict-free

// flags word (counters row 2)
#define FWG_FLAG_GOAL_ACHIEVED 1u        // sticky for the env's lifetime (fixed_wing.py:51,381-382)
#define FWG_FLAG_PREV_VALID_SHIFT 1      // bits 1..3: prev_shaping[fclass] is not None (fixed_wing.py:327-328,756,765)
#define FWG_FLAG_FIN_PENDING (1u << 7)    // the env's finished-episode record (L.fin) has not been collected yet (k_finish)
#define FWG_FLAG_LAST_FAILED (1u << 31)   // the episode's last step failed: its histories are one record shorter (integration_window)
#define FWG_FLAG_RESAMPLE_SHIFT 8        // bits 8..31: target resample counter inside the episode (RNG sub-stream)

// Philox stream ids (shared with oracle/physics.py)
#define FWG_STREAM_TURB 1u
#define FWG_STREAM_RESET_STATE 2u
#define FWG_STREAM_RESET_TARGET 3u
#define FWG_STREAM_OBS_NOISE 4u
#define FWG_STREAM_INIT_NOISE 5u
#define FWG_STREAM_MODEL 7u              // simulator.model draws (6 = the rollout head's policy noise)
#define FWG_STREAM_SIM_KEY 9u
#define FWG_STREAM_REWARD_SCALE 8u       // reward.randomize_scaling draws

struct DevObs { int type, src, window, norm; float mean, inv_var; };
struct DevTarget { int var, cls, wrap, has_delta, has_bound; float bound; };
// ranges that set_curriculum_level rescales (fixed_wing.py:224-285): kept OUT of the compile-time-specialisable block
struct DynTarget { float low, high, delta, slope_low, slope_high, amp_low, amp_high, period_low, period_high; };
// simulator.model (fixed_wing.py:532-559): the parameters re-sampled at every reset and how
struct ModelCfg {
    int n, dist;                       // listed parameters; 0 gaussian (+ clip), 1 uniform
    int idx[FWG_N_PARAMS];             // fwg_param id of listed parameter i
    float var[FWG_N_PARAMS], lo[FWG_N_PARAMS], hi[FWG_N_PARAMS];   // per listed parameter: std / half-width, clip interval
    float nominal[FWG_N_PARAMS];       // the parameter table (by fwg_param id)
    float rho, g;
};
struct DynCfg {
    float init_min[FWG_N_VARS], init_max[FWG_N_VARS];
    DynTarget target[FWG_MAX_TARGETS];
    unsigned generation;   // bumped whenever what a reset draw depends on changes (ranges, seed): prepared draws are then discarded
    ModelCfg model;
    float fs_lo[FWG_MAX_FACTORS], fs_hi[FWG_MAX_FACTORS];   // reward.randomize_scaling: scaling ~ U(lo, hi) where lo < hi
    // simulator.turbulence / simulator.turbulence_intensity sampled at every reset (fixed_wing.py:560-569): tables of the
    // per-env gust gain (see fwg_config sk_*)
    int sk_n_int, sk_n_turb, sk_idx_int, sk_idx_turb;
    float sk_cum_int[4], sk_gain_int[4], sk_cum_turb[2], sk_on_turb[2], sk_base_gain;
};
// How the kernels see it: in the CONSTANT address space.  The run-time configuration does not change during a launch, but a
// kernel that also stores to global memory cannot prove that: through a plain pointer every `dc.x` after the first store is
// a vector-memory load with a round trip of its own (the reset draw read its ranges in 4-5 SERIAL round trips per piece);
// through address space 4 they are scalar loads, batched and cached in the scalar cache
#if defined(FWG_EMU) || !defined(__HIP_DEVICE_COMPILE__)
typedef const DynCfg DynCfgK;
#define FWG_KCONST(T) const T
#else
typedef const __attribute__((address_space(4))) DynCfg DynCfgK;
#define FWG_KCONST(T) const __attribute__((address_space(4))) T   /* launch-constant data behind a plain kernel argument */
#endif

// invalidates the wave's scalar data cache and hands the pointer back through the asm statement (loads through it cannot be
// issued before it): for launch-constant data that the PREVIOUS launch wrote.  (The two halves pass through vector registers
// -- wherever the compiler keeps the pointer -- and come back as scalars: the loads through it stay scalar loads.)
template <class P>
__device__ __forceinline__ P* fwg_fresh_scalar_view(P* p) {
#if !defined(FWG_EMU) && defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long x = (unsigned long long)(size_t)p;
    unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi)::"memory");
    lo = (unsigned)__builtin_amdgcn_readfirstlane((int)lo);
    hi = (unsigned)__builtin_amdgcn_readfirstlane((int)hi);
    return (P*)(size_t)(((unsigned long long)hi << 32) | lo);
#else
    return p;
#endif
}

// The constants of the force / moment model, pre-combined from the parameter table (same names and order as the block at
// the head of DevCfg).  One set for all envs (DevCfg) unless simulator.model re-samples the table per env and episode: then
// every lane carries its own (arena section L.aero), derived on the device by the same formulas (derive_aero).
#define FWG_AERO_LIST(X)                                                                                          \
    X(half_rho_S) X(mg) X(inv_mass) X(inv_Jy) X(G1) X(G2) X(G3) X(G4) X(G5) X(G6) X(G7) X(G8) X(M) X(Ma0)          \
    X(CL0) X(CLa) X(cLq) X(CLde) X(CDp) X(kInd) X(CDb1) X(CDb2) X(cDq) X(CDde)                                    \
    X(Cm0) X(Cma) X(cmq) X(Cmde) X(Cmfp) X(chord) X(span) X(CY0) X(CYb) X(cYp) X(cYr) X(CYda)                     \
    X(Cl0) X(Clb) X(clp) X(clr) X(Clda) X(Cn0) X(Cnb) X(cnp) X(cnr) X(Cnda) X(kprop) X(kmotor) X(ktp)
#define FWG_N_AERO 49
#define FWG_AERO_GROUPS 13   // 52 words: the 49 constants | tag: episode the set is for, configuration generation, -
template <class T> struct AeroT {
#define FWG_AERO_MEMBER(n) T n;
    FWG_AERO_LIST(FWG_AERO_MEMBER)
#undef FWG_AERO_MEMBER
};
typedef AeroT<float> Aero;
// P = parameter table by fwg_param id; the formulas of lower_config (host, double) and k_model_draw (device, float)
template <class T, class PA>
__host__ __device__ inline void derive_aero(const PA& P, T rho, T g, AeroT<T>& d) {
    const T pi = (T)3.14159265358979323846;
    d.half_rho_S = (T)0.5 * rho * P[FWG_P_S_WING];
    d.mg = P[FWG_P_MASS] * g;
    d.inv_mass = (T)1 / P[FWG_P_MASS];
    d.inv_Jy = (T)1 / P[FWG_P_JY];
    const T Jx = P[FWG_P_JX], Jy = P[FWG_P_JY], Jz = P[FWG_P_JZ], Jxz = P[FWG_P_JXZ];
    const T G = Jx * Jz - Jxz * Jxz;
    d.G1 = Jxz * (Jx - Jy + Jz) / G; d.G2 = (Jz * (Jz - Jy) + Jxz * Jxz) / G;
    d.G3 = Jz / G; d.G4 = Jxz / G; d.G5 = (Jz - Jx) / Jy; d.G6 = Jxz / Jy;
    d.G7 = ((Jx - Jy) * Jx + Jxz * Jxz) / G; d.G8 = Jx / G;
    d.M = P[FWG_P_M]; d.Ma0 = P[FWG_P_M] * P[FWG_P_A_0];
    d.CL0 = P[FWG_P_C_LIFT_0]; d.CLa = P[FWG_P_C_LIFT_ALPHA];
    d.cLq = P[FWG_P_C_LIFT_Q] * P[FWG_P_C]; d.CLde = P[FWG_P_C_LIFT_DELTA_E];
    d.CDp = P[FWG_P_C_D_P]; d.kInd = (T)1 / (pi * P[FWG_P_E] * P[FWG_P_AR]);
    d.CDb1 = P[FWG_P_C_D_BETA1]; d.CDb2 = P[FWG_P_C_D_BETA2];
    d.cDq = P[FWG_P_C_D_Q] * P[FWG_P_C]; d.CDde = P[FWG_P_C_D_DELTA_E];
    d.Cm0 = P[FWG_P_C_M_0]; d.Cma = P[FWG_P_C_M_ALPHA];
    d.cmq = P[FWG_P_C_M_Q] * P[FWG_P_B]; d.Cmde = P[FWG_P_C_M_DELTA_E]; d.Cmfp = P[FWG_P_C_M_FP];
    d.chord = P[FWG_P_C]; d.span = P[FWG_P_B];
    d.CY0 = P[FWG_P_C_Y_0]; d.CYb = P[FWG_P_C_Y_BETA];
    d.cYp = P[FWG_P_C_Y_P] * P[FWG_P_B]; d.cYr = P[FWG_P_C_Y_R] * P[FWG_P_B]; d.CYda = P[FWG_P_C_Y_DELTA_A];
    d.Cl0 = P[FWG_P_C_ROLL_0]; d.Clb = P[FWG_P_C_ROLL_BETA];
    d.clp = P[FWG_P_C_ROLL_P] * P[FWG_P_B]; d.clr = P[FWG_P_C_ROLL_R] * P[FWG_P_B]; d.Clda = P[FWG_P_C_ROLL_DELTA_A];
    d.Cn0 = P[FWG_P_C_N_0]; d.Cnb = P[FWG_P_C_N_BETA];
    d.cnp = P[FWG_P_C_N_P] * P[FWG_P_B]; d.cnr = P[FWG_P_C_N_R] * P[FWG_P_B]; d.Cnda = P[FWG_P_C_N_DELTA_A];
    d.kprop = (T)0.5 * rho * P[FWG_P_S_PROP] * P[FWG_P_C_PROP];
    d.kmotor = P[FWG_P_K_MOTOR];
    d.ktp = P[FWG_P_K_T_P] * P[FWG_P_K_OMEGA] * P[FWG_P_K_OMEGA];
}
struct DevFactor { int cls, type, src, fclass, shaping, window, has_max, value_is_timesteps; float sign, inv_scaling, max, value; };

// Static configuration.  Every member is a 32-bit int/float (no padding), so that a lowered instance can be frozen into
// a constexpr object (see "specialisation" in fwgym.hip) and the compiler folds every branch/constant driven by it.
struct DevCfg {
    // ---- simulator, pre-combined in double on the host
    float dt, h, half_h, h_sixth;
    int nsub, turbulence;
    float half_rho_S, mg, inv_mass, inv_Jy;
    float G1, G2, G3, G4, G5, G6, G7, G8;
    float M, Ma0;
    float CL0, CLa, cLq, CLde;
    float CDp, kInd, CDb1, CDb2, cDq, CDde;
    float Cm0, Cma, cmq, Cmde, Cmfp, chord, span;
    float CY0, CYb, cYp, cYr, CYda;
    float Cl0, Clb, clp, clr, Clda;
    float Cn0, Cnb, cnp, cnr, Cnda;
    float kprop, kmotor, ktp;
    float con_min[FWG_N_VARS], con_max[FWG_N_VARS];
    float val_min[FWG_N_VARS], val_max[FWG_N_VARS];
    unsigned con_mask;  // bit v set when variable v has a constraint
    float act_phi[2][4], act_travel[2], dot_max[2], act_ethr;  // exact actuator transition over one micro-step
    int act_per_half;
    float dryA[FWG_N_DRYDEN * FWG_N_DRYDEN], dryB[FWG_N_DRYDEN * 4], dryC[6 * FWG_N_DRYDEN];
    int turb_increment;   // the gust sample is the first difference of the filter outputs (kept in the simulator rows)
    int sim_keys;         // simulator.turbulence / turbulence_intensity are sampled per env at every reset: per-env gust gain
    int int_window;       // integration_window when integrator observations / int_error factors use it, else 0
    int has_int_obs;      // an observation entry of value "integrator" exists (its reset value depends on the PREVIOUS episode)
    // ---- gym side
    int steps_max, obs_length, obs_step, n_obs, obs_dim, obs_noise;
    float obs_noise_mean, obs_noise_std;
    DevObs obs[FWG_MAX_OBS];
    int scale_actions;
    float scale_low, scale_high, inv_scale_span;
    float act_to_low[3], act_to_high[3], inv_act_span[3];
    int has_action_bounds;
    float act_bound_min[3], act_bound_max[3];
    int n_targets, resample_every, streak_req, streak_min_count, on_success, goal_enabled, any_dynamic_target;
    DevTarget target[FWG_MAX_TARGETS];
    int reward_potential, step_fail_timesteps;
    float step_fail_value;
    int term_present[3];
    float term_weight[3];
    int n_factors;
    DevFactor factor[FWG_MAX_FACTORS];
    int metrics, auto_reset, use_cmd_ring, store_derived;
    int obs_log;   // rows per parity of the observation row log (0 = dense observation batch)
    float rise_low, rise_high;
    int model_n;   // > 0: simulator.model -- per-env force/moment constants (L.aero), re-sampled at every reset
    int randomize_scaling;   // reward.randomize_scaling -- per-env reward scalings (L.fscale), re-sampled at every reset
    // shape instances only (see merge_values below): non-zero in the frozen object of a shape instance -- its VALUE members are
    // not the ones to compute with; zero everywhere else (values and structure in one object).  Second word: reserved.
    unsigned values_elsewhere, reserved_;
    fwg_layout L;
};
// VALUE members are read through V(c): the object itself -- or, in a shape instance, the configuration in device memory, i.e.
// what the kernel's FIRST argument points at (every kernel that is instantiated per configuration takes the DevCfg pointer
// first).  The pointer is re-read from the kernel-argument segment here rather than handed down through every signature: a
// scalar load of a launch constant, and what is read through it is launch-constant too (constant address space: scalar loads
// at the point of use).  In a frozen or generic kernel the test folds / is one comparison and V(c) is c.
__device__ __forceinline__ const DevCfg& cfg_values(const DevCfg& c) {
#if defined(FWG_EMU) || !defined(__HIP_DEVICE_COMPILE__)
    return c;
#else
    if (c.values_elsewhere == 0u) return c;
    typedef const __attribute__((address_space(4))) unsigned long long* karg_ptr;
    typedef const __attribute__((address_space(4))) DevCfg* cfg_ptr;
    const unsigned long long first_arg = *(karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    return *(const DevCfg*)(cfg_ptr)first_arg;
#endif
}
#define V(c) cfg_values(c)
__host__ __device__ constexpr DevCfg as_shape(DevCfg d) { d.values_elsewhere = 1u; return d; }

// ---- "Shape" instances (fwgym.hip "Specialisation"): a kernel frozen on a configuration's STRUCTURE -- every integer member:
// counts, types, sources, flags, the arena layout -- that reads the configuration's VALUES -- every float member, and the few
// integers that are only ever compared (the time limit, the resampling period, the goal-window count) -- from memory.  The
// function below defines the split: it returns the frozen instance with the value members taken from `r`.  A configuration runs
// on a shape instance iff merging it into the instance reproduces it bit for bit (match_spec); the kernel itself reads the
// configuration from memory and is TOLD that its structure words are the instance's (assume_structure, fwgym.hip) -- computing
// with the merged object instead loads every value up front and keeps them all alive: 45 000 spilled registers, measured.
// (Keep in step with DevCfg / DevObs / DevTarget / DevFactor: the size check below fails when a member is added.)
template <class R>
__host__ __device__ constexpr DevCfg merge_values(DevCfg m, const R* r) {
    static_assert(sizeof(DevObs) == 24 && sizeof(DevTarget) == 24 && sizeof(DevFactor) == 48, "merge_values: member list out of date");
    static_assert(sizeof(DevCfg) == 4 * (4 + 2 + 49 + 4 * FWG_N_VARS + 1 + 13 + 1 + FWG_N_DRYDEN * FWG_N_DRYDEN + FWG_N_DRYDEN * 4 + 6 * FWG_N_DRYDEN + 4 +
                                         6 + 2 + 6 * FWG_MAX_OBS + 1 + 3 + 9 + 1 + 6 + 7 + 6 * FWG_MAX_TARGETS + 2 + 1 + 3 + 3 + 1 + 12 * FWG_MAX_FACTORS +
                                         4 + 1 + 2 + 1 + 1 + 2) + sizeof(fwg_layout),
                  "merge_values: member list out of date");
    m.dt = r->dt; m.h = r->h; m.half_h = r->half_h; m.h_sixth = r->h_sixth;
#define FWG_MERGE_AERO(n) m.n = r->n;
    FWG_AERO_LIST(FWG_MERGE_AERO)
#undef FWG_MERGE_AERO
    for (int i = 0; i < FWG_N_VARS; ++i) { m.con_min[i] = r->con_min[i]; m.con_max[i] = r->con_max[i]; m.val_min[i] = r->val_min[i]; m.val_max[i] = r->val_max[i]; }
    for (int a = 0; a < 2; ++a) {
        for (int i = 0; i < 4; ++i) m.act_phi[a][i] = r->act_phi[a][i];
        m.act_travel[a] = r->act_travel[a]; m.dot_max[a] = r->dot_max[a];
    }
    m.act_ethr = r->act_ethr;
    for (int i = 0; i < FWG_N_DRYDEN * FWG_N_DRYDEN; ++i) m.dryA[i] = r->dryA[i];
    for (int i = 0; i < FWG_N_DRYDEN * 4; ++i) m.dryB[i] = r->dryB[i];
    for (int i = 0; i < 6 * FWG_N_DRYDEN; ++i) m.dryC[i] = r->dryC[i];
    m.steps_max = r->steps_max;
    m.obs_noise_mean = r->obs_noise_mean; m.obs_noise_std = r->obs_noise_std;
    for (int j = 0; j < FWG_MAX_OBS; ++j) { m.obs[j].mean = r->obs[j].mean; m.obs[j].inv_var = r->obs[j].inv_var; }
    m.scale_low = r->scale_low; m.scale_high = r->scale_high; m.inv_scale_span = r->inv_scale_span;
    for (int i = 0; i < 3; ++i) {
        m.act_to_low[i] = r->act_to_low[i]; m.act_to_high[i] = r->act_to_high[i]; m.inv_act_span[i] = r->inv_act_span[i];
        m.act_bound_min[i] = r->act_bound_min[i]; m.act_bound_max[i] = r->act_bound_max[i];
        m.term_weight[i] = r->term_weight[i];
    }
    m.resample_every = r->resample_every; m.streak_min_count = r->streak_min_count;
    for (int k = 0; k < FWG_MAX_TARGETS; ++k) m.target[k].bound = r->target[k].bound;
    m.step_fail_value = r->step_fail_value;
    for (int f = 0; f < FWG_MAX_FACTORS; ++f) {
        m.factor[f].sign = r->factor[f].sign; m.factor[f].inv_scaling = r->factor[f].inv_scaling;
        m.factor[f].max = r->factor[f].max; m.factor[f].value = r->factor[f].value;
    }
    m.rise_low = r->rise_low; m.rise_high = r->rise_high;
    return m;
}

// ---------------------------------------------------------------------------------------------------------------------
// math helpers (fp32, gfx950)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fclampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
__device__ __forceinline__ float fsignf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
// keeps a product from being fused into a later addition (-ffp-contract=fast works across statements): values that two
// code paths must reproduce bit for bit are rounded here
__device__ __forceinline__ float rounded(float x) {
#ifndef FWG_EMU
    asm volatile("" : "+v"(x));
#endif
    return x;
}
// a / b by v_rcp_f32 (1 ulp) for the episodic metrics: the IEEE division sequence is ~10x the instructions, in a branch whose
// length is the step time of the whole launch whenever any episode ends
__device__ __forceinline__ float fast_div(float a, float b) {
#ifdef FWG_EMU
    return a / b;
#else
    return a * __builtin_amdgcn_rcpf(b);
#endif
}

// atan2 for the kinematics (angle of attack, sideslip, Euler angles): odd minimax polynomial of degree 17 on [0,1]
// (max abs error 1.1e-7 in fp32, fitted offline) + octant folding; ~20 VALU instructions instead of the library's ~45.
__device__ __forceinline__ float fast_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mx > 0.f ? mn * frcp(mx) : 0.f;
    const float s = a * a;
    float p = 0.0024566026404500008f;
    p = p * s - 0.014400825835764408f;
    p = p * s + 0.03978026658296585f;
    p = p * s - 0.07234764844179153f;
    p = p * s + 0.1049889475107193f;
    p = p * s - 0.14161212742328644f;
    p = p * s + 0.19985903799533844f;
    p = p * s - 0.33332598209381104f;
    p = p * s + 0.9999998807907104f;
    float r = p * a;
    r = ay > ax ? 1.57079632679489661923f - r : r;
    r = x < 0.f ? 3.14159265358979323846f - r : r;
    return copysignf(r, y);
}

// ---- pairs of floats as ONE value: on the device they live in an aligned register pair and `a * b + c` is a single
// v_pk_fma_f32 (v_pk_mul_f32, v_pk_add_f32) -- two results per issue slot.  A wave issues at most one vector instruction every
// ~5 cycles however much is independent (tools/ub_valu.hip), and the physics wave's ~1 900 dependent instructions ARE the
// length of an env step: wherever two lanes of arithmetic have the same shape they are written on pairs.  (GCC-style vector:
// the host emulation build compiles the same source.)
typedef float f2 __attribute__((vector_size(8)));
__host__ __device__ __forceinline__ f2 mk2(float a, float b) { f2 r = {a, b}; return r; }
__host__ __device__ __forceinline__ f2 splat2(float a) { f2 r = {a, a}; return r; }
__device__ __forceinline__ f2 fmax2(f2 a, f2 b) { return mk2(fmaxf(a[0], b[0]), fmaxf(a[1], b[1])); }
__device__ __forceinline__ f2 fmin2(f2 a, f2 b) { return mk2(fminf(a[0], b[0]), fminf(a[1], b[1])); }
__device__ __forceinline__ f2 fabs2(f2 a) { return mk2(fabsf(a[0]), fabsf(a[1])); }
// two atan2 side by side: same octant folding as fast_atan2, the polynomial on pairs
__device__ __forceinline__ f2 fast_atan2x2(f2 y, f2 x) {
    const f2 ax = fabs2(x), ay = fabs2(y);
    const f2 mx = fmax2(ax, ay), mn = fmin2(ax, ay);
    const f2 a = mk2(mx[0] > 0.f ? mn[0] * frcp(mx[0]) : 0.f, mx[1] > 0.f ? mn[1] * frcp(mx[1]) : 0.f);
    const f2 s = a * a;
    f2 p = splat2(0.0024566026404500008f);
    p = p * s - splat2(0.014400825835764408f);
    p = p * s + splat2(0.03978026658296585f);
    p = p * s - splat2(0.07234764844179153f);
    p = p * s + splat2(0.1049889475107193f);
    p = p * s - splat2(0.14161212742328644f);
    p = p * s + splat2(0.19985903799533844f);
    p = p * s - splat2(0.33332598209381104f);
    p = p * s + splat2(0.9999998807907104f);
    const f2 r = p * a;
    float r0 = r[0], r1 = r[1];
    r0 = ay[0] > ax[0] ? 1.57079632679489661923f - r0 : r0;
    r1 = ay[1] > ax[1] ? 1.57079632679489661923f - r1 : r1;
    r0 = x[0] < 0.f ? 3.14159265358979323846f - r0 : r0;
    r1 = x[1] < 0.f ? 3.14159265358979323846f - r1 : r1;
    return mk2(copysignf(r0, y[0]), copysignf(r1, y[1]));
}

#define FWG_PI 3.14159265358979323846f
#define FWG_TWO_PI 6.28318530717958647692f
#define FWG_INV_TWO_PI 0.15915494309189533577f

// Python's float modulo by 2*pi followed by the fold of fixed_wing.py:910-912 (value - target for wrap states)
__device__ __forceinline__ float angle_dist(float target, float value) {
    float x = value - target + FWG_PI;
    float m = x - FWG_TWO_PI * floorf(x * FWG_INV_TWO_PI);
    // guard the fp32 edge where rounding leaves m outside [0, 2pi)
    m = m < 0.f ? m + FWG_TWO_PI : (m >= FWG_TWO_PI ? m - FWG_TWO_PI : m);
    float d = m - FWG_PI;
    return d < -FWG_PI ? d + FWG_TWO_PI : d;
}

// sign(x)*(|x| % pi - pi) for |x| > pi (fixed_wing.py:988-989)
__device__ __forceinline__ float wrap_target(float x) {
    float ax = fabsf(x);
    if (ax > FWG_PI) {
        float m = ax - FWG_PI * floorf(ax * (1.0f / FWG_PI));
        x = fsignf(x) * (m - FWG_PI);
    }
    return x;
}

// ---------------------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al.), ctr = (env_id, counter, sub, stream), key = seed
// ---------------------------------------------------------------------------------------------------------------------
struct u4 { unsigned x, y, z, w; };
__device__ __forceinline__ u4 philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// ---------------------------------------------------------------------------------------------------------------------
// async HBM -> LDS streaming (global_load_lds)
// ---------------------------------------------------------------------------------------------------------------------
typedef float fwg_v4f __attribute__((ext_vector_type(4)));   // native 16-byte vector (for __builtin_nontemporal_store)
typedef __attribute__((address_space(1))) const void* fwg_gptr;
typedef __attribute__((address_space(3))) void* fwg_lptr;

// async HBM -> LDS copy of one 16-byte group per lane (1 KiB per wave, landing as [lane][4]) -- global_load_lds_dwordx4
#ifndef FWG_RING_AUX
#define FWG_RING_AUX 0
#endif
__device__ __forceinline__ void dma_group(const float4* src_lane_ptr, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((fwg_gptr)src_lane_ptr, (fwg_lptr)lds_dst, 16, 0, FWG_RING_AUX);
}
#ifndef FWG_LAG_AUX
#define FWG_LAG_AUX 2   /* nt: measured -1.0 us per C3 step (65 536 envs) against 0, no gain for the rings or state rows */
#endif
// same, for data read once per launch (lagged observation rows): cache-policy bits as a build-time experiment knob
__device__ __forceinline__ void dma_group_once(const float4* src_lane_ptr, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((fwg_gptr)src_lane_ptr, (fwg_lptr)lds_dst, 16, 0, FWG_LAG_AUX);
}
// the compiler does not order LDS reads behind an in-flight global_load_lds: drain the vector-memory counter by hand
#ifndef FWG_DMA_DRAIN
#define FWG_DMA_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
// workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope release + acquire around the
// s_barrier, which on gfx950 drains every global load and store in flight (s_waitcnt vmcnt(0)); the two waves of k_step2
// exchange data through LDS alone, and requests issued before the barrier (prefetches for the episode-end branch, the
// bookkeeping stores) are meant to stay in flight across it
#ifdef FWG_EMU
#define FWG_BLOCK_SYNC_LDS() __syncthreads()
#else
#define FWG_BLOCK_SYNC_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
// one-way hand-shake through an LDS word (k_step2: the physics wave raises it to 1 when it has read its hand-off areas and to
// 2 when its rows are in memory; the gym wave waits only where it is about to re-use the former or overwrite the latter --
// an s_barrier would make the physics wave wait as well, and the gym wave wait for store acknowledgements it does not need)
// (a volatile access through a generic pointer would be a FLAT instruction followed by s_waitcnt vmcnt(0) -- the wave would sit
// through the acknowledgement of every store it has in flight; ds_read / ds_write on the LDS offset (the low half of the
// generic address) touch the LDS queue only)
#ifdef FWG_EMU
#define FWG_FLAG_WAIT(p, level) do { while (*reinterpret_cast<volatile const unsigned*>(p) < (unsigned)(level)) emu_yield(); } while (0)
#define FWG_FLAG_RAISE(p, level) do { *reinterpret_cast<volatile unsigned*>(p) = (unsigned)(level); } while (0)
#else
__device__ __forceinline__ unsigned fwg_lds_peek(const void* p) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
    return v;
}
#define FWG_FLAG_WAIT(p, level)                                                            \
    do {                                                                                   \
        while (fwg_lds_peek(p) < (unsigned)(level)) __builtin_amdgcn_s_sleep(1);           \
    } while (0)
#define FWG_FLAG_RAISE(p, level) asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(size_t)(p)), "v"((unsigned)(level)) : "memory")
#endif
// pins values in front of a wait: an empty asm statement the values pass through.  The compiler orders `asm volatile` statements
// among themselves, but arithmetic that is not needed until after a polling loop is SUNK below it (the loop is a basic block of
// its own) -- the physics wave's first stage ended up behind its wait for the partner's actuator message.  Values that passed
// through fwg_pin() exist before the next asm volatile statement executes.
#ifdef FWG_EMU
template <class... T> __device__ __forceinline__ void fwg_pin(T&...) {}
#else
__device__ __forceinline__ void fwg_pin() {}
template <class T, class... R> __device__ __forceinline__ void fwg_pin(T& x, R&... rest) {
    asm volatile("" : "+v"(x));
    fwg_pin(rest...);
}
#endif
// the same with a memory clobber: memory operations written after it are issued after the value exists (k_step2, gym wave: its
// bulk of row requests goes out when its first three have come back -- the CU's vector-memory path serves requests in order, 1 KiB
// per 16 cycles, and the physics wave's rows, requested at the same moment, are the ones the step's length hangs on)
template <class T> __device__ __forceinline__ void fwg_pin_mem(T& x) {
#ifndef FWG_EMU
    asm volatile("" : "+v"(x)::"memory");
#endif
}
// wave priority for the arbitration of a SIMD's issue slots between its two co-resident waves (priority first, then age: measured
// with tools/timeline.py, the younger of two busy waves on a SIMD runs at 50-75 % of its speed alone: the first of a CU's four
// workgroups finished its step in 13.2k ticks, the last in 17.7k -- and the launch lasts as long as its slowest workgroup).  Levels
// of the phases of k_step2's two waves (same-box A/B, C3, steady state, profiles/r05_priorities.txt): the gym wave's actuator
// work -- which a physics wave waits for -- above everything, the physics wave's integration above the gym wave's other
// pre-hand-over work, the gym wave's chain after the hand-over above the physics wave's tail: 12.7 -> 12.0 us per step.
#if defined(FWG_EMU) || !defined(__HIP_DEVICE_COMPILE__)
#define FWG_SETPRIO(n) do { } while (0)
#else
#define FWG_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
#ifndef FWG_PRIO_PHYS_STAGES
#define FWG_PRIO_PHYS_STAGES 2
#endif
#ifndef FWG_PRIO_PHYS_TAIL
#define FWG_PRIO_PHYS_TAIL 0
#endif
#ifndef FWG_PRIO_GYM_EARLY
#define FWG_PRIO_GYM_EARLY 3
#endif
#ifndef FWG_PRIO_GYM_PRE
#define FWG_PRIO_GYM_PRE 1
#endif
#ifndef FWG_PRIO_GYM_POST
#define FWG_PRIO_GYM_POST 3
#endif
#ifndef FWG_PRIO_GYM_END   // (the gym wave of a group that hosts an episode end, from the moment it knows)
#define FWG_PRIO_GYM_END 3
#endif
// One-way MESSAGES between the waves of a workgroup: 16-byte groups per lane in LDS whose last word is a tag.  LDS executes a
// wave's accesses in order, so a message of several groups is written data first, tagged group last; the reader polls the
// tagged group and then reads the rest.  No barrier: the writer never waits, the reader waits only for what it needs
// (tools/ub_handoff.hip: ~290-330 ticks from the write to the value in the reader's registers, against ~480 with a separate
// flag word -- and an s_barrier makes BOTH waves wait).  Tags are cleared by their writer at kernel entry, before the
// workgroup's entry barrier (a previous workgroup's leftovers in the same LDS cannot be mistaken for a message).
#define FWG_TAG_RAW 0x5A000007u      /* physics -> gym: the raw action and the actuator states the step starts from (the gym wave requests no
                                        row before the physics wave's have landed: the CU's vector-memory path serves 64 B per clock, in order) */
#define FWG_TAG_ACTS 0x5A000001u     /* gym -> physics: actuator states at t + h/2 and t + h */
#define FWG_TAG_STATE 0x5A000002u    /* physics -> gym: candidate state + Euler-angle arguments */
#define FWG_TAG_RESULT 0x5B000000u   /* physics -> gym: Va, alpha, beta | failure code in the low byte */
#define FWG_TAG_OLD 0x5A000003u      /* physics -> gym: (failed step) the last valid state and its derived values */
#define FWG_TAG_TAIL 0x5A000004u     /* gym -> physics: what the physics wave's tail work needs (step index, early / install flags) */
#ifdef FWG_EMU
__device__ __forceinline__ void fwg_msg_put(float* p, float a, float b, float c, unsigned tag) {
    p[0] = a; p[1] = b; p[2] = c; reinterpret_cast<unsigned*>(p)[3] = tag;
}
// waits (where `need`) until the group carries a tag with (tag & mask) == want; returns the group
__device__ __forceinline__ float4 fwg_msg_take(const float* p, unsigned want, unsigned mask = 0xFFFFFFFFu, bool need = true) {
    while (need && (reinterpret_cast<const volatile unsigned*>(p)[3] & mask) != want) emu_yield();
    return make_float4(p[0], p[1], p[2], p[3]);
}
#else
__device__ __forceinline__ void fwg_msg_put(float* p, float a, float b, float c, unsigned tag) {
    const fwg_v4f q = {a, b, c, __uint_as_float(tag)};
    asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)p), "v"(q) : "memory");
}
__device__ __forceinline__ float4 fwg_msg_take(const float* p, unsigned want, unsigned mask = 0xFFFFFFFFu, bool need = true) {
    fwg_v4f q;
    for (;;) {
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"((unsigned)(size_t)p) : "memory");
        if (__ballot(need && (__float_as_uint(q.w) & mask) != want) == 0ull) break;
        __builtin_amdgcn_s_sleep(1);
    }
    return make_float4(q.x, q.y, q.z, q.w);
}
#endif
// ordering of LDS traffic between the lanes of ONE wave (k_step2: the other wave of the workgroup is not involved)
#ifdef FWG_EMU
#define FWG_WAVE_SYNC() emu_wave_sync()
#else
#define FWG_WAVE_SYNC()                                          \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   \
    } while (0)
#endif
// (the lanes of a wave run in lock step on the device; the emulation's lane fibers need the rendezvous spelled out)
#ifdef FWG_EMU
#define FWG_EMU_WAVE_SYNC() emu_wave_sync()
#else
#define FWG_EMU_WAVE_SYNC() do {} while (0)
#endif
__device__ __forceinline__ void dma_wait() {
    FWG_DMA_DRAIN();
    __syncthreads();
}

