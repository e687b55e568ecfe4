// fwgym_env.h -- per-lane episode state, gym-side logic (targets, goal streak, reward, observation rows, metrics)
// and the reset routine shared by the step and reset kernels (device, fp32).
// Specification: oracle/gym_restated.py (float64), which is pinned against the reference's gym_fixed_wing/fixed_wing.py.
#pragma once
#include "fwgym_physics.h"

// kernel arguments shared by the step and reset kernels
struct KArgs {
    float* S;                 // state arena [rows][N]
    long N;
    long env_base;            // global index of env 0 of this handle (RNG streams)
    const float* actions;     // [N][3]
    float* obs;               // [N][obs_dim]
    float* rew;               // [N]
    uint8_t* done;            // [N]
    uint8_t* term;            // [N]
    float* term_obs;          // nullable [N][obs_dim]
    float* metrics;           // nullable [FWG_N_METRICS][N]
    float* tgt_out;           // nullable [N][n_targets]
    float* reduce;            // [FWG_N_REDUCE]
    const uint8_t* mask;      // reset: nullable [N]
    const float* init_state;  // reset: nullable [FWG_N_RESET_VARS][N]
    const float* init_target; // reset: nullable [n_targets][N]
    unsigned seed_lo, seed_hi;
    int slot_act, slot_end, slot_lag, bit_goal;  // ring positions of the CURRENT global step
    int lag_slots[FWG_MAX_ROWS];                 // ring slot holding the row pushed r*obs_step steps ago
};

// LDS carve (in floats) for one 64-lane block: the action windows (streamed in by global_load_lds) and, for the
// generic (non-specialised) kernel only, the per-lane scratch tables that config-driven indices address
struct LdsMap { int aring, cring, lag, stage, tab, obs, total; };
#define FWG_TAB_TGT FWG_N_VARS               // table rows: simulator variables | targets | target errors
#define FWG_TAB_ERR (FWG_N_VARS + FWG_MAX_TARGETS)
#define FWG_TAB_ROWS (FWG_N_VARS + 2 * FWG_MAX_TARGETS)
// row stride (words) of the output staging: records of 4q words with q odd are written/read with conflict-free 16-byte
// LDS accesses; any other size falls back to an odd stride and 4-byte accesses
__host__ __device__ inline bool obs_vec4(int obs_dim) { return (obs_dim % 4 == 0) && ((obs_dim / 4) % 2 == 1); }
__host__ __device__ inline int obs_stage_stride(int obs_dim) { return obs_vec4(obs_dim) ? obs_dim : (obs_dim | 1); }
__host__ __device__ inline LdsMap lds_map(int obs_dim, int n_obs, int window, int use_cmd_ring, bool generic) {
    LdsMap m;
    int o = 0;
    m.aring = o; o += window * 3 * FWG_WAVE;
    m.cring = o; o += (use_cmd_ring ? window * 3 * FWG_WAVE : 0);
    m.lag = o; o += (obs_dim - n_obs) * FWG_WAVE;          // lagged rows streamed in as SoA rows [entry][lane]
    m.stage = o; o += FWG_WAVE * obs_stage_stride(obs_dim);  // [lane][obs_dim] staging of the output records
    m.tab = o; o += generic ? FWG_TAB_ROWS * FWG_WAVE : 0;
    m.obs = o; o += generic ? obs_dim * FWG_WAVE : 0;
    m.total = (o + 3) & ~3;
    return m;
}

// Per-lane tables behind one interface.  In a specialised kernel every index is a compile-time constant after
// unrolling, so the register-array flavour costs nothing; the generic kernel indexes lane-private LDS columns
// ([entry][lane], conflict-free) with wave-uniform run-time indices.
template <int ROWS> struct RegTable {
    float v[ROWS];
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void put(int i, float x) { v[i] = x; }
};
struct LdsTable {
    float* p;
    __device__ __forceinline__ float get(int i) const { return p[i * FWG_WAVE]; }
    __device__ __forceinline__ void put(int i, float x) { p[i * FWG_WAVE] = x; }
};

struct Env {
    float y[NY];
    float wind[3];
    float dry[FWG_N_DRYDEN];
    Derived d;
    float tgt[FWG_MAX_TARGETS];
    float tprop[FWG_MAX_TARGETS][4];  // slope|amplitude, period, phase, bias
    unsigned steps, sft, flags, episode;
    float psh[3];
    float pcmd[3];
    unsigned gword[4];   // the 32-bit word of each goal window that holds this step's bit position
    unsigned wcnt;       // ones inside each window, 4 x 8 bit (windows hold success_streak_req <= 128 flags)
    unsigned gcnt[2];    // cumulative ones per window since reset, 4 x 16 bit
    float e0[3], esum[3], eabs[3], emin[3], emax[3];
    unsigned rise[3];
    unsigned settle[2];
    float perr[3];
    float sdcmd;
};

// element (row r, env e) of the SoA arena; 32-bit indices (fwg_create guarantees rows*N < 2^31) keep the address
// arithmetic to one scalar multiply + one vector add per access
#define ROW(S, N, r, e) ((S)[(unsigned)(r) * (unsigned)(N) + (unsigned)(e)])
// entry j of the record of env e in slot `slot` of a ring that starts at arena row `base` (SoA rows [slot][entry][env])
#define RING(S, N, base, slot, width, e, j) ROW(S, N, (base) + (slot) * (width) + (j), e)

// The load is split like the write-back: the simulator state is requested first (the integration starts as soon as it
// arrives); the bookkeeping rows and the HBM->LDS streams are requested afterwards and land while the integration runs.
template <bool TURB>
__device__ __forceinline__ void load_sim(const DevCfg& c, const float* __restrict__ S, long N, long e, Env& E) {
    const fwg_layout& L = c.L;
#pragma unroll
    for (int i = 0; i < NY; ++i) E.y[i] = ROW(S, N, L.phys + i, e);
#pragma unroll
    for (int i = 0; i < 3; ++i) E.wind[i] = ROW(S, N, L.wind + i, e);
    if (TURB) {
#pragma unroll
        for (int i = 0; i < FWG_N_DRYDEN; ++i) E.dry[i] = ROW(S, N, L.dryden + i, e);
    }
}

__device__ __forceinline__ void load_gym(const DevCfg& c, const float* __restrict__ S, long N, long e, Env& E, int goal_bit) {
    const fwg_layout& L = c.L;
    const unsigned* U = reinterpret_cast<const unsigned*>(S);
    // the derived rows (roll pitch yaw Va alpha beta) are write-only for the kernels: after a failed step they are
    // recomputed from the restored state
#pragma unroll
    for (int k = 0; k < FWG_MAX_TARGETS; ++k) E.tgt[k] = ROW(S, N, L.target + k, e);
    if (c.any_dynamic_target) {
#pragma unroll
        for (int k = 0; k < FWG_MAX_TARGETS; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) E.tprop[k][j] = ROW(S, N, L.target + 3 + k * 4 + j, e);
    }
    E.steps = ROW(U, N, L.counters + 0, e); E.sft = ROW(U, N, L.counters + 1, e);
    E.flags = ROW(U, N, L.counters + 2, e); E.episode = ROW(U, N, L.counters + 3, e);
    if (c.reward_potential) {
#pragma unroll
        for (int i = 0; i < 3; ++i) E.psh[i] = ROW(S, N, L.prev_shaping + i, e);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) E.pcmd[i] = ROW(S, N, L.prev_cmd + i, e);
    if (c.goal_enabled) {  // only the word that receives this step's flag is touched; the window counts are kept apart
#pragma unroll
        for (int r = 0; r < 4; ++r) E.gword[r] = ROW(U, N, L.goal_ring + r * 4 + (goal_bit >> 5), e);
        E.wcnt = ROW(U, N, L.goal_count + 0, e);
        E.gcnt[0] = ROW(U, N, L.goal_count + 1, e); E.gcnt[1] = ROW(U, N, L.goal_count + 2, e);
    }
    if (c.metrics) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            E.e0[k] = ROW(S, N, L.met + 0 + k, e); E.esum[k] = ROW(S, N, L.met + 3 + k, e);
            E.eabs[k] = ROW(S, N, L.met + 6 + k, e); E.emin[k] = ROW(S, N, L.met + 9 + k, e);
            E.emax[k] = ROW(S, N, L.met + 12 + k, e); E.rise[k] = ROW(U, N, L.met + 15 + k, e);
            E.perr[k] = ROW(S, N, L.met + 20 + k, e);
        }
        E.settle[0] = ROW(U, N, L.met + 18, e); E.settle[1] = ROW(U, N, L.met + 19, e);
        E.sdcmd = ROW(S, N, L.met + 23, e);
    }
}

// The write-back is split so that each part is issued as soon as its values are final: the simulator state right after
// the integration, the bookkeeping after the gym logic -- the store traffic then overlaps the remaining computation
// instead of forming one burst at the end of the kernel.
template <bool TURB>
__device__ __forceinline__ void store_sim(const DevCfg& c, float* __restrict__ S, long N, long e, const Env& E, bool with_wind) {
    const fwg_layout& L = c.L;
#pragma unroll
    for (int i = 0; i < NY; ++i) ROW(S, N, L.phys + i, e) = E.y[i];
    if (with_wind) {  // the steady wind only changes at reset
#pragma unroll
        for (int i = 0; i < 3; ++i) ROW(S, N, L.wind + i, e) = E.wind[i];
    }
    if (TURB) {
#pragma unroll
        for (int i = 0; i < FWG_N_DRYDEN; ++i) ROW(S, N, L.dryden + i, e) = E.dry[i];
    }
    ROW(S, N, L.derived + 0, e) = E.d.roll; ROW(S, N, L.derived + 1, e) = E.d.pitch; ROW(S, N, L.derived + 2, e) = E.d.yaw;
    ROW(S, N, L.derived + 3, e) = E.d.Va; ROW(S, N, L.derived + 4, e) = E.d.alpha; ROW(S, N, L.derived + 5, e) = E.d.beta;
}

__device__ __forceinline__ void store_gym(const DevCfg& c, float* __restrict__ S, long N, long e, const Env& E, bool at_reset,
                                          int goal_bit) {
    const fwg_layout& L = c.L;
    unsigned* U = reinterpret_cast<unsigned*>(S);
#pragma unroll
    for (int k = 0; k < FWG_MAX_TARGETS; ++k) ROW(S, N, L.target + k, e) = E.tgt[k];
    if (c.any_dynamic_target) {
#pragma unroll
        for (int k = 0; k < FWG_MAX_TARGETS; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) ROW(S, N, L.target + 3 + k * 4 + j, e) = E.tprop[k][j];
    }
    ROW(U, N, L.counters + 0, e) = E.steps; ROW(U, N, L.counters + 1, e) = E.sft;
    ROW(U, N, L.counters + 2, e) = E.flags;
    if (at_reset) ROW(U, N, L.counters + 3, e) = E.episode;  // the episode counter only changes at reset
    if (c.reward_potential) {
#pragma unroll
        for (int i = 0; i < 3; ++i) ROW(S, N, L.prev_shaping + i, e) = E.psh[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) ROW(S, N, L.prev_cmd + i, e) = E.pcmd[i];
    if (c.goal_enabled) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ROW(U, N, L.goal_ring + r * 4 + (goal_bit >> 5), e) = E.gword[r];
        ROW(U, N, L.goal_count + 0, e) = E.wcnt;
        ROW(U, N, L.goal_count + 1, e) = E.gcnt[0]; ROW(U, N, L.goal_count + 2, e) = E.gcnt[1];
    }
    if (c.metrics) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (at_reset) ROW(S, N, L.met + 0 + k, e) = E.e0[k];  // the initial error is fixed for the episode
            ROW(S, N, L.met + 3 + k, e) = E.esum[k];
            ROW(S, N, L.met + 6 + k, e) = E.eabs[k]; ROW(S, N, L.met + 9 + k, e) = E.emin[k];
            ROW(S, N, L.met + 12 + k, e) = E.emax[k]; ROW(U, N, L.met + 15 + k, e) = E.rise[k];
            ROW(S, N, L.met + 20 + k, e) = E.perr[k];
        }
        ROW(U, N, L.met + 18, e) = E.settle[0]; ROW(U, N, L.met + 19, e) = E.settle[1];
        ROW(S, N, L.met + 23, e) = E.sdcmd;
    }
}

// simulator variable table (index = fwg_var) so that config-driven indices can address it
template <class TAB>
__device__ __forceinline__ void fill_vars(const Env& E, TAB& T) {
    const float v[FWG_N_VARS] = {E.d.roll, E.d.pitch, E.d.yaw, E.y[4], E.y[5], E.y[6], E.y[7], E.y[8], E.y[9],
                                 E.y[10], E.y[11], E.y[12], E.d.Va, E.d.alpha, E.d.beta,
                                 0.5f * (E.y[13] + E.y[14]), 0.5f * (E.y[14] - E.y[13]), E.y[15],
                                 E.wind[0], E.wind[1], E.wind[2], E.y[13], E.y[14]};
#pragma unroll
    for (int i = 0; i < FWG_N_VARS; ++i) T.put(i, v[i]);
}

// _get_error (fixed_wing.py:890-900): wrap states value-target folded, others target-value
__device__ __forceinline__ float target_error(const DevTarget& t, float target, float value) {
    return t.wrap ? angle_dist(target, value) : target - value;
}

__device__ __forceinline__ unsigned pack16_get(const unsigned (&p)[2], int i) { return (p[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu; }
__device__ __forceinline__ void pack16_set(unsigned (&p)[2], int i, unsigned v) {
    const int sh = (i & 1) * 16;
    p[i >> 1] = (p[i >> 1] & ~(0xFFFFu << sh)) | ((v & 0xFFFFu) << sh);
}

// goal flags of the current state against the current targets (fixed_wing.py:916-931); bit k = target k, bit 3 = all
__device__ __forceinline__ unsigned goal_flags(const DevCfg& c, const float (&err)[3]) {
    unsigned g = 0;
    bool all = true;
#pragma unroll
    for (int k = 0; k < FWG_MAX_TARGETS; ++k) {
        if (k < c.n_targets && c.target[k].has_bound) {
            const bool ok = fabsf(err[k]) <= c.target[k].bound;
            g |= ok ? (1u << k) : 0u;
            all = all && ok;
        }
    }
    return g | (all ? 8u : 0u);
}

// Push the goal flags of one record into the four windows (target0..2, all).  A window is a ring of
// success_streak_req bits addressed by the global step counter, so the bit being overwritten is exactly the one that
// leaves the window: the ones-count of the window is updated incrementally (no popcount over the ring), the cumulative
// count feeds success_time_frac, and the metric settling index latches the first record at which a full window
// satisfies the fraction (fixed_wing.py:1116-1128).
__device__ __forceinline__ unsigned window_count(const Env& E, int r) { return (E.wcnt >> (8 * r)) & 0xFFu; }
__device__ __forceinline__ void goal_push(const DevCfg& c, Env& E, unsigned g, int bit, unsigned rec_index) {
    const unsigned n_rec = rec_index + 1;
    const unsigned m = 1u << (bit & 31);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool present = (r == 3) || (r < c.n_targets && c.target[r < 3 ? r : 0].has_bound);
        if (present) {
            const unsigned f = (g >> r) & 1u;
            const unsigned old = (E.gword[r] & m) ? 1u : 0u;
            E.gword[r] = f ? (E.gword[r] | m) : (E.gword[r] & ~m);
            E.wcnt += (f - old) << (8 * r);     // per-byte add/subtract; each byte stays within [0, 128]
            E.gcnt[r >> 1] += f << (16 * (r & 1));
            if (c.metrics && pack16_get(E.settle, r) == 0xFFFFu && n_rec >= (unsigned)c.streak_req &&
                window_count(E, r) >= (unsigned)c.streak_min_count)
                pack16_set(E.settle, r, rec_index);
        }
    }
}

// sample_target (fixed_wing.py:461-521); `given` (nullable) holds explicit targets for reset(target=...)
template <class TAB>
__device__ __forceinline__ void sample_targets(const DevCfg& c, const DynCfg& dc, const KArgs& A, long e, Env& E,
                                               const TAB& T, const float* given) {
    const unsigned env_id = (unsigned)(A.env_base + e);
    const unsigned resample = E.flags >> FWG_FLAG_RESAMPLE_SHIFT;
    E.sft = 0;
#pragma unroll
    for (int k = 0; k < FWG_MAX_TARGETS; ++k) {
        if (k >= c.n_targets) continue;
        const DevTarget& t = c.target[k];
        const DynTarget& r = dc.target[k];
        const u4 b = philox4x32(env_id, E.episode, resample, FWG_STREAM_RESET_TARGET + 256u * k, A.seed_lo, A.seed_hi);
        float low = r.low, high = r.high;
        if (t.has_delta) {
            const float x = T.get(t.var);
            low = fmaxf(low, x - r.delta);
            high = fmaxf(fminf(high, x + r.delta), low);
        }
        float v = low + (high - low) * u01(b.x);
        int cls = t.cls;
        if (given != nullptr) {
            const float gv = given[(long)k * A.N + e];
            if (gv == gv) {  // explicit target: forces class constant unless compensate (fixed_wing.py:311-315)
                v = gv;
                if (cls != FWG_TGT_COMPENSATE) cls = FWG_TGT_CONSTANT;
            }
        }
        if (c.any_dynamic_target) {
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
            if (cls == FWG_TGT_LINEAR) {
                p0 = r.slope_low + (r.slope_high - r.slope_low) * u01(b.y);
                if (u01(b.z) < 0.5f) p0 = -p0;
            } else if (cls == FWG_TGT_SINUSOIDAL) {
                p0 = r.amp_low + (r.amp_high - r.amp_low) * u01(b.y);
                p1 = r.period_low + (r.period_high - r.period_low) * u01(b.z);
                p2 = (FWG_TWO_PI * u01(b.w)) / (FWG_TWO_PI / p1);
                p3 = v - p0 * sinf(FWG_TWO_PI / p1 * ((float)E.steps + p2));
            } else if (t.cls >= FWG_TGT_LINEAR) {
                p1 = -1.f;  // marks "forced constant" for a dynamic class
            }
            E.tprop[k][0] = p0; E.tprop[k][1] = p1; E.tprop[k][2] = p2; E.tprop[k][3] = p3;
        }
        E.tgt[k] = v;
    }
    E.flags = (E.flags & ((1u << FWG_FLAG_RESAMPLE_SHIFT) - 1u)) | ((resample + 1u) << FWG_FLAG_RESAMPLE_SHIFT);
}

// _get_next_target (fixed_wing.py:933-991)
__device__ __forceinline__ void next_targets(const DevCfg& c, Env& E) {
    float nt[3] = {E.tgt[0], E.tgt[1], E.tgt[2]};
#pragma unroll
    for (int k = 0; k < FWG_MAX_TARGETS; ++k) {
        if (k >= c.n_targets) continue;
        const DevTarget& t = c.target[k];
        if (t.cls == FWG_TGT_COMPENSATE) {
            // Va compensation for the pitch target (pitch target looked up by variable id, wave-uniform)
            float pitch_tgt = 0.f, pt = 0.f;
#pragma unroll
            for (int j = 0; j < FWG_MAX_TARGETS; ++j) {
                if (j < c.n_targets && c.target[j].var == FWG_V_PITCH) {
                    pitch_tgt = E.tgt[j];
                    pt = pitch_tgt;
                    if (c.any_dynamic_target && c.target[j].cls == FWG_TGT_SINUSOIDAL && E.tprop[j][1] >= 0.f) pt = E.tprop[j][3];
                }
            }
            const float va = E.tgt[k];
            if (pt <= -0.04363323129985824f) {  // radians(-2.5)
                const float va_end = 28.434f - 40.0841f * pt;
                float slope = 0.f;
                if (va <= va_end) slope = 7.f * fmaxf(0.f, (va < va_end * 0.95f) ? 1.f : 1.f - va / (va_end * 1.5f));
                nt[k] = va + (slope * (-pitch_tgt) - 0.25f) * c.dt;
            } else if (pt >= 0.08726646259971647f) {  // radians(5)
                const float va_end = 26.27f - 41.2529f * pt;
                if (va > va_end) nt[k] = (E.sft < 750u) ? va + (va_end - va) * (1.f / 150.f) : va_end;
            }
        } else if (c.any_dynamic_target && t.cls == FWG_TGT_LINEAR) {
            if (E.tprop[k][1] >= 0.f) nt[k] = E.tgt[k] + E.tprop[k][0] * c.dt;
        } else if (c.any_dynamic_target && t.cls == FWG_TGT_SINUSOIDAL) {
            if (E.tprop[k][1] >= 0.f)
                nt[k] = E.tprop[k][0] * sinf(FWG_TWO_PI / E.tprop[k][1] * ((float)E.steps + E.tprop[k][2])) + E.tprop[k][3];
        }
        if (t.wrap) nt[k] = wrap_target(nt[k]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) E.tgt[k] = nt[k];
}

// "action" observation entry (fixed_wing.py:813-828) for the newest row: sum of |diff| over the last `w` raw actions
// (or constrained commands) of actuator ai, or the back-scaled actuator value when no action has been taken yet.
// `ring` = this lane's column of the LDS action window [slot*3 + actuator][lane].
__device__ __forceinline__ float backscale_action(const DevCfg& c, int ai, float actuator) {
    if (c.scale_actions)
        return (c.scale_high - c.scale_low) * (actuator - c.act_to_low[ai]) * c.inv_act_span[ai] + c.scale_low;
    return actuator;
}
__device__ __forceinline__ float action_obs(const DevCfg& c, const float* ring, int ai, int w, unsigned n_act, int cur_slot,
                                            float actuator) {
    if (n_act < 1u) return backscale_action(c, ai, actuator);
    const int W = c.L.window;
    const int m = (int)min(n_act, (unsigned)w);
    float s = 0.f;
#pragma unroll
    for (int k = FWG_MAX_WINDOW - 2; k >= 0; --k) {
        if (k <= W - 2 && k <= m - 2) {
            int s_new = cur_slot - k; s_new += (s_new < 0) ? W : 0;
            int s_old = cur_slot - k - 1; s_old += (s_old < 0) ? W : 0;
            s += fabsf(ring[(s_new * 3 + ai) * FWG_WAVE] - ring[(s_old * 3 + ai) * FWG_WAVE]);
        }
    }
    return s;
}

// newest observation row (un-noised, normalised) into ob[0, n_obs) and, when `push`, into the lag ring (AoS record)
template <class TAB, class OB>
__device__ __forceinline__ void build_row0(const DevCfg& c, const KArgs& A, long e, const Env& E, const TAB& T, OB& ob,
                                           const float* ring, int ring_slot, bool push, int act_slot) {
#pragma unroll
    for (int j = 0; j < FWG_MAX_OBS; ++j) {
        if (j < c.n_obs) {
            const DevObs& o = c.obs[j];
            float v;
            if (o.type == FWG_OBS_STATE) v = T.get(o.src);
            else if (o.type == FWG_OBS_TARGET_RELATIVE) v = T.get(FWG_TAB_ERR + o.src);
            else if (o.type == FWG_OBS_TARGET_ABSOLUTE) v = T.get(FWG_TAB_TGT + o.src);
            else v = action_obs(c, ring, o.src, o.window, E.steps, act_slot, T.get(FWG_V_ELEVATOR + o.src));
            if (o.norm) v = (v - o.mean) * o.inv_var;
            ob.put(j, v);
            if (push && c.obs_length > 1) RING(A.S, A.N, c.L.lag_ring, ring_slot, c.n_obs, e, j) = v;
        }
    }
}

// lagged rows r >= 1 = the records pushed r*obs_step steps ago: streamed HBM -> LDS at kernel start
// (stream_lag_rows, global_load_lds: no VGPRs while the physics runs), collected into the record here
__device__ __forceinline__ void stream_lag_rows(const DevCfg& c, const KArgs& A, long e, float* lds_lag) {
    for (int r = 1; r < c.obs_length; ++r)
        for (int j = 0; j < c.n_obs; ++j)
            dma_row(&RING(A.S, A.N, c.L.lag_ring, A.lag_slots[r], c.n_obs, e, j), lds_lag + ((r - 1) * c.n_obs + j) * FWG_WAVE);
}
template <class OB>
__device__ __forceinline__ void load_lag_rows(const DevCfg& c, const float* lag_col, OB& ob) {
#pragma unroll
    for (int r = 1; r < FWG_MAX_ROWS; ++r) {
        if (r < c.obs_length) {
#pragma unroll
            for (int j = 0; j < FWG_MAX_OBS; ++j)
                if (j < c.n_obs) ob.put(r * c.n_obs + j, lag_col[((r - 1) * c.n_obs + j) * FWG_WAVE]);
        }
    }
}

// Fix-ups of the lagged rows r >= 1 that the plain ring read cannot provide (fixed_wing.py:790-832):
//  * rows reaching back to (or before) the start of the episode, i = 1 + r*step > steps_count: the INITIAL record plus
//    a fresh U(-1,1)*dt per row, with "action" entries replaced by the CURRENT actuator value;
//  * after a failed simulator step the state/target histories are one record shorter than the action history, so
//    the non-action entries come from one slot further back.
template <class TAB, class OB>
__device__ __forceinline__ void fix_lagged_rows(const DevCfg& c, const KArgs& A, long e, const Env& E, const TAB& T, OB& ob,
                                                bool ok) {
    const int depth = c.L.lag_depth;
    const int t = (int)E.steps;
    const unsigned env_id = (unsigned)(A.env_base + e);
#pragma unroll
    for (int r = 1; r < FWG_MAX_ROWS; ++r) {
        if (r >= c.obs_length) continue;
        const int lag = r * c.obs_step;
        if (lag >= t) {
            const u4 b = philox4x32(env_id, E.steps, E.episode, FWG_STREAM_INIT_NOISE + 256u * (r >> 2), A.seed_lo, A.seed_hi);
            const unsigned bits = (r & 3) == 0 ? b.x : ((r & 3) == 1 ? b.y : ((r & 3) == 2 ? b.z : b.w));
            const float noise = (2.f * u01(bits) - 1.f) * c.dt;
            int slot0 = A.slot_lag - t; slot0 += (slot0 < 0) ? depth : 0;  // ring slot of the episode's record 0
#pragma unroll
            for (int j = 0; j < FWG_MAX_OBS; ++j) {
                if (j >= c.n_obs) continue;
                const DevObs& o = c.obs[j];
                float v;
                if (o.type == FWG_OBS_ACTION) {
                    v = backscale_action(c, o.src, T.get(FWG_V_ELEVATOR + o.src)) + noise;
                    if (o.norm) v = (v - o.mean) * o.inv_var;
                } else {
                    v = RING(A.S, A.N, c.L.lag_ring, slot0, c.n_obs, e, j) + noise * (o.norm ? o.inv_var : 1.f);
                }
                ob.put(r * c.n_obs + j, v);
            }
        } else if (!ok) {
            int slot = A.slot_lag - 1 - lag; slot += (slot < 0) ? depth : 0;
#pragma unroll
            for (int j = 0; j < FWG_MAX_OBS; ++j)
                if (j < c.n_obs && c.obs[j].type != FWG_OBS_ACTION)
                    ob.put(r * c.n_obs + j, RING(A.S, A.N, c.L.lag_ring, slot, c.n_obs, e, j));
        }
    }
}

// optional Gaussian observation noise (fixed_wing.py:836-837), fresh for every entry of every row
template <class OB>
__device__ __forceinline__ void add_obs_noise(const DevCfg& c, const KArgs& A, long e, const Env& E, OB& ob) {
    const unsigned env_id = (unsigned)(A.env_base + e);
#pragma unroll
    for (int blk = 0; blk < (FWG_MAX_OBS * FWG_MAX_ROWS) / 4; ++blk) {
        if (blk * 4 >= c.obs_dim) continue;
        const u4 b = philox4x32(env_id, E.steps, E.episode, FWG_STREAM_OBS_NOISE + 256u * blk, A.seed_lo, A.seed_hi);
        float n[4];
        box_muller(b, n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = blk * 4 + i;
            if (k < c.obs_dim) ob.put(k, ob.get(k) + c.obs_noise_mean + c.obs_noise_std * n[i]);
        }
    }
}

// The 64 observation records of this wave -> out[env0 .. env0+63][obs_dim], which is one contiguous block of the
// [N][obs_dim] batch: every lane parks its record in the LDS staging area ([lane][obs_dim]) and the wave then writes
// the block in linear order, 1 KiB per store instruction (16 B per lane) when the record size allows.  `lanes` selects
// the records to write (all, or the finished episodes for the terminal observations).  Must be called by all lanes.
template <class OB>
__device__ __forceinline__ void write_obs(const DevCfg& c, float* __restrict__ out, long env0, long N, const OB& ob,
                                          float* stage, int lane, unsigned long long lanes) {
    const int D = c.obs_dim;
    __syncthreads();  // the staging area may still be read by a previous call
    if (obs_vec4(D)) {
        float4* mine = reinterpret_cast<float4*>(stage + lane * D);
#pragma unroll
        for (int q = 0; q < (FWG_MAX_OBS * FWG_MAX_ROWS) / 4; ++q)
            if (q * 4 < D) mine[q] = make_float4(ob.get(4 * q), ob.get(4 * q + 1), ob.get(4 * q + 2), ob.get(4 * q + 3));
        __syncthreads();
        const float4* all = reinterpret_cast<const float4*>(stage);
        float4* o4 = reinterpret_cast<float4*>(out + env0 * D);
        const int total4 = FWG_WAVE * D / 4;
#pragma unroll 4
        for (int i = lane; i < total4; i += FWG_WAVE) {
            const int l = (4 * i) / D;
            if (((lanes >> l) & 1ull) && env0 + l < N) o4[i] = all[i];
        }
    } else {
        const int Ds = obs_stage_stride(D);
#pragma unroll
        for (int k = 0; k < FWG_MAX_OBS * FWG_MAX_ROWS; ++k)
            if (k < D) stage[lane * Ds + k] = ob.get(k);
        __syncthreads();
        int l = lane / D, k = lane - l * D;
        const int l_inc = FWG_WAVE / D, k_inc = FWG_WAVE - l_inc * D;
#pragma unroll 4
        for (int idx = lane; idx < FWG_WAVE * D; idx += FWG_WAVE) {
            if (((lanes >> l) & 1ull) && env0 + l < N) out[env0 * D + idx] = stage[l * Ds + k];
            k += k_inc; l += l_inc;
            if (k >= D) { k -= D; ++l; }
        }
    }
}

// FixedWingAircraft.reset (fixed_wing.py:287-336) for one lane; `g_*` are the ring positions of the LAST completed
// global step.  Fills E, the observation record ob (all rows) and the ring slots that hold initial records.
template <bool TURB, class TAB, class OB>
__device__ __forceinline__ void reset_env(const DevCfg& c, const DynCfg& dc, const KArgs& A, long e, Env& E, TAB& T, OB& ob,
                                          const float* ring, int g_end, int g_lag, int g_bit) {
    const unsigned env_id = (unsigned)(A.env_base + e);
    E.episode += 1u;
    E.steps = 0u;
    E.flags &= FWG_FLAG_GOAL_ACHIEVED;  // prev_shaping := None, resample counter := 0; goal_achieved is sticky
    // ---- initial simulator state: given values or U(init_min, init_max)
    float v0[FWG_N_RESET_VARS];
#pragma unroll
    for (int blk = 0; blk < 6; ++blk) {
        const u4 b = philox4x32(env_id, E.episode, (unsigned)blk, FWG_STREAM_RESET_STATE, A.seed_lo, A.seed_hi);
        const unsigned bits[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = blk * 4 + i;
            if (v < FWG_N_RESET_VARS) {
                float x = dc.init_min[v] + (dc.init_max[v] - dc.init_min[v]) * u01(bits[i]);
                if (A.init_state != nullptr) {
                    const float gx = A.init_state[(long)v * A.N + e];
                    if (gx == gx) x = gx;
                }
                v0[v] = x;
            }
        }
    }
    {
        float sr, cr, sp, cp, sy, cy;
        sincosf(0.5f * v0[FWG_V_ROLL], &sr, &cr);
        sincosf(0.5f * v0[FWG_V_PITCH], &sp, &cp);
        sincosf(0.5f * v0[FWG_V_YAW], &sy, &cy);
        E.y[0] = cy * cp * cr + sy * sp * sr; E.y[1] = cy * cp * sr - sy * sp * cr;
        E.y[2] = cy * sp * cr + sy * cp * sr; E.y[3] = sy * cp * cr - cy * sp * sr;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) E.y[4 + i] = v0[FWG_V_OMEGA_P + i];
    {
        const float el = fclampf(v0[FWG_V_ELEVATOR], c.val_min[FWG_V_ELEVATOR], c.val_max[FWG_V_ELEVATOR]);
        const float ai = fclampf(v0[FWG_V_AILERON], c.val_min[FWG_V_AILERON], c.val_max[FWG_V_AILERON]);
        E.y[13] = el - ai; E.y[14] = el + ai;
        E.y[15] = fclampf(v0[FWG_V_THROTTLE], c.val_min[FWG_V_THROTTLE], c.val_max[FWG_V_THROTTLE]);
        E.y[16] = 0.f; E.y[17] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) E.wind[i] = v0[FWG_V_WIND_N + i];
#pragma unroll
    for (int i = 0; i < FWG_N_DRYDEN; ++i) E.dry[i] = 0.f;
    const float gust0[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    E.d = derive<false>(E.y, E.wind, gust0);
    fill_vars(E, T);
    // ---- targets
    sample_targets(c, dc, A, e, E, T, A.init_target);
    float err[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < FWG_MAX_TARGETS; ++k) {
        if (k < c.n_targets) {
            err[k] = target_error(c.target[k], E.tgt[k], T.get(c.target[k].var));
            T.put(FWG_TAB_TGT + k, E.tgt[k]);
            T.put(FWG_TAB_ERR + k, err[k]);
        }
    }
    // ---- accumulators of the episodic metrics (record 0 = the reset-time error / goal status, fixed_wing.py:318-325)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        E.e0[k] = err[k]; E.esum[k] = err[k]; E.eabs[k] = fabsf(err[k]); E.emin[k] = err[k]; E.emax[k] = err[k];
        E.rise[k] = 0xFFFFFFFFu; E.perr[k] = err[k];
        E.pcmd[k] = 0.f; E.psh[k] = 0.f;
    }
    E.settle[0] = 0xFFFFFFFFu; E.settle[1] = 0xFFFFFFFFu;
    E.sdcmd = 0.f;
    if (c.metrics) {
#pragma unroll
        for (int k = 0; k < FWG_MAX_TARGETS; ++k)
            if (k < c.n_targets) ROW(A.S, A.N, c.L.end_ring + g_end * 3 + k, e) = err[k];
    }
    E.wcnt = 0u; E.gcnt[0] = 0u; E.gcnt[1] = 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r) E.gword[r] = 0u;
    if (c.goal_enabled) {
        unsigned* U = reinterpret_cast<unsigned*>(A.S);
#pragma unroll
        for (int w = 0; w < 16; ++w) ROW(U, A.N, c.L.goal_ring + w, e) = 0u;  // the word holding g_bit is rewritten by store_gym
        goal_push(c, E, goal_flags(c, err), g_bit, 0u);
    }
    // ---- observation: every row is the initial record (+ per-row init noise when length > 1)
    build_row0(c, A, e, E, T, ob, ring, g_lag, true, 0);
    if (c.obs_length > 1) {
#pragma unroll
        for (int r = FWG_MAX_ROWS - 1; r >= 0; --r) {
            if (r >= c.obs_length) continue;
            const u4 b = philox4x32(env_id, 0u, E.episode, FWG_STREAM_INIT_NOISE + 256u * (r >> 2), A.seed_lo, A.seed_hi);
            const unsigned bits = (r & 3) == 0 ? b.x : ((r & 3) == 1 ? b.y : ((r & 3) == 2 ? b.z : b.w));
            const float noise = (2.f * u01(bits) - 1.f) * c.dt;
#pragma unroll
            for (int j = 0; j < FWG_MAX_OBS; ++j)
                if (j < c.n_obs) ob.put(r * c.n_obs + j, ob.get(j) + noise * (c.obs[j].norm ? c.obs[j].inv_var : 1.f));
        }
    }
    if (c.obs_noise) add_obs_noise(c, A, e, E, ob);
}
