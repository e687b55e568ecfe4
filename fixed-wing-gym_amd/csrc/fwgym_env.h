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
    unsigned long long* reduce;   // [FWG_N_REDUCE] fixed-point sums (2^-20; entries 0..4 are counts): integer atomics are
                              // native and order-free, float atomicAdd compiles to a compare-and-swap loop
    const uint8_t* mask;      // reset: nullable [N]
    const float* init_state;  // reset: nullable [FWG_N_RESET_VARS][N]
    const float* init_target; // reset: nullable [n_targets][N]
    unsigned seed_lo, seed_hi;
    unsigned* mq;             // nullable: queue of the envs reset in this launch (count | env indices), consumed by k_model_draw_q
    int slot_act, slot_end, slot_lag, bit_goal;  // ring positions of the CURRENT global step
    int lag_slots[FWG_MAX_ROWS];                 // ring slot holding the row pushed r*obs_step steps ago
    // graph mode (fwg_set_graph_mode): the ring positions live on the device (StepSlots below), double-buffered -- a step
    // launch reads *slots_in and its block 0 publishes the positions of the NEXT step to *slots_out, so that a captured
    // launch sequence can be replayed
    const struct StepSlots* slots_in;
    struct StepSlots* slots_out;
    int reset_launch;                            // k_reset: positions refer to the LAST completed step (counter - 1)
    long long gnow;                              // global index of the step this launch refers to (row-log positions)
    long long log_win;                           // row-log mode: first plane of this step's window, and whether the
    int log_wrap_now;                            // parity's wrap copy is due (host-computed; recomputed in graph mode)
    // attached rollout head (fwg_attach_observer): every wave also adds the moments of its 64 observation records and
    // discounted returns to the head's accumulators (acc_*), so the head needs no pass over the batch
    unsigned long long* acc;                     // nullable [3 sets][FWG_ACC_SHARDS][acc_cols] fixed-point sums (fwgym_actor.h)
    const unsigned* acc_ctr;                     // the head's act counter: this launch adds into set *acc_ctr % 3
    const float* acc_mean;                       // running observation mean [obs_dim] the deviations are taken from
    const float* acc_ret_mean;                   // ... and the running mean of the returns
    float* acc_ret;                              // [N] discounted returns (VecNormalize.ret)
    float acc_gamma;
    int acc_cols;
#ifdef FWG_TIMELINE   /* measurement builds only (tools/ablate.py): per-wave phase time stamps [block][16] */
    long long* trace;
#endif
};
#ifdef FWG_TIMELINE
#define FWG_TL_W 32   /* stamps per wave (tools/timeline.py TLW) */
#ifndef FWG_TL_WPB
#define FWG_TL_WPB 2  /* waves per workgroup of the traced kernel (k_step2; k_rollout: 8) -- blockDim.x is a memory load + wait per stamp */
#endif
#define FWG_TL(A, i) do { if ((A).trace != nullptr) { const long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); \
        if ((threadIdx.x & 63) == 0) (A).trace[(blockIdx.x * FWG_TL_WPB + (threadIdx.x >> 6)) * FWG_TL_W + (i)] = t_; } } while (0)
#else
#define FWG_TL(A, i) do { } while (0)
#endif

// ---- batch-moment accumulators shared by k_step, k_actor_stats and k_actor_act.  Columns: 0 / 1 sum and sum of
// squares of the return deviations (from the running mean), 2 / 3 number of observations / returns, 4 + 2k / 5 + 2k the
// same two sums for observation entry k.  Sums are added as 64-bit FIXED-POINT integers (2^-20 resolution): integer
// addition is associative, so the totals -- and with them the whole rollout -- do not depend on the order the waves
// arrive in.  FWG_ACC_SHARDS copies (wave index mod 16) keep the atomics off one cache line; the reader adds them.
#ifndef FWG_ACC_SHARDS
#define FWG_ACC_SHARDS 16
#endif
#define FWG_ACC_SCALE 1048576.f
#define FWG_ACC_SETS 3   /* accumulator sets, rotating with the head's act counter (fwgym_actor.h) */
__host__ __device__ inline int acc_cols_for(int D) { return (2 * D + 4 + 31) & ~31; }   // whole 32-column chunks

// value of lane (l ^ MASK)
#ifdef FWG_EMU
template <int MASK> __device__ __forceinline__ float lane_xor(float v) { return __shfl_xor(v, MASK, FWG_WAVE); }
#else
#define FWG_DPP_MOV(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
template <int MASK> __device__ __forceinline__ float lane_xor(float v) {
    if (MASK == 1) return FWG_DPP_MOV(v, 0xB1);                  // quad_perm:[1,0,3,2]
    if (MASK == 2) return FWG_DPP_MOV(v, 0x4E);                  // quad_perm:[2,3,0,1]
    if (MASK == 8) return FWG_DPP_MOV(v, 0x128);                 // row_ror:8
    if (MASK == 4 || MASK == 16)                                 // ds_swizzle, bit-mask mode: and 0x1F, or 0, xor MASK
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x1F | (MASK << 10)));
    return __shfl_xor(v, MASK, FWG_WAVE);
}
#endif
// 64-lane sum of one value
__device__ __forceinline__ float wave_sum64(float v) {
    v += lane_xor<1>(v); v += lane_xor<2>(v); v += lane_xor<4>(v); v += lane_xor<8>(v); v += lane_xor<16>(v);
    return v + lane_xor<32>(v);
}
// 64-lane sums of 32 values at once: each exchange step halves the number of values a lane carries (a lane keeps the
// half selected by one bit of its index and hands the other half to its partner), 31 exchanges instead of 32 x 6.
// Lane l returns the total of v[l & 31].
__device__ __forceinline__ float wave_totals32(float (&v)[32], int lane) {
#define FWG_TOT_STAGE(H, MASK)                                                              \
    _Pragma("unroll") for (int i = 0; i < (H); ++i) {                                        \
        const bool up = (lane & (MASK)) != 0;                                                \
        const float keep = up ? v[2 * i + 1] : v[2 * i], send = up ? v[2 * i] : v[2 * i + 1]; \
        v[i] = keep + lane_xor<MASK>(send);                                                  \
    }
    FWG_TOT_STAGE(16, 1) FWG_TOT_STAGE(8, 2) FWG_TOT_STAGE(4, 4) FWG_TOT_STAGE(2, 8) FWG_TOT_STAGE(1, 16)
#undef FWG_TOT_STAGE
    return v[0] + lane_xor<32>(v[0]);
}
// lanes 0..31 add the chunk's totals (columns 32 chunk + lane) to shard `shard`
__device__ __forceinline__ void acc_flush(unsigned long long* acc, int cols, int shard, int chunk, int lane, float total) {
    const int col = 32 * chunk + lane;
    if (lane < 32 && col < cols) {
        const long long q = (col == 2 || col == 3) ? (long long)rintf(total) : (long long)rintf(total * FWG_ACC_SCALE);
#ifndef FWG_ABL_NO_FLUSH
        atomicAdd(acc + (long)shard * cols + col, (unsigned long long)q);
#else
        if (q == 0x7fffffffffffffffll) acc[0] = 1;
#endif
    }
}
// value of accumulator column `col` for one lane: x(k) = observation entry k (called only for k < D), dr = return deviation
#define FWG_ACC_COLUMN(col, D, valid, X, MEAN, dr, has_obs, has_ret)                                         \
    ((col) == 0 ? (dr) : (col) == 1 ? (dr) * (dr) : (col) == 2 ? (((valid) && (has_obs)) ? 1.f : 0.f)            \
     : (col) == 3 ? (((valid) && (has_ret)) ? 1.f : 0.f)                                                         \
     : ((((col) - 4) >> 1) < (D) && (valid) && (has_obs))                                                        \
           ? ((((col) & 1) == 0) ? ((X((((col) - 4) >> 1))) - (MEAN)[(((col) - 4) >> 1)])                          \
                                 : ((X((((col) - 4) >> 1))) - (MEAN)[(((col) - 4) >> 1)]) * ((X((((col) - 4) >> 1))) - (MEAN)[(((col) - 4) >> 1)])) \
           : 0.f)

// ---- observation row log (DevCfg::obs_log = L > 0, include/fwgym.h "Row-log observations"): float [obs_step][L][N][n_obs].
// A parity (global step mod obs_step) appends its records in DESCENDING row order from row P - 1 down to 0,
// P = L - (length - 1); the step that would run off the top first copies the length - 1 newest rows (0 .. length-2) to
// rows P .. L-1, so the window [newest, newest + length) is always contiguous.
__host__ __device__ inline long long log_fdiv(long long a, int b) { const long long q = a / b; return (a % b != 0 && a < 0) ? q - 1 : q; }
__host__ __device__ inline int log_pmod(long long a, int m) { return (int)(((a % m) + m) % m); }
// plane (row index over all parities) holding the record of global step gp, as seen at global step g >= gp
__host__ __device__ inline long long log_plane(int S, int L, int len, long long g, long long gp) {
    const int P = L - (len - 1);
    const int p = log_pmod(gp, S);
    const long long qp = log_fdiv(gp, S);
    const long long qc = log_fdiv(g - log_pmod(g - p, S), S);   // newest step of that parity not after g
    // A record is written at its HOME plane P - 1 - (qp mod P); the parity's wrap step carries the len - 1 newest records (home
    // planes 0 .. len - 2) to home + P, where the windows after the wrap find them.  A record one further back -- the oldest row
    // of a FAILED step's terminal observation at obs_step 1, which shows the window of the step before -- was not carried: it
    // is still at its home plane (the window reaches that plane again only P - len steps later).  Rounds 1-5 computed home + P
    // for it as well: plane L, one past the parity's planes -- 1 failure end in P on the shipped cnn configuration read its
    // oldest terminal row from whatever follows the log (found by tests/test_emu_fuzz.py, seeds 7 and 8, round 6).
    const int home = P - 1 - log_pmod(qp, P);
    const long long wraps = log_fdiv(qc, P) - log_fdiv(qp, P);
    return (long long)p * L + home + ((wraps > 0 && home <= len - 2) ? (long long)P * wraps : 0);
}
// Ring positions of one global step g (graph mode keeps them on the device): everything here is g modulo something,
// so the next step's positions follow from this step's by increments with wrap-around -- no division on the device.
struct StepSlots {
    long long gnow;                              // the global step index g
    long long log_win;                           // row log: first plane of step g's window
    int log_wrap_now;                            // row log: the parity's wrap copy is due at step g
    int slot_act, slot_end, slot_lag, bit_goal;
    int gmod_s, qmod_p;                          // row log: g mod obs_step, (g div obs_step) mod P
    int lag_slots[FWG_MAX_ROWS];
    int pad_;
};
__host__ __device__ inline int slots_pmod(long long a, int m) { return m > 0 ? (int)(((a % m) + m) % m) : 0; }
// positions of step g from scratch (host; k_reset)
__host__ __device__ inline StepSlots make_slots(int obs_step, int obs_log, int obs_length, int window, int lag_depth, int streak_req,
                                                long long g) {
    StepSlots s;
    s.gnow = g;
    s.log_win = 0; s.log_wrap_now = 0; s.gmod_s = 0; s.qmod_p = 0; s.pad_ = 0;
    if (obs_log > 0) {
        const int P = obs_log - (obs_length - 1);
        s.gmod_s = log_pmod(g, obs_step);
        s.qmod_p = log_pmod(log_fdiv(g, obs_step), P);
        s.log_win = (long long)s.gmod_s * obs_log + (P - 1 - s.qmod_p);   // = log_plane(obs_step, obs_log, obs_length, g, g)
        s.log_wrap_now = s.qmod_p == 0;
    }
    s.slot_act = slots_pmod(g, window);
    s.slot_end = slots_pmod(g, FWG_END_RING);
    s.slot_lag = slots_pmod(g, lag_depth);
    s.bit_goal = slots_pmod(g, streak_req);
    for (int r = 0; r < FWG_MAX_ROWS; ++r) s.lag_slots[r] = slots_pmod(g - (long long)r * obs_step, lag_depth);
    return s;
}
__host__ __device__ inline int slots_inc(int v, int m) { return (m > 0 && v + 1 < m) ? v + 1 : 0; }
// positions of step g + 1 from those of step g
__host__ __device__ inline StepSlots next_slots(int obs_step, int obs_log, int obs_length, int window, int lag_depth, int streak_req,
                                                const StepSlots& c) {
    StepSlots s = c;
    s.gnow = c.gnow + 1;
    if (obs_log > 0) {
        const int P = obs_log - (obs_length - 1);
        s.gmod_s = c.gmod_s + 1;
        if (s.gmod_s >= obs_step) { s.gmod_s = 0; s.qmod_p = slots_inc(c.qmod_p, P); }
        s.log_win = (long long)s.gmod_s * obs_log + (P - 1 - s.qmod_p);
        s.log_wrap_now = s.qmod_p == 0;
    }
    s.slot_act = slots_inc(c.slot_act, window);
    s.slot_end = slots_inc(c.slot_end, FWG_END_RING);
    s.slot_lag = slots_inc(c.slot_lag, lag_depth);
    s.bit_goal = slots_inc(c.bit_goal, streak_req);
    for (int r = 0; r < FWG_MAX_ROWS; ++r) s.lag_slots[r] = slots_inc(c.lag_slots[r], lag_depth);
    return s;
}
// ring positions of this launch: host-computed kernel arguments, or read from the device-resident StepSlots
__device__ __forceinline__ void apply_slots(KArgs& A, const StepSlots& s) {
    A.gnow = s.gnow; A.log_win = s.log_win; A.log_wrap_now = s.log_wrap_now;
    A.slot_act = s.slot_act; A.slot_end = s.slot_end; A.slot_lag = s.slot_lag; A.bit_goal = s.bit_goal;
#pragma unroll
    for (int r = 0; r < FWG_MAX_ROWS; ++r) A.lag_slots[r] = s.lag_slots[r];
}
// (read through the CONSTANT address space: the record of THIS launch is not written during it -- block 0 publishes the next
// one into the other buffer -- and as scalar loads the positions cost no vector-memory round trip: through a plain pointer the
// physics wave, which also stores the next record, read them with global_load + s_waitcnt BEFORE requesting its first rows)
__device__ __forceinline__ StepSlots load_slots(const StepSlots* p) {
    FWG_KCONST(StepSlots)* q = (FWG_KCONST(StepSlots)*)p;
    StepSlots s;
    s.gnow = q->gnow; s.log_win = q->log_win; s.log_wrap_now = q->log_wrap_now;
    s.slot_act = q->slot_act; s.slot_end = q->slot_end; s.slot_lag = q->slot_lag; s.bit_goal = q->bit_goal;
    s.gmod_s = q->gmod_s; s.qmod_p = q->qmod_p; s.pad_ = 0;
#pragma unroll
    for (int r = 0; r < FWG_MAX_ROWS; ++r) s.lag_slots[r] = q->lag_slots[r];
    return s;
}
__device__ __forceinline__ KArgs resolve_slots(const DevCfg& c, const KArgs& A0) {
    KArgs A = A0;
    if (A0.slots_in != nullptr) {
        // Freshness under graph replay: the record was written by the launch before this one (a vector store) and the same
        // address was scalar-loaded two launches ago, so a scalar data cache that survived two dispatches would hand the kernel
        // the positions of step g - 2.  Every AQL kernel-dispatch packet -- the kernel nodes of a replayed graph included --
        // carries an acquire fence that invalidates that cache; what graph nodes skip is the completion-signal handshake.  Not
        // relying on it was measured (round 6, same-box A/B, profiles/r06_slots_ab.txt): s_dcache_inv once per wave
        // (FWG_SLOTS_DCACHE_INV) costs +0.3 us per C3 step -- every wave re-fetches its kernel arguments and configuration
        // words --, scalar loads with GLC (miss the scalar cache) 91 us per step instead of 10.5.  The default trusts the
        // dispatch; tests/test_obs_log.py replays 64-step graphs through nine time limits against eager launches, where a stale
        // record shows as a wrong observation window.
#ifdef FWG_SLOTS_DCACHE_INV
        const StepSlots* in = fwg_fresh_scalar_view(A0.slots_in);
#else
        const StepSlots* in = A0.slots_in;
#endif
        if (A0.reset_launch)   // k_reset: positions of the LAST completed step (rare launch: computed from scratch)
            apply_slots(A, make_slots(c.obs_step, c.obs_log, c.obs_length, c.L.window, c.L.lag_depth, c.streak_req, load_slots(in).gnow - 1));
        else
            apply_slots(A, load_slots(in));
    }
    return A;
}

__device__ __forceinline__ float* log_row(const DevCfg& c, float* log, long N, long e, long long plane) {
    return log + ((plane * N + e) * c.n_obs);
}
// record r (0 = newest) of the observation record `ob` -> row r of the window starting at plane `win`
template <class OB>
__device__ __forceinline__ void log_store_row(const DevCfg& c, float* log, long N, long e, long long win, int r, const OB& ob) {
    float* row = log_row(c, log, N, e, win + r);
    if ((c.n_obs & 3) == 0) {
#pragma unroll
        for (int q = 0; q < FWG_MAX_OBS / 4; ++q)
            if (4 * q < c.n_obs)
                reinterpret_cast<float4*>(row)[q] = make_float4(ob.get(r * c.n_obs + 4 * q), ob.get(r * c.n_obs + 4 * q + 1),
                                                                ob.get(r * c.n_obs + 4 * q + 2), ob.get(r * c.n_obs + 4 * q + 3));
    } else {
#pragma unroll
        for (int j = 0; j < FWG_MAX_OBS; ++j)
            if (j < c.n_obs) row[j] = ob.get(r * c.n_obs + j);
    }
}
// rows r >= 1 of the window starting at plane `win` (the records of steps g - r obs_step) -> ob (rare per-lane reads)
template <class OB>
__device__ __forceinline__ void log_load_rows(const DevCfg& c, const float* log, long N, long e, long long win, OB& ob) {
#pragma unroll
    for (int r = 1; r < FWG_MAX_ROWS; ++r) {
        if (r < c.obs_length) {
            const float* row = log + (((win + r) * N + e) * c.n_obs);
            if ((c.n_obs & 3) == 0) {   // 16 bytes per load (the rows are 16-byte aligned then)
#pragma unroll
                for (int q = 0; q < FWG_MAX_OBS / 4; ++q) {
                    if (4 * q < c.n_obs) {
                        const float4 v = reinterpret_cast<const float4*>(row)[q];
                        ob.put(r * c.n_obs + 4 * q, v.x); ob.put(r * c.n_obs + 4 * q + 1, v.y);
                        ob.put(r * c.n_obs + 4 * q + 2, v.z); ob.put(r * c.n_obs + 4 * q + 3, v.w);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < FWG_MAX_OBS; ++j)
                    if (j < c.n_obs) ob.put(r * c.n_obs + j, row[j]);
            }
        }
    }
}
// the parity's wrap step: carry the length - 1 newest rows to the top of the log (all lanes, coalesced per row)
__device__ __forceinline__ void log_wrap(const DevCfg& c, float* log, long N, long e, long long g, bool valid, int due) {
    const int S = c.obs_step, L = c.obs_log, P = L - (c.obs_length - 1);
    if (!due) return;   // wave-uniform
    const long long base = (long long)log_pmod(g, S) * L;
    if (valid) {
        for (int r = 0; r < c.obs_length - 1; ++r) {
            const float* src = log + (((base + r) * N + e) * c.n_obs);
            float* dst = log + (((base + P + r) * N + e) * c.n_obs);
            for (int j = 0; j < c.n_obs; ++j) dst[j] = src[j];
        }
    }
}

// LDS carve (in floats) for one 64-lane block: the action windows (streamed in by global_load_lds) and, for the
// generic (non-specialised) kernel only, the per-lane scratch tables that config-driven indices address
struct LdsMap { int aring, cring, lag, stage, flag, tab, obs, total; };
#define FWG_TAB_TGT FWG_N_VARS               // table rows: simulator variables | targets | target errors
#define FWG_TAB_ERR (FWG_N_VARS + FWG_MAX_TARGETS)
#define FWG_TAB_INT (FWG_N_VARS + 2 * FWG_MAX_TARGETS)   // ... | windowed error sums (integrator observations)
#define FWG_TAB_ROWS (FWG_N_VARS + 3 * FWG_MAX_TARGETS)
// row stride (words) of the output staging: records of 4q words with q odd are written/read with conflict-free 16-byte
// LDS accesses; any other size falls back to an odd stride and 4-byte accesses
__host__ __device__ inline bool obs_vec4(int obs_dim) { return (obs_dim % 4 == 0) && ((obs_dim / 4) % 2 == 1); }
__host__ __device__ inline int obs_stage_stride(int obs_dim) { return obs_vec4(obs_dim) ? obs_dim : (obs_dim | 1); }
// k_step2 hand-off areas (per lane): new state, noise, actuator states.  They ALIAS the output staging area: every use of
// the staging area comes after the physics wave's first hand-shake mark (FWG_FLAG_WAIT level 1: its hand-off areas are read),
// every hand-off access before it.  The mark itself is a word of its OWN (LdsMap::flag, outside the staging area: the physics
// wave raises it to 2 long after its partner may have begun staging output records, so it must not alias a staged value).
// Workgroup residency is bounded by LDS (4 workgroups per CU need <= 40 KiB each; a fifth area would cost a fourth of the chip).
#define FWG_COOP_ENDS 4   // ending lanes per wave whose terminal rows the wave copies cooperatively (k_step2, partner_rows)
#define FWG_HAND_WORDS 28   /* physics -> gym, seven 16-byte groups per lane (odd: conflict-free): y[4..15] | the five Euler-angle arguments, tag |
                               Va alpha beta, tag + failure code | (failed step) alpha beta of the last valid state, tag */
#define FWG_TAIL_WORDS 4    /* gym -> physics: step index, padding-row index, install flag, tag */
#define FWG_ACT_WORDS 12    /* gym -> physics: actuator states at t + h/2 and t + h (2 x 5), tag: three groups per lane */
#define FWG_SPLIT_WORDS (FWG_HAND_WORDS + FWG_TAIL_WORDS + FWG_ACT_WORDS)
__host__ __device__ inline LdsMap lds_map(int obs_dim, int n_obs, int window, int use_cmd_ring, bool generic, int obs_log = 0,
                                          bool split = false) {
    LdsMap m;
    int o = 0;
    const int rows = obs_dim / n_obs, ng = (n_obs + 3) / 4;
    m.aring = o; o += window * 4 * FWG_WAVE;               // action window [slot][lane][4]
    m.cring = o; o += (use_cmd_ring ? window * 4 * FWG_WAVE : 0);
    m.lag = o; o += obs_log > 0 ? 0 : (rows - 1) * ng * 4 * FWG_WAVE;   // lagged records [row-1][group][lane][4] (dense batch only)
    const int stage = FWG_WAVE * obs_stage_stride(obs_dim), hand = split ? FWG_WAVE * FWG_SPLIT_WORDS : 0;
    m.stage = o; o += stage > hand ? stage : hand;         // [lane][obs_dim] staging of the output records
    m.flag = o; o += split ? 4 : 0;                        // k_step2: the one-way hand-shake mark (one word used)
    m.tab = o; o += generic ? FWG_TAB_ROWS * FWG_WAVE : 0;
    m.obs = o; o += generic ? obs_dim * FWG_WAVE : 0;
    m.total = (o + 3) & ~3;
    return m;
}

// Per-lane tables behind one interface.  In a specialised kernel every index is a compile-time constant after
// unrolling, so the register-array flavour costs nothing; the generic kernel indexes lane-private LDS columns
// ([entry][lane], conflict-free) with wave-uniform run-time indices.
template <int ROWS> struct RegTable {
    static constexpr bool in_regs = true;
    float v[ROWS];
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void put(int i, float x) { v[i] = x; }
};
struct LdsTable {
    static constexpr bool in_regs = false;
    float* p;
    __device__ __forceinline__ float get(int i) const { return p[i * FWG_WAVE]; }
    __device__ __forceinline__ void put(int i, float x) { p[i * FWG_WAVE] = x; }
};

// ---- The end-error ring holds the episode's CUMULATIVE error sums (one 16-byte slot per record, 51 slots): the sum over the
// last 50 records -- end_error, fixed_wing.py:1106-1107 -- is the difference of two slots, one load at the episode end instead of
// the whole window; integration_window sums likewise.  The sums are 42-bit FIXED POINT (2^-22 resolution, three to a slot): a
// float32 running sum reaches 10^3 - 10^4 late in a 2 000-step episode and the difference of two of them then carries 1e-4 - 1e-3
// of absolute error (rounds 1-3); integer sums are exact, so the window sum carries only the 50 quantisation errors of its own
// terms (< 6e-6, 1e-7 on the mean) whatever the episode length.  |error| < 512 by the quantiser; the sums wrap modulo 2^42 and
// every use is a difference reduced back to 42 bits (fix_wrap).
#define FWG_ESUM_FRAC 22
#define FWG_ESUM_SCALE 4194304.f
struct Fix3 { long long s[3]; };
__device__ __forceinline__ long long fix_quant(float err) {
#ifdef FWG_EMU   /* the device's v_cvt_i32_f32: round to nearest even, saturating, NaN -> 0 */
    const float x = rintf(err * FWG_ESUM_SCALE);
    if (!(x == x)) return 0ll;
    return x >= 2147483648.f ? 2147483647ll : (x <= -2147483648.f ? -2147483648ll : (long long)x);
#else
    return (long long)__float2int_rn(err * FWG_ESUM_SCALE);   // (v_cvt_i32_f32 saturates)
#endif
}
// A DIFFERENCE of ring values, reduced to the ring's 42 bits: the packed sums wrap modulo 2^42 (|sum| < 2^41 holds for ~1k steps
// of a saturated error, not for an episode without a step limit), and every window sum is a difference of two of them -- exact
// in modular arithmetic for any episode length as long as the window's own sum fits (51 terms of < 2^31: it does)
__device__ __forceinline__ long long fix_wrap(long long d) { return (long long)((unsigned long long)d << 22) >> 22; }
__device__ __forceinline__ float fix_to_float(long long d) {   // d 2^-22 for |d| < 2^47: two exact conversions and one fma
    const int hi = (int)(d >> 16), lo = (int)(d & 0xFFFFll);
    return ((float)hi * 65536.f + (float)lo) * (1.f / FWG_ESUM_SCALE);
}
__device__ __forceinline__ float4 fix3_pack(const Fix3& f) {
    const unsigned long long M = 0x3FFFFFFFFFFull;
    const unsigned long long a = (unsigned long long)f.s[0] & M, b = (unsigned long long)f.s[1] & M, c = (unsigned long long)f.s[2] & M;
    const unsigned long long lo = a | (b << 42), hi = (b >> 22) | (c << 20);
    return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
                       __uint_as_float((unsigned)(hi >> 32)));
}
__device__ __forceinline__ Fix3 fix3_unpack(const float4& q) {
    const unsigned long long M = 0x3FFFFFFFFFFull;
    const unsigned long long lo = (unsigned long long)__float_as_uint(q.x) | ((unsigned long long)__float_as_uint(q.y) << 32);
    const unsigned long long hi = (unsigned long long)__float_as_uint(q.z) | ((unsigned long long)__float_as_uint(q.w) << 32);
    Fix3 f;
    f.s[0] = (long long)((lo & M) << 22) >> 22;
    f.s[1] = (long long)((((lo >> 42) | (hi << 22)) & M) << 22) >> 22;
    f.s[2] = (long long)(((hi >> 20) & M) << 22) >> 22;
    return f;
}

struct Env {
    float y[NY];
    float wind[3];
    float dry[FWG_N_DRYDEN];
    float gust[6];       // increment turbulence: this step's gust sample (computed when the filter advanced one step ago)
    float gust_gain;     // simulator.turbulence / turbulence_intensity sampled per env and episode: gain on the gust (cold row)
    float int_reset[3];  // integration_window: what the integrator entries of the NEXT reset observation show (fixed_wing.py:317-321)
    Derived d;
    float tgt[FWG_MAX_TARGETS];
    float tprop[FWG_MAX_TARGETS][4];  // slope|amplitude, period, phase, bias
    unsigned steps, sft, flags, episode;
    float psh[3];
    unsigned gw;         // the word of the goal-window ring that holds this step's position (8 positions x 4 flags)
    unsigned wcnt;       // ones inside each of the 4 windows (target0..2, all), 4 x 8 bit
    unsigned gcnt[2];    // cumulative ones per window since reset, 4 x 16 bit
    float e0[3], esum[3], eabs[3], emin[3], emax[3];
    unsigned rise[3];
    unsigned settle[2];
    float perr[3];
    float sdcmd;
    float fscale[FWG_MAX_FACTORS];    // reward.randomize_scaling: this env's 1 / scaling per reward factor (read-only in the step)
    float was_emin[3], was_emax[3];   // values as loaded: the rarely-changing groups are written back only when they changed
    unsigned was_rise[3];
};

// Arena addressing.  The arena is an array of 16-byte GROUPS [group][env]: word w of env e lives in group w>>2,
// component w&3.  A wave touching one group of its 64 envs moves 1 KiB with ONE vector-memory instruction -- the per-CU
// address path costs about the same per instruction whether a lane moves 4 or 16 bytes, so the number of instructions,
// not the bytes, is what the layout minimises.  32-bit indices (fwg_create guarantees groups*N < 2^28).
// (address = a wave-uniform base -- arena + group * N * 16, scalar arithmetic -- plus a 32-bit per-lane byte offset e * 16: the
// form global_load / global_store take as `saddr + voffset`, so that ONE vector register addresses every group of the lane
// instead of a 64-bit address pair per group held across the kernel)
#define GROUP(S, N, g, e) (*reinterpret_cast<float4*>(reinterpret_cast<char*>(S) + (size_t)(unsigned)(g) * (size_t)(unsigned)(N) * 16u + (size_t)((unsigned)(e) << 4)))
#define CGROUP(S, N, g, e) (*reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S) + (size_t)(unsigned)(g) * (size_t)(unsigned)(N) * 16u + (size_t)((unsigned)(e) << 4)))
// rows written now and read back only some steps later (lag ring, end-error ring): streaming stores, measured -0.3 us
// per C3 step; FWG_NO_NT_RING_STORES restores plain stores
__device__ __forceinline__ void store_group_once(float* S, long N, int g, long e, float4 v) {
#if !defined(FWG_NO_NT_RING_STORES) && !defined(FWG_EMU)
    const fwg_v4f q = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(q, reinterpret_cast<fwg_v4f*>(S) + ((unsigned)g * (unsigned)N + (unsigned)e));
#else
    GROUP(S, N, g, e) = v;
#endif
}
// state rows loaded with a streaming hint (experiment knob FWG_NT_STATE_LOADS) and the plain form
__device__ __forceinline__ float4 load_group(const float* S, long N, int g, long e) {
#if defined(FWG_NT_STATE_LOADS) && !defined(FWG_EMU)
    const fwg_v4f q = __builtin_nontemporal_load(reinterpret_cast<const fwg_v4f*>(S) + ((unsigned)g * (unsigned)N + (unsigned)e));
    return make_float4(q.x, q.y, q.z, q.w);
#else
    return CGROUP(S, N, g, e);
#endif
}
__device__ __forceinline__ float u2f(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ unsigned f2u(float f) { return __float_as_uint(f); }

// simulator block: y[18] | dryden[8] | gust[6] = 8 groups with increment turbulence, 7 with the filter outputs (the last two
// words unused), 5 when turbulence is off
template <bool TURB>
__device__ __forceinline__ void load_sim(const DevCfg& c, const float* __restrict__ S, long N, long e, Env& E) {
    const int g0 = c.L.sim >> 2;
    float f[32];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (g < (TURB ? (c.turb_increment ? 8 : 7) : 5)) {
            const float4 q = load_group(S, N, g0 + g, e);
            f[4 * g] = q.x; f[4 * g + 1] = q.y; f[4 * g + 2] = q.z; f[4 * g + 3] = q.w;
        } else {
            f[4 * g] = 0.f; f[4 * g + 1] = 0.f; f[4 * g + 2] = 0.f; f[4 * g + 3] = 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < NY; ++i) E.y[i] = f[i];
    if (TURB) {
#pragma unroll
        for (int i = 0; i < FWG_N_DRYDEN; ++i) E.dry[i] = f[NY + i];
#pragma unroll
        for (int i = 0; i < 6; ++i) E.gust[i] = f[NY + FWG_N_DRYDEN + i];
    }
}
// per-episode constants (written by reset only): steady wind + episode counter
__device__ __forceinline__ void load_cold(const DevCfg& c, const float* __restrict__ S, long N, long e, Env& E) {
    const float4 w = CGROUP(S, N, (c.L.cold >> 2), e);
    E.wind[0] = w.x; E.wind[1] = w.y; E.wind[2] = w.z; E.episode = f2u(w.w);
    E.gust_gain = 1.f;
    if (c.sim_keys) E.gust_gain = CGROUP(S, N, (c.L.cold >> 2) + 1, e).x;
}

// simulator.model: this lane's force / moment constants (arena section L.aero) / the shared set of the configuration
__device__ __forceinline__ void load_aero(const DevCfg& c, const float* __restrict__ S, long N, long e, Aero& a) {
    float v[4 * FWG_AERO_GROUPS];
#pragma unroll
    for (int g = 0; g < FWG_AERO_GROUPS; ++g) {
        const float4 q = CGROUP(S, N, (c.L.aero >> 2) + g, e);
        v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
    }
    int i = 0;
#define FWG_AERO_TAKE(n) a.n = v[i++];
    FWG_AERO_LIST(FWG_AERO_TAKE)
#undef FWG_AERO_TAKE
}
__device__ __forceinline__ void aero_from_cfg(const DevCfg& c, Aero& a) {
#define FWG_AERO_TAKE(n) a.n = c.n;
    FWG_AERO_LIST(FWG_AERO_TAKE)
#undef FWG_AERO_TAKE
}

// The write-back is split so that each part is issued as soon as its values are final: the simulator block right after
// the integration, the bookkeeping after the gym logic -- the store traffic then overlaps the remaining computation
// instead of forming one burst at the end of the kernel.
template <bool TURB>
__device__ __forceinline__ void store_sim(const DevCfg& c, float* __restrict__ S, long N, long e, const Env& E) {
    const int g0 = c.L.sim >> 2;
    float f[32];
#pragma unroll
    for (int i = 0; i < NY; ++i) f[i] = E.y[i];
#pragma unroll
    for (int i = 0; i < FWG_N_DRYDEN; ++i) f[NY + i] = TURB ? E.dry[i] : 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) f[NY + FWG_N_DRYDEN + i] = (TURB && c.turb_increment) ? E.gust[i] : 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g)
        if (g < (TURB ? (c.turb_increment ? 8 : 7) : 5)) GROUP(S, N, g0 + g, e) = make_float4(f[4 * g], f[4 * g + 1], f[4 * g + 2], f[4 * g + 3]);
    // derived values of the committed state: for host views (controllers, rendering) -- and, with turbulence, the air data for
    // the kernel itself: after a FAILED step the simulator's Va / alpha / beta stay what the last committed step left (PyFly does
    // not touch its state objects then), i.e. values derived with THAT step's gust, which is gone by the time the failure is known
    if (c.store_derived) GROUP(S, N, (c.L.derived >> 2), e) = make_float4(E.d.roll, E.d.pitch, E.d.yaw, E.d.Va);
    if (c.store_derived || TURB) GROUP(S, N, (c.L.derived >> 2) + 1, e) = make_float4(E.d.alpha, E.d.beta, E.d.Va, 0.f);
}
__device__ __forceinline__ void store_cold(const DevCfg& c, float* __restrict__ S, long N, long e, const Env& E) {
    GROUP(S, N, (c.L.cold >> 2), e) = make_float4(E.wind[0], E.wind[1], E.wind[2], u2f(E.episode));
    if (c.sim_keys) GROUP(S, N, (c.L.cold >> 2) + 1, e) = make_float4(E.gust_gain, 0.f, 0.f, 0.f);
}

// bookkeeping block, 9 groups.  Written every step: 0: tgt0 tgt1 tgt2 steps|sft<<16 | 1: flags wcnt gcnt0 gcnt1 and, with
// metrics, 2: esum[0..2] perr0 | 3: eabs[0..2] perr1 | 4: sdcmd perr2 settle0 settle1.  Read every step but written
// only when a lane of the wave changed them (running extremes, rise-time latches, the episode's initial errors):
// 5: emin[0..2] rise0 | 6: emax[0..2] rise1 | 7: rise2 e0[0..2].  8: psh[0..2] (potential rewards only).
// Then 3 groups of target properties (linear/sinusoidal targets only) and the goal-window ring as 16 plain word rows.
// `part`: 3 = everything; 1 = what the work BEFORE the hand-over reads (targets, counters, flags, goal word, target properties,
// reward scalings); 2 = the metric accumulators and the previous shaping terms, read after it (k_step2 requests them when its
// first work is under way: the rows of a launch are all wanted in its first microsecond, and what is asked for then queues behind
// everything else that is)
__device__ __forceinline__ void load_gym(const DevCfg& c, const float* __restrict__ S, long N, long e, Env& E, int goal_bit, int part = 3) {
    const int g0 = c.L.gym >> 2;
    float4 q[9];
#pragma unroll
    for (int g = 0; g < 9; ++g)
        if (((g < 2) && (part & 1)) || ((part & 2) && ((g >= 2 && g < 8 && c.metrics) || (g == 8 && c.reward_potential)))) q[g] = load_group(S, N, g0 + g, e);
    if (part & 2) {
        if (c.metrics) {
            E.esum[0] = q[2].x; E.esum[1] = q[2].y; E.esum[2] = q[2].z; E.perr[0] = q[2].w;
            E.eabs[0] = q[3].x; E.eabs[1] = q[3].y; E.eabs[2] = q[3].z; E.perr[1] = q[3].w;
            E.sdcmd = q[4].x; E.perr[2] = q[4].y; E.settle[0] = f2u(q[4].z); E.settle[1] = f2u(q[4].w);
            E.emin[0] = q[5].x; E.emin[1] = q[5].y; E.emin[2] = q[5].z; E.rise[0] = f2u(q[5].w);
            E.emax[0] = q[6].x; E.emax[1] = q[6].y; E.emax[2] = q[6].z; E.rise[1] = f2u(q[6].w);
            E.rise[2] = f2u(q[7].x); E.e0[0] = q[7].y; E.e0[1] = q[7].z; E.e0[2] = q[7].w;
#pragma unroll
            for (int k = 0; k < 3; ++k) { E.was_emin[k] = E.emin[k]; E.was_emax[k] = E.emax[k]; E.was_rise[k] = E.rise[k]; }
        }
        if (c.reward_potential) { E.psh[0] = q[8].x; E.psh[1] = q[8].y; E.psh[2] = q[8].z; }
    }
    if (!(part & 1)) return;
    if (c.randomize_scaling) {
#pragma unroll
        for (int g = 0; g < FWG_MAX_FACTORS / 4; ++g) {
            if (4 * g < c.n_factors) {
                const float4 v = CGROUP(S, N, (c.L.fscale >> 2) + g, e);
                E.fscale[4 * g] = v.x; E.fscale[4 * g + 1] = v.y; E.fscale[4 * g + 2] = v.z; E.fscale[4 * g + 3] = v.w;
            }
        }
    }
    E.tgt[0] = q[0].x; E.tgt[1] = q[0].y; E.tgt[2] = q[0].z;
    E.steps = f2u(q[0].w) & 0xFFFFu; E.sft = f2u(q[0].w) >> 16;
    E.flags = f2u(q[1].x); E.wcnt = f2u(q[1].y); E.gcnt[0] = f2u(q[1].z); E.gcnt[1] = f2u(q[1].w);
    if (c.any_dynamic_target) {
#pragma unroll
        for (int k = 0; k < FWG_MAX_TARGETS; ++k) {
            const float4 t = CGROUP(S, N, (c.L.tprop >> 2) + k, e);
            E.tprop[k][0] = t.x; E.tprop[k][1] = t.y; E.tprop[k][2] = t.z; E.tprop[k][3] = t.w;
        }
    }
    if (c.goal_enabled)  // only the word holding this step's position; plain rows [word][env] so the access is 256 B per wave
        E.gw = reinterpret_cast<const unsigned*>(S)[((unsigned)c.L.goal + (unsigned)(goal_bit >> 3)) * (unsigned)N + (unsigned)e];
}

// `all`: after a reset (every group is new; may be called by a subset of the lanes).  Otherwise EVERY lane of the wave
// must call (the rarely-changing groups are written when any lane of the wave changed them -- a wave-wide vote);
// `valid` gates the stores.
__device__ __forceinline__ void store_gym(const DevCfg& c, float* __restrict__ S, long N, long e, const Env& E, int goal_bit,
                                          bool valid, bool all) {
    const int g0 = c.L.gym >> 2;
    bool d5 = true, d6 = true, d7 = true;
    if (c.metrics && !all) {
        const bool c5 = E.emin[0] != E.was_emin[0] || E.emin[1] != E.was_emin[1] || E.emin[2] != E.was_emin[2] || E.rise[0] != E.was_rise[0];
        const bool c6 = E.emax[0] != E.was_emax[0] || E.emax[1] != E.was_emax[1] || E.emax[2] != E.was_emax[2] || E.rise[1] != E.was_rise[1];
        const bool c7 = E.rise[2] != E.was_rise[2];
        d5 = __ballot(c5 && valid) != 0ull; d6 = __ballot(c6 && valid) != 0ull; d7 = __ballot(c7 && valid) != 0ull;
    }
    if (!valid) return;
    GROUP(S, N, g0 + 0, e) = make_float4(E.tgt[0], E.tgt[1], E.tgt[2], u2f((E.steps & 0xFFFFu) | (E.sft << 16)));
    GROUP(S, N, g0 + 1, e) = make_float4(u2f(E.flags), u2f(E.wcnt), u2f(E.gcnt[0]), u2f(E.gcnt[1]));
    if (c.metrics) {
        GROUP(S, N, g0 + 2, e) = make_float4(E.esum[0], E.esum[1], E.esum[2], E.perr[0]);
        GROUP(S, N, g0 + 3, e) = make_float4(E.eabs[0], E.eabs[1], E.eabs[2], E.perr[1]);
        GROUP(S, N, g0 + 4, e) = make_float4(E.sdcmd, E.perr[2], u2f(E.settle[0]), u2f(E.settle[1]));
        if (d5) GROUP(S, N, g0 + 5, e) = make_float4(E.emin[0], E.emin[1], E.emin[2], u2f(E.rise[0]));
        if (d6) GROUP(S, N, g0 + 6, e) = make_float4(E.emax[0], E.emax[1], E.emax[2], u2f(E.rise[1]));
        if (d7) GROUP(S, N, g0 + 7, e) = make_float4(u2f(E.rise[2]), E.e0[0], E.e0[1], E.e0[2]);
    }
    if (c.reward_potential) GROUP(S, N, g0 + 8, e) = make_float4(E.psh[0], E.psh[1], E.psh[2], 0.f);
    if (c.any_dynamic_target) {
#pragma unroll
        for (int k = 0; k < FWG_MAX_TARGETS; ++k)
            GROUP(S, N, (c.L.tprop >> 2) + k, e) = make_float4(E.tprop[k][0], E.tprop[k][1], E.tprop[k][2], E.tprop[k][3]);
    }
    if (c.goal_enabled)
        reinterpret_cast<unsigned*>(S)[((unsigned)c.L.goal + (unsigned)(goal_bit >> 3)) * (unsigned)N + (unsigned)e] = E.gw;
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
            const bool ok = fabsf(err[k]) <= V(c).target[k].bound;
            g |= ok ? (1u << k) : 0u;
            all = all && ok;
        }
    }
    return g | (all ? 8u : 0u);
}

// Push the goal flags of one record into the four windows (target0..2, all).  The windows are ONE ring of
// success_streak_req positions x 4 flag bits (8 positions per 32-bit word), addressed by the global step counter, so
// the nibble being overwritten is exactly the one that leaves the windows: the ones-count of the window is updated incrementally (no popcount over the ring), the cumulative
// count feeds success_time_frac, and the metric settling index latches the first record at which a full window
// satisfies the fraction (fixed_wing.py:1116-1128).
__device__ __forceinline__ unsigned window_count(const Env& E, int r) { return (E.wcnt >> (8 * r)) & 0xFFu; }
__device__ __forceinline__ void goal_push(const DevCfg& c, Env& E, unsigned g, int bit, unsigned rec_index) {
    const unsigned n_rec = rec_index + 1;
    const int sh = 4 * (bit & 7);
    unsigned present = 8u;
#pragma unroll
    for (int r = 0; r < 3; ++r) present |= (r < c.n_targets && c.target[r].has_bound) ? (1u << r) : 0u;
    // the nibble being overwritten was written streak_req records ago: it belongs to THIS episode only from record
    // streak_req on -- before that it is a leftover of an earlier episode and counts as empty (so a reset need not clear
    // the ring)
    const unsigned neu = g & present, old = rec_index >= (unsigned)c.streak_req ? (E.gw >> sh) & 0xFu : 0u;
    E.gw = (E.gw & ~(0xFu << sh)) | (neu << sh);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if ((present >> r) & 1u) {
            const unsigned f = (neu >> r) & 1u;
            E.wcnt += (f - ((old >> r) & 1u)) << (8 * r);   // per-byte add/subtract; each byte stays within [0, 128]
            E.gcnt[r >> 1] += f << (16 * (r & 1));
            if (c.metrics && pack16_get(E.settle, r) == 0xFFFFu && n_rec >= (unsigned)c.streak_req &&
                window_count(E, r) >= (unsigned)V(c).streak_min_count)
                pack16_set(E.settle, r, rec_index);
        }
    }
}

// sample_target (fixed_wing.py:461-521); `given` (nullable) holds explicit targets for reset(target=...)
template <class TAB>
__device__ __forceinline__ void sample_targets(const DevCfg& c, DynCfgK& dc, const KArgs& A, long e, Env& E,
                                               const TAB& T, const float* given) {
    const unsigned env_id = (unsigned)(A.env_base + e);
    const unsigned resample = (E.flags & ~FWG_FLAG_LAST_FAILED) >> FWG_FLAG_RESAMPLE_SHIFT;
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
                nt[k] = va + (slope * (-pitch_tgt) - 0.25f) * V(c).dt;
            } else if (pt >= 0.08726646259971647f) {  // radians(5)
                const float va_end = 26.27f - 41.2529f * pt;
                if (va > va_end) nt[k] = (E.sft < 750u) ? va + (va_end - va) * (1.f / 150.f) : va_end;
            }
        } else if (c.any_dynamic_target && t.cls == FWG_TGT_LINEAR) {
            if (E.tprop[k][1] >= 0.f) nt[k] = E.tgt[k] + E.tprop[k][0] * V(c).dt;
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
// `ring` = this lane's entry of the LDS action window [slot][lane][4] (streamed in with 16-byte global_load_lds).
__device__ __forceinline__ float backscale_action(const DevCfg& c, int ai, float actuator) {
    if (c.scale_actions)
        return (V(c).scale_high - V(c).scale_low) * (actuator - V(c).act_to_low[ai]) * V(c).inv_act_span[ai] + V(c).scale_low;
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
            s += fabsf(ring[s_new * (4 * FWG_WAVE) + ai] - ring[s_old * (4 * FWG_WAVE) + ai]);
        }
    }
    return s;
}

// the same sum over a window held in registers by age (win[k] = the entry of k steps ago, win[0] = this step's), n_act >= 1
__device__ __forceinline__ float action_obs_win(const DevCfg& c, const float (&win)[FWG_MAX_WINDOW][3], int ai, int w, unsigned n_act) {
    const int W = c.L.window;
    const int m = (int)min(n_act, (unsigned)w);
    float s = 0.f;
#pragma unroll
    for (int k = FWG_MAX_WINDOW - 2; k >= 0; --k) {
        if (k <= W - 2 && k <= m - 2) {
            const float a = ai == 0 ? win[k][0] : (ai == 1 ? win[k][1] : win[k][2]);
            const float b = ai == 0 ? win[k + 1][0] : (ai == 1 ? win[k + 1][1] : win[k + 1][2]);
            s += fabsf(a - b);
        }
    }
    return s;
}

// newest observation row (un-noised, normalised) into ob[0, n_obs) and, when `push`, into the lag ring (AoS record)
// int_pad (reset only, integration_window): the integrator entries of the PUSHED record -- the source of later padding rows,
// (W + 1) e0 -- differ from those of the reset observation itself (which show the previous episode's sums, see reset_finish)
template <class TAB, class OB>
__device__ __forceinline__ void build_row0(const DevCfg& c, const KArgs& A, long e, const Env& E, const TAB& T, OB& ob,
                                           const float* ring, int ring_slot, bool push, int act_slot,
                                           const float* pre_action = nullptr, const float* int_pad = nullptr) {
    float padv[FWG_MAX_OBS];
#pragma unroll
    for (int j = 0; j < FWG_MAX_OBS; ++j) {
        padv[j] = 0.f;
        if (j < c.n_obs) {
            const DevObs& o = c.obs[j]; const DevObs& ov = V(c).obs[j];
            float v;
            if (o.type == FWG_OBS_STATE) v = T.get(o.src);
            else if (o.type == FWG_OBS_TARGET_RELATIVE) v = T.get(FWG_TAB_ERR + o.src);
            else if (o.type == FWG_OBS_TARGET_ABSOLUTE) v = T.get(FWG_TAB_TGT + o.src);
            else if (o.type == FWG_OBS_TARGET_INTEGRATOR) v = T.get(FWG_TAB_INT + o.src);
            else v = pre_action != nullptr ? pre_action[j]   // already summed while the integration ran (step kernel)
                                           : action_obs(c, ring, o.src, o.window, E.steps, act_slot, T.get(FWG_V_ELEVATOR + o.src));
            float vp = (int_pad != nullptr && o.type == FWG_OBS_TARGET_INTEGRATOR) ? int_pad[o.src] : v;
            if (o.norm) { v = (v - ov.mean) * ov.inv_var; vp = (vp - ov.mean) * ov.inv_var; }
            ob.put(j, v);
            padv[j] = vp;
        }
    }
    if (push && c.obs_length > 1) {  // the record enters the lag ring as ceil(n_obs/4) 16-byte groups
        const int ng = c.L.lag_groups;
#pragma unroll
        for (int g = 0; g < FWG_MAX_OBS / 4; ++g)
            if (g < ng)
                store_group_once(A.S, A.N, (c.L.lag_ring >> 2) + ring_slot * ng + g, e,
                                 make_float4(padv[4 * g], 4 * g + 1 < c.n_obs ? padv[4 * g + 1] : 0.f,
                                             4 * g + 2 < c.n_obs ? padv[4 * g + 2] : 0.f, 4 * g + 3 < c.n_obs ? padv[4 * g + 3] : 0.f));
    }
}

// lagged rows r >= 1 = the records pushed r*obs_step steps ago: streamed HBM -> LDS at kernel start
// (stream_lag_rows, global_load_lds: no VGPRs while the physics runs), collected into the record here
__device__ __forceinline__ void stream_lag_rows(const DevCfg& c, const KArgs& A, long e, float* lds_lag) {
    const int ng = c.L.lag_groups;
    for (int r = 1; r < c.obs_length; ++r)
        for (int g = 0; g < ng; ++g)
            dma_group_once(&CGROUP(A.S, A.N, (c.L.lag_ring >> 2) + A.lag_slots[r] * ng + g, e), lds_lag + ((r - 1) * ng + g) * (4 * FWG_WAVE));
}
// (keep_from: rows whose lag reaches this many steps back are padding rows this lane has prepared already -- early_rows_pre)
template <class OB>
__device__ __forceinline__ void load_lag_rows(const DevCfg& c, const float* lag_lane, OB& ob, int keep_from = 1 << 30) {
    const int ng = c.L.lag_groups;
#pragma unroll
    for (int r = 1; r < FWG_MAX_ROWS; ++r) {
        if (r < c.obs_length && r * c.obs_step < keep_from) {
#pragma unroll
            for (int g = 0; g < FWG_MAX_OBS / 4; ++g) {
                if (g < ng) {
                    const float4 q = *reinterpret_cast<const float4*>(lag_lane + ((r - 1) * ng + g) * (4 * FWG_WAVE));
                    ob.put(r * c.n_obs + 4 * g, q.x);
                    if (4 * g + 1 < c.n_obs) ob.put(r * c.n_obs + 4 * g + 1, q.y);
                    if (4 * g + 2 < c.n_obs) ob.put(r * c.n_obs + 4 * g + 2, q.z);
                    if (4 * g + 3 < c.n_obs) ob.put(r * c.n_obs + 4 * g + 3, q.w);
                }
            }
        }
    }
}

// Values requested long before their use (a foreseen episode end's prefetches) are "touched" where they have surely landed:
// the compiler places its s_waitcnt at the touch, where nothing younger is in flight, instead of at the first use -- there,
// after divergent code that issued a number of stores it cannot count, it falls back to vmcnt(0) and the wave sits through
// the acknowledgement of every store it has just issued (2-3k ticks on the episode-end path, tools/timeline.py).
// Likewise a load inside a RARE branch whose result is used after the join: untouched, the wait lands at the use, on the
// path of every wave, and counts the stores issued before the branch (fix_lagged_rows: the dense batch paid 2-4k ticks per
// step for the lag-ring reads of lanes that almost never exist)
#ifdef FWG_EMU
#define FWG_TOUCH(x) ((void)(x))
#else
#define FWG_TOUCH(x) asm volatile("" ::"v"(x))
#endif
// entry j of the record of env e in ring slot `slot` (rare per-lane fix-up reads)
__device__ __forceinline__ float lag_entry(const DevCfg& c, const KArgs& A, long e, int slot, int j) {
    return A.S[(((unsigned)(c.L.lag_ring >> 2) + (unsigned)(slot * c.L.lag_groups + (j >> 2))) * (unsigned)A.N + (unsigned)e) * 4u + (unsigned)(j & 3)];
}

// Fix-ups of the lagged rows r >= 1 that the plain ring read cannot provide (fixed_wing.py:790-832):
//  * rows reaching back to (or before) the start of the episode, i = 1 + r*step > steps_count: the INITIAL record plus
//    a fresh U(-1,1)*dt per row, with "action" entries replaced by the CURRENT actuator value;
//  * after a failed simulator step the state/target histories are one record shorter than the action history, so
//    the non-action entries come from one slot further back.
// The step kernel splits the first case (row-log mode): early_rows_pre computes everything that does not need this step's
// state -- the initial record, the per-row noise -- before the integration has finished; fix_lagged_rows(pre = true) then
// only fills in the "action" entries.
// the per-row uniform bits of the "initial record" noise: one Philox block per four rows (stream FWG_STREAM_INIT_NOISE + 256 q,
// component r & 3), each block computed once
// STEPPED (every draw after the reset's, c1 = steps >= 1): row 0 is never a padding row then, so row r takes component r - 1
// -- a window of up to five rows needs ONE Philox block per step instead of two (oracle PhiloxStream.init_noise)
template <bool STEPPED>
__device__ __forceinline__ void init_noise_bits(const DevCfg& c, unsigned env_id, unsigned c1, unsigned c2, unsigned seed_lo,
                                                unsigned seed_hi, unsigned (&bits)[FWG_MAX_ROWS]) {
    constexpr int SH = STEPPED ? 1 : 0;
    if (STEPPED) bits[0] = 0u;
#pragma unroll
    for (int q = 0; q < (FWG_MAX_ROWS + 3) / 4; ++q) {
        u4 b = u4{0u, 0u, 0u, 0u};
        if (q * 4 + SH < c.obs_length) b = philox4x32(env_id, c1, c2, FWG_STREAM_INIT_NOISE + 256u * q, seed_lo, seed_hi);
        if (4 * q + SH < FWG_MAX_ROWS) bits[4 * q + SH] = b.x;
        if (4 * q + 1 + SH < FWG_MAX_ROWS) bits[4 * q + 1 + SH] = b.y;
        if (4 * q + 2 + SH < FWG_MAX_ROWS) bits[4 * q + 2 + SH] = b.z;
        if (4 * q + 3 + SH < FWG_MAX_ROWS) bits[4 * q + 3 + SH] = b.w;
    }
}
// the per-row noise U(-1, 1) dt of the padding rows of an env `t` steps into its episode (0 for rows that are not padding)
__device__ __forceinline__ void early_row_noise(const DevCfg& c, const KArgs& A, long e, unsigned steps, unsigned episode,
                                                float (&row_noise)[FWG_MAX_ROWS]) {
    unsigned bits[FWG_MAX_ROWS];
    init_noise_bits<true>(c, (unsigned)(A.env_base + e), steps, episode, A.seed_lo, A.seed_hi, bits);
#pragma unroll
    for (int r = 0; r < FWG_MAX_ROWS; ++r) {
        row_noise[r] = 0.f;
        if (r >= 1 && r < c.obs_length && r * c.obs_step >= (int)steps) row_noise[r] = rounded((2.f * u01(bits[r]) - 1.f) * V(c).dt);
    }
}
// the padding rows of one env from its record 0 and the per-row noise, straight into the row log (k_step2, physics wave,
// after the integration): "action" entries = the CURRENT actuator value, back-scaled (fixed_wing.py:814-820)
__device__ __forceinline__ void early_rows_to_log(const DevCfg& c, const KArgs& A, long e, unsigned steps, const float (&rec)[FWG_MAX_OBS],
                                                  const float (&row_noise)[FWG_MAX_ROWS], const float (&actuator)[3], long long win) {
#pragma unroll
    for (int r = 1; r < FWG_MAX_ROWS; ++r) {
        if (r >= c.obs_length || r * c.obs_step < (int)steps) continue;
        const float noise = row_noise[r];
        float v[FWG_MAX_OBS];
#pragma unroll
        for (int j = 0; j < FWG_MAX_OBS; ++j) {
            if (j >= c.n_obs) continue;
            const DevObs& o = c.obs[j]; const DevObs& ov = V(c).obs[j];
            if (o.type == FWG_OBS_ACTION) {
                v[j] = backscale_action(c, o.src, actuator[o.src]) + noise;
                if (o.norm) v[j] = (v[j] - ov.mean) * ov.inv_var;
            } else {
                v[j] = rec[j] + noise * (o.norm ? ov.inv_var : 1.f);
            }
        }
        float* dst = log_row(c, A.obs, A.N, e, win + r);
        if ((c.n_obs & 3) == 0) {
#pragma unroll
            for (int q = 0; q < FWG_MAX_OBS / 4; ++q)
                if (4 * q < c.n_obs) reinterpret_cast<float4*>(dst)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        } else {
#pragma unroll
            for (int j = 0; j < FWG_MAX_OBS; ++j)
                if (j < c.n_obs) dst[j] = v[j];
        }
    }
}
// (slot: where the episode's record 0 sits -- row-log mode: slot 0 of the one-slot ring; dense batch: `steps` slots behind the current one)
__device__ __forceinline__ void early_rows_request(const DevCfg& c, const KArgs& A, long e, float (&rec)[FWG_MAX_OBS], int slot = 0) {
#pragma unroll
    for (int g = 0; g < FWG_MAX_OBS / 4; ++g) {
        if (g < c.L.lag_groups) {
            const float4 q = CGROUP(A.S, A.N, (c.L.lag_ring >> 2) + slot * c.L.lag_groups + g, e);
            rec[4 * g] = q.x; rec[4 * g + 1] = q.y; rec[4 * g + 2] = q.z; rec[4 * g + 3] = q.w;
        }
    }
}
template <class OB>
__device__ __forceinline__ void early_rows_pre(const DevCfg& c, const KArgs& A, long e, const Env& E, OB& ob, const float (&rec)[FWG_MAX_OBS],
                                               float (&row_noise)[FWG_MAX_ROWS]) {
    const int t = (int)E.steps;
    unsigned bits[FWG_MAX_ROWS];
    init_noise_bits<true>(c, (unsigned)(A.env_base + e), E.steps, E.episode, A.seed_lo, A.seed_hi, bits);
#pragma unroll
    for (int r = 1; r < FWG_MAX_ROWS; ++r) {
        row_noise[r] = 0.f;
        if (r >= c.obs_length || r * c.obs_step < t) continue;
        const float noise = rounded((2.f * u01(bits[r]) - 1.f) * V(c).dt);
        row_noise[r] = noise;
#pragma unroll
        for (int j = 0; j < FWG_MAX_OBS; ++j)
            if (j < c.n_obs && c.obs[j].type != FWG_OBS_ACTION)
                ob.put(r * c.n_obs + j, rec[j] + noise * (c.obs[j].norm ? V(c).obs[j].inv_var : 1.f));
    }
}
template <class TAB, class OB>
__device__ __forceinline__ void fix_lagged_rows(const DevCfg& c, const KArgs& A, long e, const Env& E, const TAB& T, OB& ob,
                                                bool ok, bool pre, const float (&pre_noise)[FWG_MAX_ROWS]) {
    const int depth = c.L.lag_depth;
    const int t = (int)E.steps;
    unsigned bits[FWG_MAX_ROWS] = {};
    if (!pre && t <= (c.obs_length - 1) * c.obs_step) init_noise_bits<true>(c, (unsigned)(A.env_base + e), E.steps, E.episode, A.seed_lo, A.seed_hi, bits);
#pragma unroll
    for (int r = 1; r < FWG_MAX_ROWS; ++r) {
        if (r >= c.obs_length) continue;
        const int lag = r * c.obs_step;
        if (lag >= t && pre) {
#pragma unroll
            for (int j = 0; j < FWG_MAX_OBS; ++j) {
                if (j >= c.n_obs || c.obs[j].type != FWG_OBS_ACTION) continue;
                const DevObs& o = c.obs[j]; const DevObs& ov = V(c).obs[j];
                float v = backscale_action(c, o.src, T.get(FWG_V_ELEVATOR + o.src)) + pre_noise[r];
                if (o.norm) v = (v - ov.mean) * ov.inv_var;
                ob.put(r * c.n_obs + j, v);
            }
        } else if (lag >= t) {
            const float noise = rounded((2.f * u01(bits[r]) - 1.f) * V(c).dt);
            int slot0 = A.slot_lag - t; slot0 += (slot0 < 0) ? depth : 0;  // ring slot of the episode's record 0
            if (c.obs_log > 0) slot0 = 0;   // row-log mode: the ring has one slot and holds exactly that record
#pragma unroll
            for (int j = 0; j < FWG_MAX_OBS; ++j) {
                if (j >= c.n_obs) continue;
                const DevObs& o = c.obs[j]; const DevObs& ov = V(c).obs[j];
                float v;
                if (o.type == FWG_OBS_ACTION) {
                    v = backscale_action(c, o.src, T.get(FWG_V_ELEVATOR + o.src)) + noise;
                    if (o.norm) v = (v - ov.mean) * ov.inv_var;
                } else {
                    const float q = lag_entry(c, A, e, slot0, j);
                    FWG_TOUCH(q);   // (waited for HERE, inside the rare branch)
                    v = q + noise * (o.norm ? ov.inv_var : 1.f);
                }
                ob.put(r * c.n_obs + j, v);
            }
        } else if (!ok) {
            int slot = A.slot_lag - 1 - lag; slot += (slot < 0) ? depth : 0;
            const float* older = c.obs_log > 0 ? A.obs + ((log_plane(c.obs_step, c.obs_log, c.obs_length, A.gnow, A.gnow - 1 - lag) * A.N + e) * c.n_obs)
                                               : nullptr;
#pragma unroll
            for (int j = 0; j < FWG_MAX_OBS; ++j)
                if (j < c.n_obs && c.obs[j].type != FWG_OBS_ACTION) {
                    // (the episode's record 0 sits in the log with its reset-time noise; the clean copy is in the ring slot)
                    const float q = c.obs_log > 0 ? (lag == t - 1 ? lag_entry(c, A, e, 0, j) : older[j]) : lag_entry(c, A, e, slot, j);
                    FWG_TOUCH(q);
                    ob.put(r * c.n_obs + j, q);
                }
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
            if (k < c.obs_dim) ob.put(k, ob.get(k) + V(c).obs_noise_mean + V(c).obs_noise_std * n[i]);
        }
    }
}

// The 64 observation records of this wave -> out[env0 .. env0+63][obs_dim], which is one contiguous block of the
// [N][obs_dim] batch: every lane parks its record in the LDS staging area ([lane][obs_dim]) and the wave then writes
// the block in linear order, 1 KiB per store instruction (16 B per lane) when the record size allows.  `lanes` selects
// the records to write (all, or the finished episodes for the terminal observations).  Must be called by all lanes.
// The observation batch is never read back by the kernels: streaming (non-temporal) stores keep it from displacing the
// state arena, which the next launch re-reads, out of the L2 / Infinity Cache -- unless a rollout head is attached (`reread`),
// which reads exactly this batch in the very next launch: plain stores then.
__device__ __forceinline__ void stream_store4(float4* p, float4 v) {
    fwg_v4f x;
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    __builtin_nontemporal_store(x, reinterpret_cast<fwg_v4f*>(p));
}
// ROLE 0: the workgroup is one wave and uses the workgroup barrier; otherwise (k_step2) only THIS wave touches the staging area
template <int ROLE> __device__ __forceinline__ void obs_sync() {
    if (ROLE == 0) __syncthreads(); else FWG_WAVE_SYNC();
}
template <int ROLE, class OB>
__device__ __forceinline__ void write_obs(const DevCfg& c, float* __restrict__ out, long env0, long N, const OB& ob,
                                          float* stage, int lane, unsigned long long lanes, bool reread = false) {
    const int D = c.obs_dim;
    obs_sync<ROLE>();  // the staging area may still be read by a previous call
    if (obs_vec4(D)) {
        float4* mine = reinterpret_cast<float4*>(stage + lane * D);
#pragma unroll
        for (int q = 0; q < (FWG_MAX_OBS * FWG_MAX_ROWS) / 4; ++q)
            if (q * 4 < D) mine[q] = make_float4(ob.get(4 * q), ob.get(4 * q + 1), ob.get(4 * q + 2), ob.get(4 * q + 3));
        obs_sync<ROLE>();
        const float4* all = reinterpret_cast<const float4*>(stage);
        float4* o4 = reinterpret_cast<float4*>(out + env0 * D);
        const int total4 = FWG_WAVE * D / 4;
#ifdef FWG_ABL_OBS_LOOP
        if (false) {
#else
        if (OB::in_regs) {
#endif
            // specialised kernels (D a constant): ALL the reads, then all the stores -- as a loop of read, wait,
            // store the fifteen 1 KiB rows of a 60-entry batch cost one LDS round trip EACH, ~4k ticks per wave and step
            constexpr int NQ = (FWG_MAX_OBS * FWG_MAX_ROWS) / 4;
            float4 buf[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (q * 4 < D) buf[q] = all[lane + q * FWG_WAVE];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q * 4 < D) {
                    const int i = lane + q * FWG_WAVE, l = (4 * i) / D;
                    if (((lanes >> l) & 1ull) && env0 + l < N) { if (reread) o4[i] = buf[q]; else stream_store4(o4 + i, buf[q]); }
                }
            }
        } else {
#pragma unroll 4
            for (int i = lane; i < total4; i += FWG_WAVE) {
                const int l = (4 * i) / D;
                if (((lanes >> l) & 1ull) && env0 + l < N) { if (reread) o4[i] = all[i]; else stream_store4(o4 + i, all[i]); }
            }
        }
    } else {
        const int Ds = obs_stage_stride(D);
#pragma unroll
        for (int k = 0; k < FWG_MAX_OBS * FWG_MAX_ROWS; ++k)
            if (k < D) stage[lane * Ds + k] = ob.get(k);
        obs_sync<ROLE>();
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

// ---- finished-episode record.  An episode end costs the wave that hosts it the length of everything its ONE ending lane
// does, so the step kernel only parks the episode's raw accumulators (7 groups, section L.fin) and marks the env
// (FWG_FLAG_FIN_PENDING); the episodic metrics (get_metric, fixed_wing.py:1095-1162) and the success sums are computed from
// the record by k_finish (fwg_finish_episodes / fwg_reduce_success*), off the step's critical path.
struct FinRec {
    float e0[3], esum[3], eabs[3], emin[3], emax[3], end_sum[3];   // end_sum: sum of the last <= 50 errors
    unsigned rise[3], settle[2], gcnt[2];
    float sdcmd;
    unsigned steps, n_rec;   // steps taken; records in the episode histories (steps + 1 unless the last step failed)
};
__device__ __forceinline__ void fin_store(const DevCfg& c, float* __restrict__ S, long N, long e, const FinRec& R) {
    const int g0 = c.L.fin >> 2;
    GROUP(S, N, g0 + 0, e) = make_float4(R.e0[0], R.e0[1], R.e0[2], u2f(R.steps | ((R.n_rec - R.steps) << 16)));
    GROUP(S, N, g0 + 1, e) = make_float4(R.esum[0], R.esum[1], R.esum[2], R.sdcmd);
    GROUP(S, N, g0 + 2, e) = make_float4(R.eabs[0], R.eabs[1], R.eabs[2], u2f(R.settle[0]));
    GROUP(S, N, g0 + 3, e) = make_float4(R.emin[0], R.emin[1], R.emin[2], u2f(R.settle[1]));
    GROUP(S, N, g0 + 4, e) = make_float4(R.emax[0], R.emax[1], R.emax[2], u2f(R.gcnt[0]));
    GROUP(S, N, g0 + 5, e) = make_float4(R.end_sum[0], R.end_sum[1], R.end_sum[2], u2f(R.gcnt[1]));
    GROUP(S, N, g0 + 6, e) = make_float4(u2f(R.rise[0]), u2f(R.rise[1]), u2f(R.rise[2]), 0.f);
}
__device__ __forceinline__ void fin_load(const DevCfg& c, const float* __restrict__ S, long N, long e, FinRec& R) {
    const int g0 = c.L.fin >> 2;
    float4 q[7];
#pragma unroll
    for (int g = 0; g < 7; ++g) q[g] = CGROUP(S, N, g0 + g, e);
    R.e0[0] = q[0].x; R.e0[1] = q[0].y; R.e0[2] = q[0].z; R.steps = f2u(q[0].w) & 0xFFFFu; R.n_rec = R.steps + (f2u(q[0].w) >> 16);
    R.esum[0] = q[1].x; R.esum[1] = q[1].y; R.esum[2] = q[1].z; R.sdcmd = q[1].w;
    R.eabs[0] = q[2].x; R.eabs[1] = q[2].y; R.eabs[2] = q[2].z; R.settle[0] = f2u(q[2].w);
    R.emin[0] = q[3].x; R.emin[1] = q[3].y; R.emin[2] = q[3].z; R.settle[1] = f2u(q[3].w);
    R.emax[0] = q[4].x; R.emax[1] = q[4].y; R.emax[2] = q[4].z; R.gcnt[0] = f2u(q[4].w);
    R.end_sum[0] = q[5].x; R.end_sum[1] = q[5].y; R.end_sum[2] = q[5].z; R.gcnt[1] = f2u(q[5].w);
    R.rise[0] = f2u(q[6].x); R.rise[1] = f2u(q[6].y); R.rise[2] = f2u(q[6].z);
}
// the metrics block column and the contributions to the success sums of one finished episode
__device__ __forceinline__ void finish_metrics(const DevCfg& c, const FinRec& R, float (&mt)[FWG_N_METRICS], float (&red)[FWG_N_REDUCE]) {
#pragma unroll
    for (int i = 0; i < FWG_N_METRICS; ++i) mt[i] = NAN;
#pragma unroll
    for (int i = 0; i < FWG_N_REDUCE; ++i) red[i] = 0.f;
    const int end_cnt = (int)min(R.n_rec, (unsigned)FWG_END_WINDOW);
#pragma unroll
    for (int k = 0; k < FWG_MAX_TARGETS; ++k) {
        if (k >= c.n_targets) continue;
        const unsigned lo = R.rise[k] & 0xFFFFu, hi = R.rise[k] >> 16;
        mt[FWG_M_RISE_TIME + k] = (lo == 0xFFFFu || hi == 0xFFFFu) ? NAN : (float)lo - (float)hi;
        const float ext = R.e0[k] > 0.f ? R.emin[k] : R.emax[k];
        mt[FWG_M_OVERSHOOT + k] = (fsignf(ext) == fsignf(R.e0[k])) ? NAN : fabsf(fast_div(ext, R.e0[k]));
        mt[FWG_M_TOTAL_ERROR + k] = R.eabs[k];
        mt[FWG_M_AVG_ERROR + k] = fabsf(R.e0[k]) >= 0.01f ? fabsf(fast_div(R.esum[k], (float)R.n_rec * R.e0[k])) : NAN;
        mt[FWG_M_END_ERROR + k] = fabsf(fast_div(R.end_sum[k], (float)end_cnt));
    }
    mt[FWG_M_CONTROL_VARIATION] = fast_div(R.sdcmd, 3.f * V(c).dt * (float)(R.steps - 1u));
    if (c.goal_enabled) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool present = (r == 3) || (r < c.n_targets && c.target[r < 3 ? r : 0].has_bound);
            if (present) {
                const unsigned st = pack16_get(R.settle, r);
                mt[FWG_M_SETTLING_TIME + r] = st == 0xFFFFu ? NAN : (float)st;
                mt[FWG_M_SUCCESS + r] = st == 0xFFFFu ? 0.f : 1.f;
                mt[FWG_M_SUCCESS_TIME_FRAC + r] = fast_div((float)pack16_get(R.gcnt, r), (float)R.n_rec);
            }
        }
    }
    red[0] = 1.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[1 + r] = mt[FWG_M_SUCCESS + r] == 1.f ? 1.f : 0.f;
        red[12 + r] = mt[FWG_M_SUCCESS_TIME_FRAC + r] == mt[FWG_M_SUCCESS_TIME_FRAC + r] ? mt[FWG_M_SUCCESS_TIME_FRAC + r] : 0.f;
    }
    red[5] = mt[FWG_M_CONTROL_VARIATION] == mt[FWG_M_CONTROL_VARIATION] ? mt[FWG_M_CONTROL_VARIATION] : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        red[6 + k] = mt[FWG_M_END_ERROR + k] == mt[FWG_M_END_ERROR + k] ? mt[FWG_M_END_ERROR + k] : 0.f;
        red[9 + k] = mt[FWG_M_TOTAL_ERROR + k] == mt[FWG_M_TOTAL_ERROR + k] ? mt[FWG_M_TOTAL_ERROR + k] : 0.f;
    }
}
// success sums: counts (0..4) as integers, the rest in 2^-20 fixed point (exact, order-free)
__device__ __forceinline__ unsigned long long reduce_fixed(int i, float v) {
    return (unsigned long long)(long long)rintf(i < 5 ? v : v * FWG_ACC_SCALE);
}
// one lane collects its own pending record (rare: the env ends a second episode before any fwg_finish_episodes).
// (measured: as a real -- noinline, cold -- function it made EVERY launch 40 % slower, 16.8 instead of 12.1 us per C3 step:
// a call anywhere in the kernel brings the stack set-up to every wave's entry, 0.7k -> 4.7k ticks)
__device__ __forceinline__ void fin_collect_pending(const DevCfg& c, const KArgs& A, long e) {
    FinRec R;
    fin_load(c, A.S, A.N, e, R);
    float mt[FWG_N_METRICS], red[FWG_N_REDUCE];
    finish_metrics(c, R, mt, red);
    if (A.metrics != nullptr) {
        for (int i = 0; i < FWG_N_METRICS; ++i) A.metrics[(unsigned)i * (unsigned)A.N + (unsigned)e] = mt[i];
    }
    for (int i = 0; i < FWG_N_REDUCE; ++i)
        if (red[i] != 0.f) atomicAdd(A.reduce + i, reduce_fixed(i, red[i]));
}

// FixedWingAircraft.reset (fixed_wing.py:287-336) for one lane, in two parts:
//   reset_sample : everything that depends only on (env id, episode index, seed) and the optional given values -- the
//                  sampled initial state and targets, the derived angles, the per-row initial noise.  Pure (no memory
//                  writes), Philox-heavy.  The step kernel runs it for lanes that reach steps_max WHILE the physics wave
//                  still integrates (the draw does not depend on how the episode ends).
//   reset_finish : installs the draw in E, initialises the bookkeeping, writes the ring slots that hold initial records
//                  and builds the observation record ob (all rows).  `g_*` = ring positions of the LAST completed step.
// reset_env = both back to back (reset kernel; episode ends that were not foreseen).
struct ResetDraw {
    float gust_gain;
    float y[NY], wind[3];
    Derived d;
    float tgt[FWG_MAX_TARGETS], tprop[FWG_MAX_TARGETS][4];
    unsigned flags, episode;
    float row_noise[FWG_MAX_ROWS];
};

// ---- the draw in pieces, so that it can be computed in one go (reset kernel, unforeseen situations) or one piece per env
// step, off the critical path (draw_stage_step below).  Identical arithmetic either way.
// (1) sampled initial values of the variables [4 BLK0, 4 (BLK0 + NBLK)): given values or U(init_min, init_max)
template <int BLK0, int NBLK>
__device__ __forceinline__ void draw_state_values(DynCfgK& dc, const KArgs& A, long e, unsigned episode_new,
                                                  float (&v0)[FWG_N_RESET_VARS + 3]) {
    const unsigned env_id = (unsigned)(A.env_base + e);
#pragma unroll
    for (int blk = BLK0; blk < BLK0 + NBLK; ++blk) {
        const u4 b = philox4x32(env_id, episode_new, (unsigned)blk, FWG_STREAM_RESET_STATE, A.seed_lo, A.seed_hi);
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
}
// (2a) state vector and derived angles from the initial values
__device__ __forceinline__ void draw_state(const DevCfg& c, const float (&v0)[FWG_N_RESET_VARS + 3], ResetDraw& D) {
    {
        float sr, cr, sp, cp, sy, cy;
        sincosf(0.5f * v0[FWG_V_ROLL], &sr, &cr);
        sincosf(0.5f * v0[FWG_V_PITCH], &sp, &cp);
        sincosf(0.5f * v0[FWG_V_YAW], &sy, &cy);
        D.y[0] = cy * cp * cr + sy * sp * sr; D.y[1] = cy * cp * sr - sy * sp * cr;
        D.y[2] = cy * sp * cr + sy * cp * sr; D.y[3] = sy * cp * cr - cy * sp * sr;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) D.y[4 + i] = v0[FWG_V_OMEGA_P + i];
    {
        const float el = fclampf(v0[FWG_V_ELEVATOR], V(c).val_min[FWG_V_ELEVATOR], V(c).val_max[FWG_V_ELEVATOR]);
        const float ai = fclampf(v0[FWG_V_AILERON], V(c).val_min[FWG_V_AILERON], V(c).val_max[FWG_V_AILERON]);
        D.y[13] = el - ai; D.y[14] = el + ai;
        D.y[15] = fclampf(v0[FWG_V_THROTTLE], V(c).val_min[FWG_V_THROTTLE], V(c).val_max[FWG_V_THROTTLE]);
        D.y[16] = 0.f; D.y[17] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) D.wind[i] = v0[FWG_V_WIND_N + i];
    const float gust0[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    D.d = derive<false>(D.y, D.wind, gust0);
}
// (2b) the sampled targets for that state (D.y, D.wind, D.d set)
template <class TAB>
__device__ __forceinline__ void draw_targets(const DevCfg& c, DynCfgK& dc, const KArgs& A, long e, unsigned episode_new, TAB& T, ResetDraw& D) {
    Env R;   // scratch: only the fields sample_targets / fill_vars touch
    R.episode = episode_new;
    R.steps = 0u;
    R.flags = 0u;   // prev_shaping := None, resample counter := 0 (the sticky goal bit is merged by reset_finish)
#pragma unroll
    for (int i = 0; i < NY; ++i) R.y[i] = D.y[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) R.wind[i] = D.wind[i];
    R.d = D.d;
    fill_vars(R, T);
    sample_targets(c, dc, A, e, R, T, A.init_target);
    D.gust_gain = 1.f;
    if (c.sim_keys) {   // simulator.turbulence / turbulence_intensity (fixed_wing.py:560-569): one uniform per sampled key
        const unsigned env_id = (unsigned)(A.env_base + e);
        float on = dc.sk_base_gain, g = 1.f;
        if (dc.sk_n_turb > 0) {
            const float u = u01(philox4x32(env_id, episode_new, (unsigned)dc.sk_idx_turb, FWG_STREAM_SIM_KEY, A.seed_lo, A.seed_hi).x);
            on = dc.sk_on_turb[dc.sk_n_turb - 1];
            for (int i = dc.sk_n_turb - 2; i >= 0; --i) on = u < dc.sk_cum_turb[i] ? dc.sk_on_turb[i] : on;
        }
        if (dc.sk_n_int > 0) {
            const float u = u01(philox4x32(env_id, episode_new, (unsigned)dc.sk_idx_int, FWG_STREAM_SIM_KEY, A.seed_lo, A.seed_hi).x);
            g = dc.sk_gain_int[dc.sk_n_int - 1];
            for (int i = dc.sk_n_int - 2; i >= 0; --i) g = u < dc.sk_cum_int[i] ? dc.sk_gain_int[i] : g;
        }
        D.gust_gain = on * g;
    }
#pragma unroll
    for (int k = 0; k < FWG_MAX_TARGETS; ++k) {
        D.tgt[k] = R.tgt[k];
#pragma unroll
        for (int i = 0; i < 4; ++i) D.tprop[k][i] = c.any_dynamic_target ? R.tprop[k][i] : 0.f;
    }
    D.flags = R.flags; D.episode = R.episode;
}
template <class TAB>
__device__ __forceinline__ void draw_state_and_targets(const DevCfg& c, DynCfgK& dc, const KArgs& A, long e, unsigned episode_new,
                                                       const float (&v0)[FWG_N_RESET_VARS + 3], TAB& T, ResetDraw& D) {
    draw_state(c, v0, D);
    draw_targets(c, dc, A, e, episode_new, T, D);
}
// (3) per-row initial noise of the lagged rows (fixed_wing.py:792-795,831-832)
__device__ __forceinline__ void draw_row_noise(const DevCfg& c, const KArgs& A, long e, unsigned episode_new, ResetDraw& D) {
    unsigned bits[FWG_MAX_ROWS] = {};
    if (c.obs_length > 1) init_noise_bits<false>(c, (unsigned)(A.env_base + e), 0u, episode_new, A.seed_lo, A.seed_hi, bits);
#pragma unroll
    for (int r = 0; r < FWG_MAX_ROWS; ++r) {
        D.row_noise[r] = 0.f;
        if (c.obs_length > 1 && r < c.obs_length) D.row_noise[r] = (2.f * u01(bits[r]) - 1.f) * V(c).dt;
    }
}

template <class TAB>
__device__ __forceinline__ void reset_sample(const DevCfg& c, DynCfgK& dc, const KArgs& A, long e, unsigned episode_old,
                                             unsigned flags_old, TAB& T, ResetDraw& D) {
    (void)flags_old;
    float v0[FWG_N_RESET_VARS + 3];
    draw_state_values<0, 6>(dc, A, e, episode_old + 1u, v0);
    draw_state_and_targets(c, dc, A, e, episode_old + 1u, v0, T, D);
    draw_row_noise(c, A, e, episode_old + 1u, D);
}

// ---- The NEXT episode's draw, prepared ahead of time.  After every reset the step kernel computes the draw of the episode
// that will follow, one piece per env step in the gym wave's idle time before the barrier (4 steps), into a cold arena
// section (L.draw, 11 groups + 3 with dynamic targets); when the episode ends -- time limit, failure or success alike --
// reset_finish only has to load it.  Stage (flags bits 4..6): 0 nothing yet | 1 values 0..7 | 2 values 0..15 | 3 values 0..20 |
// 4 state, angles | 5 targets | 6 complete.  Every piece carries the configuration generation (DynCfg::generation, bumped by
// fwg_update_config / fwg_seed): a draw sampled from other ranges or another seed is discarded.
// Final layout (groups): 0-3 y[0..15] | 4 wind, - | 5 roll pitch yaw Va | 6 alpha beta tgt0 tgt1 | 7 tgt2 - - - |
// 8-9 row noise | 10 generation, episode, flags, - | 11-13 target properties.  Stages 1-2 keep the raw values in groups 0-5.
#define FWG_DRAW_STAGE_SHIFT 4
#define FWG_DRAW_STAGE_MASK (7u << FWG_DRAW_STAGE_SHIFT)
#define FWG_DRAW_READY 6u
__device__ __forceinline__ unsigned draw_stage_of(unsigned flags) { return (flags & FWG_DRAW_STAGE_MASK) >> FWG_DRAW_STAGE_SHIFT; }
// part 0: state, wind, derived angles | part 1: targets (+ alpha, beta, which share their group) | part 2: row noise
__device__ __forceinline__ void draw_store_final(const DevCfg& c, float* __restrict__ S, long N, long e, const ResetDraw& D, int part) {
    const int g0 = c.L.draw >> 2;
    if (part == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) GROUP(S, N, g0 + g, e) = make_float4(D.y[4 * g], D.y[4 * g + 1], D.y[4 * g + 2], D.y[4 * g + 3]);
        GROUP(S, N, g0 + 4, e) = make_float4(D.wind[0], D.wind[1], D.wind[2], 0.f);
        GROUP(S, N, g0 + 5, e) = make_float4(D.d.roll, D.d.pitch, D.d.yaw, D.d.Va);
        GROUP(S, N, g0 + 6, e) = make_float4(D.d.alpha, D.d.beta, 0.f, 0.f);
    } else if (part == 1) {
        GROUP(S, N, g0 + 6, e) = make_float4(D.d.alpha, D.d.beta, D.tgt[0], D.tgt[1]);
        GROUP(S, N, g0 + 7, e) = make_float4(D.tgt[2], D.gust_gain, 0.f, 0.f);
        if (c.any_dynamic_target) {
#pragma unroll
            for (int k = 0; k < FWG_MAX_TARGETS; ++k)
                GROUP(S, N, g0 + 11 + k, e) = make_float4(D.tprop[k][0], D.tprop[k][1], D.tprop[k][2], D.tprop[k][3]);
        }
    } else {
        GROUP(S, N, g0 + 8, e) = make_float4(D.row_noise[0], D.row_noise[1], D.row_noise[2], D.row_noise[3]);
        GROUP(S, N, g0 + 9, e) = make_float4(D.row_noise[4], D.row_noise[5], D.row_noise[6], D.row_noise[7]);
    }
}
__device__ __forceinline__ void draw_load_final(const DevCfg& c, const float* __restrict__ S, long N, long e, ResetDraw& D) {
    const int g0 = c.L.draw >> 2;
    float4 q[10];
#pragma unroll
    for (int g = 0; g < 10; ++g) q[g] = CGROUP(S, N, g0 + g, e);
#pragma unroll
    for (int g = 0; g < 4; ++g) { D.y[4 * g] = q[g].x; D.y[4 * g + 1] = q[g].y; D.y[4 * g + 2] = q[g].z; D.y[4 * g + 3] = q[g].w; }
    D.y[16] = 0.f; D.y[17] = 0.f;
    D.wind[0] = q[4].x; D.wind[1] = q[4].y; D.wind[2] = q[4].z;
    D.d.roll = q[5].x; D.d.pitch = q[5].y; D.d.yaw = q[5].z; D.d.Va = q[5].w;
    D.d.alpha = q[6].x; D.d.beta = q[6].y; D.tgt[0] = q[6].z; D.tgt[1] = q[6].w; D.tgt[2] = q[7].x; D.gust_gain = q[7].y;
    D.row_noise[0] = q[8].x; D.row_noise[1] = q[8].y; D.row_noise[2] = q[8].z; D.row_noise[3] = q[8].w;
    D.row_noise[4] = q[9].x; D.row_noise[5] = q[9].y; D.row_noise[6] = q[9].z; D.row_noise[7] = q[9].w;
#pragma unroll
    for (int k = 0; k < FWG_MAX_TARGETS; ++k) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c.any_dynamic_target) t = CGROUP(S, N, g0 + 11 + k, e);
        D.tprop[k][0] = t.x; D.tprop[k][1] = t.y; D.tprop[k][2] = t.z; D.tprop[k][3] = t.w;
    }
}
__device__ __forceinline__ void touch4(const float4& q) { FWG_TOUCH(q.x); FWG_TOUCH(q.y); FWG_TOUCH(q.z); FWG_TOUCH(q.w); }
__device__ __forceinline__ void touch_draw(const DevCfg& c, const ResetDraw& D) {
#pragma unroll
    for (int i = 0; i < 16; ++i) FWG_TOUCH(D.y[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) FWG_TOUCH(D.wind[i]);
    FWG_TOUCH(D.d.roll); FWG_TOUCH(D.d.pitch); FWG_TOUCH(D.d.yaw); FWG_TOUCH(D.d.Va); FWG_TOUCH(D.d.alpha); FWG_TOUCH(D.d.beta);
    FWG_TOUCH(D.gust_gain);
#pragma unroll
    for (int k = 0; k < FWG_MAX_TARGETS; ++k) {
        FWG_TOUCH(D.tgt[k]);
        if (c.any_dynamic_target) { FWG_TOUCH(D.tprop[k][0]); FWG_TOUCH(D.tprop[k][1]); FWG_TOUCH(D.tprop[k][2]); FWG_TOUCH(D.tprop[k][3]); }
    }
#pragma unroll
    for (int r = 0; r < FWG_MAX_ROWS; ++r) FWG_TOUCH(D.row_noise[r]);
}
// generation | episode the draw is for | flags after sample_targets
__device__ __forceinline__ float4 draw_tag(const float* __restrict__ S, long N, long e, const DevCfg& c) { return CGROUP(S, N, (c.L.draw >> 2) + 10, e); }
// one piece of the next episode's draw (called for lanes whose stage is below FWG_DRAW_READY); returns the new stage.
// Six pieces of at most two or three Philox blocks each, so that a piece fits into the gym wave's wait for its partner:
// 0-2 the sampled initial values (two blocks each) | 3 state vector and derived angles | 4 targets | 5 per-row noise
template <class TAB>
__device__ __forceinline__ unsigned draw_stage_step(const DevCfg& c, DynCfgK& dc, const KArgs& A, long e, unsigned episode_now,
                                                    unsigned stage, TAB& T) {
    const int g0 = c.L.draw >> 2;
    const unsigned episode_new = episode_now + 1u;
    if (stage != 0u) {   // pieces of another configuration generation / seed / episode are discarded
        const float4 tag = draw_tag(A.S, A.N, e, c);
        if (f2u(tag.x) != dc.generation || f2u(tag.y) != episode_new) stage = 0u;
    }
    float v0[FWG_N_RESET_VARS + 3];
    if (stage == 0u) {
        draw_state_values<0, 2>(dc, A, e, episode_new, v0);
#pragma unroll
        for (int g = 0; g < 2; ++g) GROUP(A.S, A.N, g0 + g, e) = make_float4(v0[4 * g], v0[4 * g + 1], v0[4 * g + 2], v0[4 * g + 3]);
        GROUP(A.S, A.N, g0 + 10, e) = make_float4(u2f(dc.generation), u2f(episode_new), 0.f, 0.f);
        return 1u;
    }
    if (stage == 1u) {
        draw_state_values<2, 2>(dc, A, e, episode_new, v0);
#pragma unroll
        for (int g = 2; g < 4; ++g) GROUP(A.S, A.N, g0 + g, e) = make_float4(v0[4 * g], v0[4 * g + 1], v0[4 * g + 2], v0[4 * g + 3]);
        return 2u;
    }
    if (stage == 2u) {
        draw_state_values<4, 2>(dc, A, e, episode_new, v0);
        GROUP(A.S, A.N, g0 + 4, e) = make_float4(v0[16], v0[17], v0[18], v0[19]);
        GROUP(A.S, A.N, g0 + 5, e) = make_float4(v0[20], 0.f, 0.f, 0.f);
        return 3u;
    }
    ResetDraw D;
    if (stage == 3u) {
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const float4 q = CGROUP(A.S, A.N, g0 + g, e);
            v0[4 * g] = q.x;
            if (4 * g + 1 < FWG_N_RESET_VARS + 3) { v0[4 * g + 1] = q.y; v0[4 * g + 2] = q.z; v0[4 * g + 3] = q.w; }
        }
        draw_state(c, v0, D);
        draw_store_final(c, A.S, A.N, e, D, 0);
        return 4u;
    }
    if (stage == 4u) {
        float4 q[7];
#pragma unroll
        for (int g = 0; g < 7; ++g) q[g] = CGROUP(A.S, A.N, g0 + g, e);
#pragma unroll
        for (int g = 0; g < 4; ++g) { D.y[4 * g] = q[g].x; D.y[4 * g + 1] = q[g].y; D.y[4 * g + 2] = q[g].z; D.y[4 * g + 3] = q[g].w; }
        D.y[16] = 0.f; D.y[17] = 0.f;
        D.wind[0] = q[4].x; D.wind[1] = q[4].y; D.wind[2] = q[4].z;
        D.d.roll = q[5].x; D.d.pitch = q[5].y; D.d.yaw = q[5].z; D.d.Va = q[5].w; D.d.alpha = q[6].x; D.d.beta = q[6].y;
        draw_targets(c, dc, A, e, episode_new, T, D);
        draw_store_final(c, A.S, A.N, e, D, 1);
        GROUP(A.S, A.N, g0 + 10, e) = make_float4(u2f(dc.generation), u2f(episode_new), u2f(D.flags), 0.f);
        return 5u;
    }
    draw_row_noise(c, A, e, episode_new, D);
    draw_store_final(c, A.S, A.N, e, D, 2);
    return FWG_DRAW_READY;
}

// The new episode's observation window straight into the row log (row-log mode, k_step2): for an episode end that is known
// before the integration (time limit) the gym wave does this while its partner integrates -- the window does not depend on
// how the old episode ends.  Also pushes the episode's clean record 0 (build_row0).  Same arithmetic as reset_finish.
// (dense: the same rows into the env's record of the dense [N][obs_dim] batch, which the gym wave's staged write then leaves out)
__device__ __forceinline__ void reset_rows_to_log(const DevCfg& c, const KArgs& A, long e, const ResetDraw& D, const float* ring,
                                                  int g_lag, long long win, bool dense = false) {
    Env R;
    R.steps = 0u;
#pragma unroll
    for (int i = 0; i < NY; ++i) R.y[i] = D.y[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) R.wind[i] = D.wind[i];
    R.d = D.d;
    RegTable<FWG_TAB_ROWS> T2;
    fill_vars(R, T2);
#pragma unroll
    for (int k = 0; k < FWG_MAX_TARGETS; ++k) {
        if (k < c.n_targets) {
            T2.put(FWG_TAB_TGT + k, D.tgt[k]);
            T2.put(FWG_TAB_ERR + k, target_error(c.target[k], D.tgt[k], T2.get(c.target[k].var)));
        }
    }
    RegTable<FWG_MAX_OBS> row;
    build_row0(c, A, e, R, T2, row, ring, g_lag, true, 0);
#pragma unroll
    for (int r = 0; r < FWG_MAX_ROWS; ++r) {
        if (r >= c.obs_length) continue;
        const float noise = D.row_noise[r];
        float* dst = dense ? A.obs + e * c.obs_dim + r * c.n_obs : log_row(c, A.obs, A.N, e, win + r);
        if ((c.n_obs & 3) == 0) {
#pragma unroll
            for (int q = 0; q < FWG_MAX_OBS / 4; ++q) {
                if (4 * q < c.n_obs) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = row.get(4 * q + i) + noise * (c.obs[4 * q + i].norm ? V(c).obs[4 * q + i].inv_var : 1.f);
                    reinterpret_cast<float4*>(dst)[q] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < FWG_MAX_OBS; ++j)
                if (j < c.n_obs) dst[j] = row.get(j) + noise * (c.obs[j].norm ? V(c).obs[j].inv_var : 1.f);
        }
    }
}

// The new episode's state and gym-side bookkeeping from its draw; leaves the target errors in `err` and the variable table in T
template <bool TURB, class TAB>
__device__ __forceinline__ void reset_state(const DevCfg& c, const KArgs& A, long e, Env& E, TAB& T, int g_end, int g_bit,
                                            const ResetDraw& D, bool have_gw, float (&err)[3]) {
    if (c.model_n > 0) {   // simulator.model: the set prepared for this episode by k_model_draw becomes the current one
#pragma unroll
        for (int g = 0; g < FWG_AERO_GROUPS; ++g)
            GROUP(A.S, A.N, (c.L.aero >> 2) + g, e) = CGROUP(A.S, A.N, (c.L.aero_next >> 2) + g, e);
#pragma unroll
        for (int g = 0; g < FWG_AERO_GROUPS; ++g)   // (FWG_N_PARAMS <= 4 FWG_AERO_GROUPS)
            if (4 * g < c.model_n) GROUP(A.S, A.N, (c.L.model_raw >> 2) + g, e) = CGROUP(A.S, A.N, (c.L.model_raw_next >> 2) + g, e);
    }
    if (c.randomize_scaling) {   // reward.randomize_scaling: likewise (the values are read by the following steps only)
#pragma unroll
        for (int g = 0; g < FWG_MAX_FACTORS / 4; ++g)
            GROUP(A.S, A.N, (c.L.fscale >> 2) + g, e) = CGROUP(A.S, A.N, (c.L.fscale_next >> 2) + g, e);
    }
    // the set of the episode AFTER this one is now due: the env goes on the queue the next launch's k_model_draw_q works off
    if ((c.model_n > 0 || c.randomize_scaling) && A.mq != nullptr) A.mq[1u + atomicAdd(A.mq, 1u)] = (unsigned)e;
    E.episode = D.episode;
    E.steps = 0u;
    E.sft = 0u;
    // (the sticky goal bit may date from this very step; a finished-episode record stays pending across the reset)
    E.flags = (E.flags & (FWG_FLAG_GOAL_ACHIEVED | FWG_FLAG_FIN_PENDING)) | (D.flags & ~(FWG_FLAG_GOAL_ACHIEVED | FWG_FLAG_FIN_PENDING));
#pragma unroll
    for (int i = 0; i < NY; ++i) E.y[i] = D.y[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) E.wind[i] = D.wind[i];
#pragma unroll
    for (int i = 0; i < FWG_N_DRYDEN; ++i) E.dry[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) E.gust[i] = 0.f;
    E.gust_gain = D.gust_gain;
    E.d = D.d;
#pragma unroll
    for (int k = 0; k < FWG_MAX_TARGETS; ++k) {
        E.tgt[k] = D.tgt[k];
#pragma unroll
        for (int i = 0; i < 4; ++i) E.tprop[k][i] = D.tprop[k][i];
    }
    fill_vars(E, T);
#pragma unroll
    for (int k = 0; k < 3; ++k) err[k] = 0.f;
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
        E.psh[k] = 0.f;
    }
    E.settle[0] = 0xFFFFFFFFu; E.settle[1] = 0xFFFFFFFFu;
    E.sdcmd = 0.f;
    if (c.metrics) {   // record 0 of the cumulative error sums (fixed point, see Fix3)
        Fix3 s0;
#pragma unroll
        for (int k = 0; k < 3; ++k) s0.s[k] = fix_quant(err[k]);
        GROUP(A.S, A.N, (c.L.end_ring >> 2) + g_end, e) = fix3_pack(s0);
    }
    E.wcnt = 0u; E.gcnt[0] = 0u; E.gcnt[1] = 0u;
    if (c.goal_enabled) {   // the ring keeps its old contents (see goal_push); only the word holding g_bit is touched
        // (have_gw: the step kernel holds that very word in E.gw already -- it is the one this step's record went into)
        if (!have_gw) E.gw = reinterpret_cast<const unsigned*>(A.S)[((unsigned)c.L.goal + (unsigned)(g_bit >> 3)) * (unsigned)A.N + (unsigned)e];
        goal_push(c, E, goal_flags(c, err), g_bit, 0u);   // ... and written back by store_gym
    }
}

template <bool TURB, class TAB, class OB>
__device__ __forceinline__ void reset_finish(const DevCfg& c, const KArgs& A, long e, Env& E, TAB& T, OB& ob, const float* ring,
                                             int g_end, int g_lag, int g_bit, const ResetDraw& D, bool have_gw = false,
                                             bool rows_done = false) {
    float err[3];
    reset_state<TURB>(c, A, e, E, T, g_end, g_bit, D, have_gw, err);
    if (rows_done) return;   // (k_step2, foreseen end: the partner wave wrote the cold and simulator rows and the observation window)
    store_cold(c, A.S, A.N, e, E);
    // ---- observation: every row is the initial record (+ per-row init noise when length > 1)
    float int_pad[3] = {0.f, 0.f, 0.f};
    if (c.int_window) {
        // integrator entries (fixed_wing.py:804-810).  The reference builds the reset observation BEFORE it re-creates its
        // histories (:317-321): the entries show the PREVIOUS episode's windowed sum + (W + 1) times its initial error
        // (E.int_reset, set by the caller from the old episode), W e0 on the env's very first reset ("history is None");
        // rows that later pad the window use (W + 1) e0 of the NEW episode
#pragma unroll
        for (int k = 0; k < FWG_MAX_TARGETS; ++k) {
            if (k < c.n_targets) {
                T.put(FWG_TAB_INT + k, D.episode == 1u ? (float)c.int_window * err[k] : E.int_reset[k]);
                int_pad[k] = (float)(c.int_window + 1) * err[k];
            }
        }
    }
    build_row0(c, A, e, E, T, ob, ring, g_lag, true, 0, nullptr, c.int_window ? int_pad : nullptr);
    if (c.obs_length > 1) {
#pragma unroll
        for (int r = FWG_MAX_ROWS - 1; r >= 0; --r) {
            if (r >= c.obs_length) continue;
            const float noise = D.row_noise[r];
#pragma unroll
            for (int j = 0; j < FWG_MAX_OBS; ++j)
                if (j < c.n_obs) ob.put(r * c.n_obs + j, ob.get(j) + noise * (c.obs[j].norm ? V(c).obs[j].inv_var : 1.f));
        }
    }
    if (c.obs_noise) add_obs_noise(c, A, e, E, ob);
}

template <bool TURB, class TAB, class OB>
__device__ __forceinline__ void reset_env(const DevCfg& c, DynCfgK& dc, const KArgs& A, long e, Env& E, TAB& T, OB& ob,
                                          const float* ring, int g_end, int g_lag, int g_bit) {
    ResetDraw D;
    reset_sample(c, dc, A, e, E.episode, E.flags, T, D);
    reset_finish<TURB>(c, A, e, E, T, ob, ring, g_end, g_lag, g_bit, D);
}
