// fwgym.hip -- libfwgym.so: fused HIP kernels for gfx950 (MI355X) + the C ABI declared in include/fwgym.h.
//
// Replaces, for a batch of N independent aircraft, the reference hot path FixedWingAircraft.step()/reset()
// (gym_fixed_wing/fixed_wing.py:287-437) INCLUDING the simulator it calls (pyfly PyFly.step, fixed_wing.py:358).
//
// Mapping to the hardware: one wavefront lane = one aircraft, one 64-lane workgroup per wave so that 65 536 envs give
// 1 024 workgroups = one wave on every SIMD of the 256 CUs.  Persistent state is SoA [field][env] in HBM (every
// access is a 256-B coalesced row segment per wave); the un-predictable, config-driven parts (which variable feeds
// which observation entry) are resolved through LDS tables [entry][lane] addressed with wave-uniform indices; lagged
// observation rows and the action window stream from HBM straight into LDS (global_load_lds, no VGPR round trip)
// while the RK4 integration runs; the [env][obs_dim] output is transposed through a padded LDS tile so that the
// HBM writes are fully coalesced.  Everything -- action scaling, actuator + 6-DOF RK4, Dryden filter, goal/streak,
// reward, target propagation, observation, metrics, auto-reset -- is ONE launch per env step.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "fwgym_env.h"
#include "fwgym_actor.h"   // the rollout head (k_rollout below runs it in the same launch as the env step)

__device__ __forceinline__ float wave_sum(float v) { return wave_sum64(v); }

// ---------------------------------------------------------------------------------------------------------------------
// Specialisation.  The static configuration of an environment (DevCfg: observation layout, reward factors, aircraft
// constants, integrator setup ...) never changes after construction, so the build freezes the configurations it knows
// about (csrc/generated/specs.inc, written by fwg_dump_spec through __graft_entry__.build()) into constexpr objects;
// a kernel instantiated with SPEC >= 0 reads its configuration from that object and the compiler folds every
// config-driven loop, branch and constant (no scalar loads, no descriptor interpretation).  SPEC = -1 is the generic
// kernel reading the same structure from memory; fwg_create picks the specialised instance when the lowered
// configuration is bit-identical to a frozen one.  Both run the same source.
// ---------------------------------------------------------------------------------------------------------------------
struct SpecWords { unsigned w[sizeof(DevCfg) / 4]; };
static_assert(sizeof(SpecWords) == sizeof(DevCfg), "DevCfg must be made of 32-bit members only");
#if defined(FWG_SPECS_FILE)      /* run-time specialisation (gym_fixed_wing/jit.py): one frozen configuration */
#include FWG_SPECS_FILE
#elif defined(FWG_WITH_SPECS)    /* build-time presets */
#include "generated/specs.inc"
#else
#define FWG_SPEC_LIST(X)
#endif
#ifndef FWG_SHAPE_LIST   /* the frozen configurations that are ALSO built as shape instances (structure frozen, values from memory) */
#define FWG_SHAPE_LIST(X)
#endif
#define FWG_SHAPE_BASE FWG_INSTANCE_SHAPE   /* kernel instance FWG_SHAPE_BASE + i = the shape instance of frozen configuration i */
#define FWG_DEFINE_SPEC(i) static constexpr DevCfg kSpec##i = __builtin_bit_cast(DevCfg, kSpecWords##i);
FWG_SPEC_LIST(FWG_DEFINE_SPEC)
template <int SPEC> struct SpecCfg {
    static constexpr int obs_dim = 16;   // (generic kernels never instantiate the fused rollout launch)
    static constexpr bool rollout_ok = false;
    static constexpr bool one_rk4_step = false;   // (the wave team of k_step2 exists for specialised configurations only)
    static __device__ __forceinline__ const DevCfg& get(const DevCfg* cp) { return *cp; }
};
#define FWG_SPEC_GETTER(i)                                                                                \
    template <> struct SpecCfg<i> {                                                                       \
        static constexpr int obs_dim = kSpec##i.obs_dim;                                                  \
        static constexpr bool one_rk4_step = kSpec##i.nsub == 1;                                          \
        /* the fused head + step launch (k_rollout): dense observation batch within the head's 64 entries */ \
        static constexpr bool rollout_ok = kSpec##i.obs_log == 0 && kSpec##i.obs_dim <= 64;               \
        static __device__ __forceinline__ const DevCfg& get(const DevCfg*) { return kSpec##i; }           \
    };
FWG_SPEC_LIST(FWG_SPEC_GETTER)
// Shape instances: what a configuration outside the frozen ones runs on when only its VALUES differ from one of them (a
// reward scaling, a constraint, the time limit, an aircraft constant, a noise level ...) and no run-time compiler is at hand:
// the kernel computes with the frozen object's STRUCTURE (kShape: every loop bound, branch and index folds as in the frozen
// kernel) and reads every VALUE member through V(c) from the configuration in memory (fwgym_dev.h cfg_values): scalar loads at
// the point of use.  (Measured dead ends: a kernel-local merged copy of the object -- a 3 KB alloca the compiler does not split,
// 45 000 spilled registers; the memory configuration + __builtin_assume on its structure words -- not propagated, as generic.)
#define FWG_SHAPE_OBJECT(i) static constexpr DevCfg kShape##i = as_shape(kSpec##i);
FWG_SHAPE_LIST(FWG_SHAPE_OBJECT)
#define FWG_SHAPE_GETTER(i)                                                                               \
    template <> struct SpecCfg<FWG_SHAPE_BASE + i> {                                                      \
        static constexpr int obs_dim = kSpec##i.obs_dim;                                                  \
        static constexpr bool one_rk4_step = kSpec##i.nsub == 1;                                          \
        static constexpr bool rollout_ok = false;   /* (the fused rollout launch: frozen configurations only) */ \
        static __device__ __forceinline__ const DevCfg& get(const DevCfg*) { return kShape##i; }          \
    };
FWG_SHAPE_LIST(FWG_SHAPE_GETTER)

// ---------------------------------------------------------------------------------------------------------------------
// step kernel
// ---------------------------------------------------------------------------------------------------------------------
template <int SPEC> struct KernelTypes {  // generic kernel: lane-private LDS columns addressed by run-time indices
    typedef LdsTable Tab;
    typedef LdsTable Obs;
    static constexpr bool generic = true;
    static __device__ __forceinline__ Tab tab(float* lds, const LdsMap& M, int lane) { return Tab{lds + M.tab + lane}; }
    static __device__ __forceinline__ Obs obs(float* lds, const LdsMap& M, int lane) { return Obs{lds + M.obs + lane}; }
};
#define FWG_SPEC_TYPES(i)                                                                                     \
    template <> struct KernelTypes<i> { /* specialised kernel: everything in registers */                     \
        typedef RegTable<FWG_TAB_ROWS> Tab;                                                                   \
        typedef RegTable<kSpec##i.obs_dim> Obs;                                                               \
        static constexpr bool generic = false;                                                                \
        static __device__ __forceinline__ Tab tab(float*, const LdsMap&, int) { return Tab(); }               \
        static __device__ __forceinline__ Obs obs(float*, const LdsMap&, int) { return Obs(); }               \
    };
FWG_SPEC_LIST(FWG_SPEC_TYPES)
#define FWG_SHAPE_TYPES(i) template <> struct KernelTypes<FWG_SHAPE_BASE + i> : KernelTypes<i> {};
FWG_SHAPE_LIST(FWG_SHAPE_TYPES)

// Attached rollout head (fwg_attach_observer): this wave's batch moments of the observation records and discounted
// returns it produced, added to the head's accumulators (fwgym_env.h "batch-moment accumulators").  Called by all lanes.
// HS (k_rollout: head and env step in one launch): the running means and the act counter are the ones the head phase of
// THIS launch computed -- `hs_mean` / `hs_ret_mean` in LDS, `hs_ctr` -- not the published copies in memory (which the first
// block of this same launch is writing).
struct NoHeadStats { static constexpr bool enabled = false; const float* mean = nullptr; float ret_mean = 0.f; unsigned ctr = 0u; };
struct HeadStats { static constexpr bool enabled = true; const float* mean; float ret_mean; unsigned ctr; };
template <class OB, class HS = NoHeadStats>
__device__ __forceinline__ void step_moments(const DevCfg& c, const KArgs& A, const OB& ob, float reward, bool done, bool valid,
                                             int lane, long e, int sub, float ret_prev, const HS& hs = HS()) {
#ifndef FWG_ABL_NO_ACC
    const int D = c.obs_dim;
    // VecNormalize.step_wait: ret = ret * gamma + r; ret_rms.update(ret); ret[done] = 0
    // (ret_prev = A.acc_ret[e], requested with the bookkeeping rows: a load issued HERE, after the step's stores, would sit
    // through the acknowledgement of every one of them -- 3-4k ticks at the end of every wave, tools/timeline_rollout.py)
    float dr = 0.f;
    if (valid) {
        const float r = ret_prev * A.acc_gamma + reward;
        A.acc_ret[e] = done ? 0.f : r;
        dr = r - (HS::enabled ? hs.ret_mean : *(FWG_KCONST(float)*)A.acc_ret_mean);
    }
    // the accumulator set of this launch: the head's act counter mod 3 (fwgym_actor.h)
    const unsigned ctr = HS::enabled ? hs.ctr : *(FWG_KCONST(unsigned)*)A.acc_ctr;
    unsigned long long* acc = A.acc + (size_t)(ctr % FWG_ACC_SETS) * FWG_ACC_SHARDS * A.acc_cols;
    // the running means the deviations are taken from: written by the head's previous launch, constant during this one.
    // Read through a plain pointer (after this kernel's stores) every one of them was a vector load with a wait of its own --
    // 2 D serial round trips, most of what the attached moments cost (§5); as scalar loads they arrive in a few batches.
    // (k_rollout: from LDS.  Loaded UNCONDITIONALLY, all at once: inside the per-column conditionals the compiler gave every
    // one of them a branch, a ds_read and a wait of its own -- 24 serial LDS round trips at the very end of every wave)
    FWG_KCONST(float)* mean_k = (FWG_KCONST(float)*)A.acc_mean;
    const float one = valid ? 1.f : 0.f;
#pragma unroll
    for (int chunk = 0; chunk < (2 * FWG_MAX_OBS * FWG_MAX_ROWS + 4 + 31) / 32; ++chunk) {
        if (32 * chunk < 2 * D + 4) {
            // the 16 observation entries this chunk's columns belong to: column 4 + 2 k / 5 + 2 k = deviation of entry k / its
            // square (layout of fwgym_env.h, FWG_ACC_COLUMN); entry of column 32 chunk + i: k = 16 chunk - 2 + (i >> 1)
            float dev[16], mean[16];
            if (HS::enabled) {   // k_rollout: the updated running means from LDS -- five 16-byte reads, all issued before the first
                // use (rounded(): an empty asm the value passes through; it pins the reads HERE, for all lanes -- left alone the
                // compiler sinks every one of them into a branch on `valid` of its own, each with its own wait)
                float4 q[5];
#pragma unroll
                for (int g = 0; g < 5; ++g) {
                    const int k0 = 16 * chunk - 4 + 4 * g;   // entries k0 .. k0 + 3 (the head keeps 64 means: zeros beyond D)
                    q[g] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k0 >= 0 && k0 < D && k0 + 3 < FWG_ACT_MAX_OBS) q[g] = *reinterpret_cast<const float4*>(hs.mean + k0);
                }
#pragma unroll
                for (int g = 0; g < 5; ++g) { q[g].x = rounded(q[g].x); q[g].y = rounded(q[g].y); q[g].z = rounded(q[g].z); q[g].w = rounded(q[g].w); }
#pragma unroll
                for (int j = 0; j < 16; ++j) {   // entry k = 16 chunk - 2 + j sits at position j + 2 of the 20 values read
                    const float4 g = q[(j + 2) >> 2];
                    mean[j] = ((j + 2) & 3) == 0 ? g.x : ((j + 2) & 3) == 1 ? g.y : ((j + 2) & 3) == 2 ? g.z : g.w;
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int k = 16 * chunk - 2 + j;
                dev[j] = 0.f;
                if (k >= 0 && k < D && k < FWG_MAX_OBS * FWG_MAX_ROWS) {
                    const float m = HS::enabled ? mean[j] : mean_k[k];
                    dev[j] = valid ? ob.get(k) - m : 0.f;
                }
            }
            float v[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int col = 32 * chunk + i;
                v[i] = col == 0 ? dr : col == 1 ? dr * dr : (col == 2 || col == 3) ? one : ((i & 1) == 0 ? dev[i >> 1] : dev[i >> 1] * dev[i >> 1]);
            }
#ifdef FWG_ABL_NO_TOTALS
            acc_flush(acc, A.acc_cols, sub & (FWG_ACC_SHARDS - 1), chunk, lane, v[0] + v[7] + v[13] + v[27]);
#else
            acc_flush(acc, A.acc_cols, sub & (FWG_ACC_SHARDS - 1), chunk, lane, wave_totals32(v, lane));
#endif
        }
    }
#endif
}

// One env step for the 64 environments of a wave.  ROLE 0: the whole step in one wave (k_step).  ROLES 1 / 2: the step
// split over the two waves of a 128-thread workgroup (k_step2), which then run on two SIMDs of the CU at the same time
// -- a single wave issues at most one vector instruction every ~5 cycles however much is independent
// (tools/ub_valu.hip), so the only way to shorten the step's dependent chain is to put its independent parts on
// different waves:
//   role 1 ("physics"): actions + simulator rows -> integration -> hands the new state to its partner through LDS,
//                       then advances the Dryden filter and writes the simulator rows back;
//   role 2 ("gym"):     bookkeeping rows, action windows, the turbulence noise (Philox + Box-Muller, handed to the
//                       physics wave through LDS) while the integration runs; then goal / reward / targets / observation
//                       / metrics / auto-reset on the state it receives.
// Synchronisation: workgroup barrier A = state and noise handed over; then a ONE-WAY mark in LDS that only the physics wave
// raises (1 = its hand-off areas are read, 2 = its rows are in memory) and the gym wave waits for only where it re-uses the
// former (output staging) or overwrites the latter (an env it re-initialises itself) -- see FWG_FLAG_RAISE / FWG_FLAG_WAIT.
// k_step2, physics wave: the actuator states at t + h/2 and t + h, advanced by the gym wave while this wave evaluates the first
// stage (a tagged message, see fwg_msg_take: this wave waits only if its partner is late)
struct PartnerActuators {
    static constexpr bool enabled = true;
    const float* p;
    // (the stages see the deflections only: the rates at t + h/2 stay with the partner)
    __device__ __forceinline__ void fetch_half(float (&a_half)[5]) const {
        const float4 q = fwg_msg_take(p, FWG_TAG_ACTS);
        a_half[0] = q.x; a_half[1] = q.y; a_half[2] = q.z;
    }
    __device__ __forceinline__ void fetch_full(float (&a_full)[5]) const {
        const float4 q2 = fwg_msg_take(p + 8, FWG_TAG_ACTS);
        const float4 q1 = reinterpret_cast<const float4*>(p)[1];
        a_full[0] = q1.x; a_full[1] = q1.y; a_full[2] = q1.z; a_full[3] = q1.w; a_full[4] = q2.x;
    }
};
// measurement builds: a time stamp after every stage of the integration (tools/timeline.py)
struct StageStamps {
    const KArgs& A;
    __device__ __forceinline__ void operator()(int st) const { FWG_TL(A, 26 + st); }
};
// k_step2, physics wave: the new state leaves for the gym wave as soon as it exists (sim_step `hand`)
struct HandToGym {
    static constexpr bool enabled = true;
    float* h;   // this lane's hand-off area (FWG_HAND_WORDS)
    const KArgs* A;
    __device__ __forceinline__ void state(const float (&yy)[NY], const float (&ea)[5]) const {
        float4* h4 = reinterpret_cast<float4*>(h);
        h4[0] = make_float4(yy[4], yy[5], yy[6], yy[7]);
        h4[1] = make_float4(yy[8], yy[9], yy[10], yy[11]);
        h4[2] = make_float4(yy[12], yy[13], yy[14], yy[15]);
        h4[3] = make_float4(ea[0], ea[1], ea[2], ea[3]);
        fwg_msg_put(h + 16, ea[4], 0.f, 0.f, FWG_TAG_STATE);
        FWG_TL(*A, 30);
    }
    __device__ __forceinline__ void result(float Va, float alpha, float beta, int fail) const {
        fwg_msg_put(h + 20, Va, alpha, beta, FWG_TAG_RESULT | (unsigned)fail);
        FWG_TL(*A, 31);
    }
};

// =====================================================================================================================
// step_wave: ONE env step of 64 environments (fixed_wing.py:338-437), in three ROLES compiled from this one body
//   ROLE 0  one wave does everything                    (k_step: the generic kernel and FWGYM_SPLIT=0)
//   ROLE 1  PHYSICS wave | ROLE 2  GYM wave of a pair   (k_step2, k_rollout: `if (SPLIT && PHYS)` / `if (SPLIT && GYM)` blocks)
// and two observation LAYOUTS (c.obs_log > 0: row log + zero-copy window; 0: dense batch with a lag ring in the arena).
//
// MAP (in source order; [P] physics wave / one wave, [G] gym wave / one wave)
//   A  entry      ring positions (resolve_slots), LDS areas, message tags cleared by their WRITER, one workgroup barrier.
//                 [P] requests the simulator rows (+ per-lane aircraft constants), reads the raw action;  [G] requests nothing
//                 until the raw action / actuator state arrive as a message (FWG_TAG_RAW), then its part-1 bookkeeping rows.
//   B  simulator  [P] action scaling -> sim_step (RK4; asks for the deflections at t + h/2 and t + h: FWG_TAG_ACTS) -> sends the
//                 candidate state + Euler-angle arguments (FWG_TAG_STATE), then Va / alpha / beta + failure code (FWG_TAG_RESULT);
//                 a FAILED step re-sends the untouched state (FWG_TAG_OLD) with the COMMITTED step's air data (store_sim).
//                 [G] meanwhile: actuator micro-steps, turbulence noise (Philox), everything of the gym logic that does not
//                 depend on this step's integration (counters, action-class reward factors, "action" observation entries,
//                 control variation, target propagation), one piece of the next episode's prepared draw, the padding rows of
//                 early-episode lanes, and -- for an end that is FORESEEN (time limit) -- the prefetch of what its end branch
//                 reads (pre_end / pre_draw / pre_rows).  Sends FWG_TAG_TAIL (step index, padding-row index, install flag).
//   C  gym logic  [G] errors, goal window / streak, state reward factors, metrics accumulators, done / termination code;
//                 an UNFORESEEN end (failure, success, or a foreseen end whose last step failed) requests its operands here,
//                 before the bookkeeping stores.  Scalar outputs (reward, done, code) leave as soon as they are final.
//   D  observation [G] record 0 of this step (build_row0), lagged rows (row log: planes of the log; dense: lag ring), the
//                 terminal observation of an ending lane, the episode-end branch (finished-episode record, reset_finish).
//   E  stores     [G] bookkeeping rows (store_gym), observation (row log: one row; dense: staged through LDS, write_obs).
//   T  tail       [P] after its last message: Dryden advance + next gust sample, simulator rows back (store_sim), and the side
//                 work that prepares the NEXT step: padding rows of early lanes into the log (tail_rows), and for a foreseen
//                 end with a valid prepared draw the whole next episode (pre_install: window, record 0, simulator + cold rows;
//                 partner_rows: the old window's lagged rows copied into the terminal batch first).  Raises the one-way mark
//                 (level 1: hand-off areas read; level 2: its row stores acknowledged).
//
// INVARIANTS the episode-end paths rest on (each was a bug once; the test that holds it is named)
//   I1  A failed step leaves state AND air data as the last committed step left them (Va / alpha / beta derived with THAT
//       step's gust, kept in the second derived group): tests/test_emu_parity.py::test_failed_steps_under_turbulence_...,
//       tests/test_gpu_oracle_coverage.py [fail_prone]; mutant `airdata` of tools/mutation_check.py.
//   I2  The partner installs the next episode ONLY for a foreseen end whose last step SUCCEEDED (pre_rows / end_p carry
//       `fail == 0`), in both layouts: a failed step's terminal observation reads lagged rows one record further back -- in the
//       dense layout out of the very lag-ring slot an install pushes the new record 0 into.  tests/test_emu_coverage.py
//       [lockstep-dense], tests/test_gpu_oracle_coverage.py [fail_prone_lockstep-dense]; mutant `install_on_failed_last_step`.
//   I3  What the gym wave reads of an ending lane's OLD window it has in registers (FWG_TOUCH) before FWG_TAG_TAIL leaves:
//       after that message the partner may overwrite those planes / slots at any time.
//   I4  The gym wave re-initialises an env itself (unforeseen end) only after mark level 2: the partner's stores of the old
//       episode's simulator rows are acknowledged.
//   I5  No draw piece in a wave that hosts an end; a prepared draw is used only with a matching (generation, episode) tag,
//       otherwise the end draws on the spot (tests/test_emu_parity.py::test_prepared_draw_is_discarded_...).
//   I6  Ring slots are positions of the GLOBAL step counter (StepSlots); an env's validity inside a ring is its own age.
// The fuzzer (tests/test_emu_fuzz.py) runs the three roles against each other and the oracle over drawn configurations.
// =====================================================================================================================
// `sub` = index of the 64-environment group this wave (pair of waves) steps: the workgroup index in k_step / k_step2, four
// groups per workgroup in k_rollout, whose head phase leaves the actions in LDS (`act_lds`: this group's [64][4] floats) and
// its updated running statistics in `hs`.
template <bool TURB, int SPEC, int ROLE, class HS = NoHeadStats>
__device__ __forceinline__ void step_wave(const DevCfg* __restrict__ cp, const DynCfg* __restrict__ dp, const KArgs& A0, float* lds, int sub,
                                          const float* act_lds = nullptr, const HS& hs = HS()) {
    typedef KernelTypes<SPEC> KT;
    constexpr bool PHYS = ROLE != 2, GYM = ROLE != 1, SPLIT = ROLE != 0;
    const DevCfg& c = SpecCfg<SPEC>::get(cp);
    KArgs A_ = resolve_slots(c, A0);
    // An attached rollout head's running means / return mean / act counter are read through the CONSTANT address space (scalar
    // loads, §5) although the launch before this one wrote them.  That is sound only if the scalar data cache is clean at kernel
    // entry; the runtime's acquire at a dispatch does that -- as far as it is documented.  One s_dcache_inv per wave makes it
    // explicit (the pointers pass through the asm statement, so the compiler cannot hoist their loads above it).
    if constexpr (!HS::enabled && SPLIT && GYM) {   // (the three live in one ActorStats record: one pointer through the asm)
        if (A_.acc != nullptr) {
            const char* st = reinterpret_cast<const char*>(fwg_fresh_scalar_view(A_.acc_mean)) - offsetof(ActorStats, mean);
            A_.acc_mean = reinterpret_cast<const float*>(st + offsetof(ActorStats, mean));
            A_.acc_ret_mean = reinterpret_cast<const float*>(st + offsetof(ActorStats, ret_mean));
            A_.acc_ctr = reinterpret_cast<const unsigned*>(st + offsetof(ActorStats, act_counter));
        }
    }
    const KArgs& A = A_;
    if (PHYS && A0.slots_out != nullptr && sub == 0 && (threadIdx.x & (FWG_WAVE - 1)) == 0)
        *A0.slots_out = next_slots(c.obs_step, c.obs_log, c.obs_length, c.L.window, c.L.lag_depth, c.streak_req, load_slots(A0.slots_in));
#ifdef FWG_ABL_EMPTY  // FWG_ABL_*: measurement-only switches (tools/ablate.py), never defined in the product build
    return;
#endif
    DynCfgK& dc = *(DynCfgK*)dp;
    FWG_TL(A, 0);
    const int lane = threadIdx.x & (FWG_WAVE - 1);
    const long env0 = (long)sub * FWG_WAVE;
    const bool valid = env0 + lane < A.N;
    const long e = valid ? env0 + lane : A.N - 1;
    const LdsMap M = lds_map(c.obs_dim, c.n_obs, c.L.window, c.use_cmd_ring, KT::generic, c.obs_log, SPLIT);
    const fwg_layout& L = c.L;
    const int W = L.window;
    typename KT::Tab T = KT::tab(lds, M, lane);
    typename KT::Obs ob = KT::obs(lds, M, lane);
    float* aring = lds + M.aring + lane * 4;   // this lane's entries of the raw-action window [slot][lane][4]
    float* cring = lds + M.cring + lane * 4;   // ... of the constrained-command window (only when observations need it)
    // split kernel hand-off areas, aliasing the output staging area (see lds_map)
    float* hand = lds + M.stage + lane * FWG_HAND_WORDS;                       // physics -> gym (tagged messages, fwgym_dev.h)
    float* tailm = lds + M.stage + FWG_WAVE * FWG_HAND_WORDS + lane * FWG_TAIL_WORDS;   // gym -> physics
    float* acts = lds + M.stage + FWG_WAVE * (FWG_HAND_WORDS + FWG_TAIL_WORDS) + lane * FWG_ACT_WORDS;   // gym -> physics
    float* mark = lds + M.flag;   // the one-way hand-shake mark: cleared by the physics wave before the entry barrier, raised by it later
    // The gym wave advances the actuators for the physics wave (they depend on the commands and the actuator states only), in
    // the time it would otherwise wait for its bookkeeping rows; the physics wave picks them up after its first stage.  (Round 2
    // measured this SLOWER by 1.0-1.2 us with a __syncthreads() inside the stage loop as the hand-over: both waves waited, and
    // for every store in flight; as a tagged message the physics wave waits for nothing unless its partner is late.)
    constexpr bool ext_act = SPLIT && SpecCfg<SPEC>::one_rk4_step;

    // ---- phase A: issue every load up front.  The action windows stream HBM -> LDS (global_load_lds, no VGPRs, they
    // are addressed by the run-time ring slot); everything else goes to registers.
    // k_step2 with the actuators on the gym wave: that wave requests NOTHING at kernel entry -- the physics wave's rows, which
    // the step's length hangs on, have the CU's vector-memory path (64 B per clock, requests served in order: the 4 workgroups'
    // 36 KiB of simulator rows take ~600 clocks by themselves) to themselves, and the raw action and the actuator states the gym
    // wave needs first come over from the physics wave through LDS as soon as they have landed there
    constexpr bool gym_waits = SPLIT && GYM && ext_act;
    float raw[3] = {0.f, 0.f, 0.f};
    if (!gym_waits) {
#pragma unroll
        for (int i = 0; i < 3; ++i) raw[i] = HS::enabled ? act_lds[lane * 4 + i] : A.actions[e * 3 + i];   // [N][3]: a wave reads 768 contiguous bytes
    }
    Env E;
    float4 act_q3 = make_float4(0.f, 0.f, 0.f, 0.f), act_q4 = act_q3;
    if (PHYS) load_sim<TURB>(c, A.S, A.N, e, E);
    if (!gym_waits) load_cold(c, A.S, A.N, e, E);
    // per-lane force / moment constants (simulator.model; the generic kernel keeps ONE code path): requested with the
    // simulator rows, so that the first right-hand side does not start by waiting for 13 more round trips
    Aero la;
    if (PHYS && (KT::generic || c.model_n > 0)) {
        if (c.model_n > 0) load_aero(c, A.S, A.N, e, la); else aero_from_cfg(c, la);
    }
    // k_step2, row-log mode, no attached observer: the PHYSICS wave owns the work that prepares the state the NEXT step starts
    // from, in its idle tail (after the integration, while the gym wave runs its post-barrier chain):
    //  * tail_rows   -- the padding rows of lanes in the first steps of an episode (record 0 + fresh per-row noise + the new
    //                   actuator values), written straight into the row log;
    //  * pre_install -- for an episode end known before the integration (time limit) with a valid prepared draw: the new
    //                   episode's observation window, its record 0, its simulator and cold rows.
    // Which lanes are concerned it learns from the gym wave through LDS at the barrier (a load of the counters of its own would
    // have to be waited for at the kernel entry or be kept from the scheduler's hoisting: both measured slower).
    // (steps that fail or end an episode otherwise are completed by the gym wave as before)
    // (HS::enabled -- the fused rollout launch -- has an observer attached by construction)
    const bool tail_rows = SPLIT && !HS::enabled && c.obs_log > 0 && c.obs_length > 1 && A.acc == nullptr;
    // (dense batch: the same, the new window going into the env's record of the batch -- unless every step re-draws observation noise)
    // (a foreseen end whose LAST STEP FAILS is not installed by the physics wave, in either layout: the failed step's terminal
    // observation takes lagged rows from one record further back -- in the dense layout out of the very lag-ring slot the install
    // pushes the new episode's record 0 into; rounds 4-5 installed regardless there: 1 of ~300 time-limit ends)
    const bool pre_install = SPLIT && !HS::enabled && c.auto_reset && (c.obs_log > 0 || !c.obs_noise) &&
                             V(c).steps_max > (c.obs_length - 1) * c.obs_step + 1 &&
                             A.acc == nullptr && !c.has_int_obs;   // (integrator entries of a reset observation depend on how the old episode ends)
    // row-log mode: the lagged rows of such an end's TERMINAL observation never pass through the gym wave -- the partner copies
    // them, log -> terminal batch, before it writes the new window over their planes (a few lanes: one 4-byte load and store per
    // ending lane, the wave's lanes taking one word each, where the gym wave spent twelve 16-byte loads before the hand-over
    // and as many stores on its chain after it).  The gym wave reads them back only in the rare cases that need them in registers
    // (the last step failed, the prepared draw was stale): the partner then installs nothing and the end takes the unforeseen path
    const bool partner_rows = pre_install && c.obs_log > 0 && c.obs_length > 1;

    // k_step2 entry: every wave clears the tags of the messages it is going to WRITE (what a previous workgroup left in this
    // LDS must not be mistaken for one), then ONE workgroup barrier -- at the point where both waves wait for their first rows
    // anyway; from here on the two waves meet through tagged messages only
    if (SPLIT) {
        if (PHYS) {
            fwg_msg_put(hand + 16, 0.f, 0.f, 0.f, 0u);
            fwg_msg_put(hand + 20, 0.f, 0.f, 0.f, 0u);
            fwg_msg_put(hand + 24, 0.f, 0.f, 0.f, 0u);
            fwg_msg_put(hand + 8, 0.f, 0.f, 0.f, 0u);
            if (lane == 0) FWG_FLAG_RAISE(mark, 0);
        } else {
            fwg_msg_put(acts, 0.f, 0.f, 0.f, 0u);
            fwg_msg_put(acts + 8, 0.f, 0.f, 0.f, 0u);
            fwg_msg_put(tailm, 0.f, 0.f, 0.f, 0u);
        }
#ifndef FWG_ABL_NO_ENTRY_BARRIER   /* (measurement only: what the entry barrier costs; without it stale tags are possible) */
        FWG_BLOCK_SYNC_LDS();
#endif
    }
    if (SPLIT) FWG_SETPRIO(PHYS ? FWG_PRIO_PHYS_STAGES : FWG_PRIO_GYM_EARLY);
    if (SPLIT && ext_act) {
        if (PHYS) {   // (the first thing this wave does with its rows)
            float4* h4 = reinterpret_cast<float4*>(hand);
            h4[0] = make_float4(E.y[13], E.y[14], E.y[15], E.y[16]);
            h4[1] = make_float4(E.y[17], raw[0], raw[1], raw[2]);
            fwg_msg_put(hand + 8, 0.f, 0.f, 0.f, FWG_TAG_RAW);
        } else {
            fwg_msg_take(hand + 8, FWG_TAG_RAW);
            const float4 q0 = reinterpret_cast<const float4*>(hand)[0], q1 = reinterpret_cast<const float4*>(hand)[1];
            act_q3 = make_float4(0.f, q0.x, q0.y, q0.z); act_q4 = make_float4(q0.w, q1.x, 0.f, 0.f);
            raw[0] = q1.y; raw[1] = q1.z; raw[2] = q1.w;
            load_cold(c, A.S, A.N, e, E);
        }
    }
    float ret_prev = 0.f;
    // The raw-action window (and the constrained-command window where observations use it) BY AGE: wa[k] = the action taken k
    // steps ago, wa[0] = this step's.  k_step2: plain loads from the arena's ring at the (wave-uniform) slots behind the current
    // one, straight into registers -- no LDS copy, no run-time slot arithmetic in the consumers, exact waits (behind a
    // global_load_lds the compiler turns every wait for a vector-memory result into vmcnt(0)).  One-wave kernel: read out of the
    // LDS copy streamed in at kernel start, once.
    // (plain float arrays: an array of float4 filled through a reinterpret_cast load ends up in scratch memory)
    float wa[FWG_MAX_WINDOW][3], wc[FWG_MAX_WINDOW][3];
    float4 int_old = make_float4(0.f, 0.f, 0.f, 0.f);   // integration_window: cumulative error sums W + 1 records back
    float4 sum_prev = make_float4(0.f, 0.f, 0.f, 0.f);  // the cumulative error sums through the previous record (packed Fix3)
    // rows that only the work AFTER the hand-over reads (k_step2: requested when the pre-hand-over work begins, see load_gym)
    auto late_rows = [&]() {
#ifndef FWG_ABL_NO_LATE_ROWS
        if (SPLIT) load_gym(c, A.S, A.N, e, E, A.bit_goal, 2);
#endif
        if (c.metrics) {   // S_(t-1): through the previous record
            int slot = A.slot_end - 1; slot += (slot < 0) ? FWG_END_RING : 0;
            sum_prev = CGROUP(A.S, A.N, (L.end_ring >> 2) + slot, e);
        }
        if (c.int_window) {   // S_(t-1-W): the record W + 1 positions before this step's
            int slot = A.slot_end - (c.int_window + 1); slot += (slot < 0) ? FWG_END_RING : 0;
            int_old = CGROUP(A.S, A.N, (L.end_ring >> 2) + slot, e);
        }
    };
    if (GYM) {
        // (k_step2: behind the arrival of the actuator groups -- the partner's rows, requested at the same moment, are served first)
        // second wave of requests: bookkeeping rows to registers, action windows and lagged observation rows HBM -> LDS;
        // issued only now so that the wait for the simulator state above does not have to drain them (vmcnt is in-order)
        // order = order of need (returns are in order): bookkeeping rows, action windows, lagged observation rows
#ifdef FWG_ABL_NO_LATE_ROWS
        constexpr bool LATE = false;
#else
        constexpr bool LATE = SPLIT;
#endif
        load_gym(c, A.S, A.N, e, E, A.bit_goal, LATE ? 1 : 3);
        // the cumulative error sums this step builds on (fixed point, fwgym_env.h Fix3): their ring slots are positions of the
        // GLOBAL step counter, so they are requested here, with the bookkeeping rows -- requested inside the pre-barrier work
        // (rounds 4a) their round trip was waited for at the end of it: every gym wave reached barrier A ~2k ticks later, which
        // is nothing for a plain wave (its partner integrates for longer) and the whole margin of a wave that also computes a
        // piece of the next draw, pads early rows or prefetches an episode end
        if (!LATE) late_rows();
        if (A.acc != nullptr && valid) ret_prev = A.acc_ret[e];   // attached rollout head: the env's discounted return so far
        // k_step2: the action windows by PLAIN loads, parked in LDS by this wave itself once they have landed (below).  Behind a
        // global_load_lds the compiler turns every wait for a vector-memory result into vmcnt(0); with plain loads only, the wait
        // for the actions and the actuator groups (first in the queue) is exact, and the actuators are advanced while the
        // bookkeeping rows and the windows are still on their way
        if (SPLIT) {
#pragma unroll
            for (int k = 1; k < FWG_MAX_WINDOW; ++k) {
                if (k < W) {
                    int slot = A.slot_act - k; slot += (slot < 0) ? W : 0;
                    const float4 qa = CGROUP(A.S, A.N, (L.act_ring >> 2) + slot, e);
                    wa[k][0] = qa.x; wa[k][1] = qa.y; wa[k][2] = qa.z;
                    if (c.use_cmd_ring) {
                        const float4 qc = CGROUP(A.S, A.N, (L.cmd_ring >> 2) + slot, e);
                        wc[k][0] = qc.x; wc[k][1] = qc.y; wc[k][2] = qc.z;
                    }
                }
            }
        }
    }
    // history["action"].append(action) (fixed_wing.py:345): the raw action enters the window first (HBM copy here, the
    // LDS copy once the streamed window has landed, i.e. after the integration)
    if (GYM && valid) GROUP(A.S, A.N, (L.act_ring >> 2) + A.slot_act, e) = make_float4(raw[0], raw[1], raw[2], 0.f);

    // ---- phase B: action scaling (fixed_wing.py:349-354,439-459) and the simulator step (fixed_wing.py:358)
    float cmd[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        cmd[i] = raw[i];
        if (c.scale_actions)
            cmd[i] = (V(c).act_to_high[i] - V(c).act_to_low[i]) * (fclampf(raw[i], V(c).scale_low, V(c).scale_high) - V(c).scale_low) *
                         V(c).inv_scale_span + V(c).act_to_low[i];
    }
    float cmd_c[3], sp[3];
    constrain_commands(c, cmd, cmd_c, sp);
    if (GYM && c.use_cmd_ring && valid)
        GROUP(A.S, A.N, (L.cmd_ring >> 2) + A.slot_act, e) = make_float4(cmd_c[0], cmd_c[1], cmd_c[2], 0.f);
    float gust[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (PHYS && TURB) {
        if (c.turb_increment) {
#pragma unroll
            for (int i = 0; i < 6; ++i) gust[i] = E.gust[i];
        } else {
            dryden_output(c, E.dry, gust);
        }
        if (c.sim_keys) {
#pragma unroll
            for (int i = 0; i < 6; ++i) gust[i] *= E.gust_gain;
        }
    }
    if (GYM) {
        if (SPLIT && ext_act) {   // the partner's actuator states at t + h/2 and t + h (it needs them after its first stage): computed
            // while the rows requested above are on their way (plain loads: the wait here covers the actions and the two actuator
            // groups only); the streamed windows are requested afterwards -- behind a global_load_lds every wait is a full drain
            const float a0[5] = {act_q3.y, act_q3.z, act_q3.w, act_q4.x, act_q4.y};
            // (two messages: the deflections at t + h/2 leave as soon as they exist -- the partner's second stage asks for them --,
            // the state at t + h follows; the fourth stage and the end of the step need it)
            float a_[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) a_[i] = a0[i];
            sanitize_actuators(c, a_);
            _Pragma("unroll") for (int m = 0; m < c.act_per_half; ++m) advance_actuators(c, a_, sp);
            fwg_msg_put(acts, a_[0], a_[1], a_[2], FWG_TAG_ACTS);
            FWG_TL(A, 24);
            _Pragma("unroll") for (int m = 0; m < c.act_per_half; ++m) advance_actuators(c, a_, sp);
            *reinterpret_cast<float4*>(acts + 4) = make_float4(a_[0], a_[1], a_[2], a_[3]);
            fwg_msg_put(acts + 8, a_[4], 0.f, 0.f, FWG_TAG_ACTS);
            FWG_SETPRIO(FWG_PRIO_GYM_PRE);
            FWG_TL(A, 25);
        }
        if (!SPLIT) {
            for (int s = 0; s < W; ++s) dma_group(&CGROUP(A.S, A.N, (L.act_ring >> 2) + s, e), lds + M.aring + s * (4 * FWG_WAVE));
            if (c.use_cmd_ring)
                for (int s = 0; s < W; ++s) dma_group(&CGROUP(A.S, A.N, (L.cmd_ring >> 2) + s, e), lds + M.cring + s * (4 * FWG_WAVE));
        }
#ifndef FWG_ABL_NO_LAG
        // (k_step2: requested by the gym wave once its bookkeeping rows and action windows have landed, see below)
        if (c.obs_log == 0) { if (!SPLIT) stream_lag_rows(c, A, e, lds + M.lag); }
        else log_wrap(c, A.obs, A.N, e, A.gnow, valid, A.log_wrap_now);
#endif
    }
    int fail = 0;
    if (PHYS) {
#ifdef FWG_ABL_NO_SIM
        E.d = derive<TURB>(E.y, E.wind, gust);
        if constexpr (SPLIT) {   // the partner still waits for its two messages
            float ea[5];
            euler_args(E.y, ea);
            const HandToGym hq{hand, &A};
            hq.state(E.y, ea);
            hq.result(E.d.Va, E.d.alpha, E.d.beta, 0);
        }
#else
        FWG_TL(A, 1);
        if constexpr (SPLIT) {
            // the wave team: actuators from the partner (after the first stage), the new state to the partner as soon as it exists,
            // constraint checks evaluated once after the last stage
            constexpr bool ONE = SpecCfg<SPEC>::one_rk4_step;
#ifdef FWG_ABL_NO_DEFER
            constexpr bool DEF = false;
#else
            constexpr bool DEF = ONE;
#endif
            typedef typename std::conditional<ONE, PartnerActuators, NoExtActuators>::type EXT;
#ifdef FWG_TIMELINE
            typedef StageStamps HOOK;
#else
            typedef NoStageHook HOOK;
#endif
            if (c.model_n > 0) fail = sim_step<TURB, EXT, Aero, HOOK, HandToGym, DEF>(c, la, E.y, sp, E.wind, gust, E.d, EXT{acts}, HOOK{A}, HandToGym{hand, &A});
            else fail = sim_step<TURB, EXT, DevCfg, HOOK, HandToGym, DEF>(c, V(c), E.y, sp, E.wind, gust, E.d, EXT{acts}, HOOK{A}, HandToGym{hand, &A});
        } else if (KT::generic || c.model_n > 0) {
            fail = sim_step<TURB, NoExtActuators, Aero>(c, la, E.y, sp, E.wind, gust, E.d);
        } else {
            fail = sim_step<TURB>(c, V(c), E.y, sp, E.wind, gust, E.d);
        }
#endif
        FWG_TL(A, 2);
        if (SPLIT) FWG_SETPRIO(FWG_PRIO_PHYS_TAIL);
        if (__ballot(fail != 0) != 0ull) {   // (rare) state was left untouched: derived values of the last valid state
            if (fail != 0) {
                E.d = derive<TURB>(E.y, E.wind, gust);
                if (TURB) {   // (air data: as the last committed step left them -- derived with its gust, see store_sim)
                    const float4 q = CGROUP(A.S, A.N, (c.L.derived >> 2) + 1, e);
                    E.d.alpha = q.x; E.d.beta = q.y; E.d.Va = q.z;
                }
                if (SPLIT) {   // ... which replace, for the partner, what the failing lane sent before it knew
                    float4* h4 = reinterpret_cast<float4*>(hand);
                    h4[0] = make_float4(E.y[4], E.y[5], E.y[6], E.y[7]);
                    h4[1] = make_float4(E.y[8], E.y[9], E.y[10], E.y[11]);
                    h4[2] = make_float4(E.y[12], E.y[13], E.y[14], E.y[15]);
                    h4[3] = make_float4(E.d.roll, E.d.pitch, E.d.yaw, E.d.Va);
                    fwg_msg_put(hand + 24, E.d.alpha, E.d.beta, 0.f, FWG_TAG_OLD);
                }
            }
        }
    }
    float n[4] = {0.f, 0.f, 0.f, 0.f};   // the step's four standard normals for the Dryden filter
    const unsigned steps_before = E.steps;   // (gym wave: the index of this step inside its episode)
    if (TURB && GYM && !SPLIT) {   // (counters: the step and episode indices BEFORE this step; independent of the integration)
        const u4 b = philox4x32((unsigned)(A.env_base + e), E.steps, E.episode, FWG_STREAM_TURB, A.seed_lo, A.seed_hi);
        box_muller(b, n);
    }
    // ---- the part of the gym bookkeeping that does not depend on this step's integration (it needs the streamed action
    // windows only): in the split kernel the gym wave does it while the physics wave integrates
    bool done = false;
    unsigned term = FWG_TERM_NONE;
    float fval_action[FWG_MAX_FACTORS];   // values of the reward factors of class "action" (fixed_wing.py:686-700)
    float obs_action[FWG_MAX_OBS];        // "action" entries of the newest observation row (fixed_wing.py:813-828)
    float tgt_next[3] = {0.f, 0.f, 0.f};  // targets propagated by one step, valid unless the target is resampled
    // (the next episode's reset draw is prepared here too, piece by piece, in the steps after each reset).  An episode that
    // ends at steps_max is known before the integration: everything its episode-end branch reads from memory -- the
    // prepared draw, the end-error record, the lagged rows of the terminal observation -- is requested here, so that the
    // round trips run while the physics wave integrates
    bool pre_end = false, pre_draw = false, pre_rows = false;
    float4 pre_tag = make_float4(0.f, 0.f, 0.f, 0.f), pre_old = make_float4(0.f, 0.f, 0.f, 0.f);
    Fix3 S_prev = {{0ll, 0ll, 0ll}};                    // ... unpacked at the end of the pre-barrier work (exact: fixed point)
    ResetDraw RD;
    // lanes in the first steps of an episode: the padding rows of their observation, up to the "action" entries
    bool pre_early = false;
    float early_noise[FWG_MAX_ROWS] = {};
    unsigned tail_early = 0u;
    float sdcmd_add = 0.f;   // control variation of this step (added to the episode's sum once its row is here)
    auto gym_prework = [&]() {
        FWG_TL(A, 16);
#ifndef FWG_ABL_NO_LATE_ROWS
        if (SPLIT) late_rows();
#endif
        if (!SPLIT) {   // one-wave kernel: the windows out of their LDS copy (landed: dma_wait above), by age
#pragma unroll
            for (int k = 1; k < FWG_MAX_WINDOW; ++k) {
                if (k < W) {
                    int slot = A.slot_act - k; slot += (slot < 0) ? W : 0;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        wa[k][i] = aring[slot * (4 * FWG_WAVE) + i];
                        if (c.use_cmd_ring) wc[k][i] = cring[slot * (4 * FWG_WAVE) + i];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) { wa[0][i] = raw[i]; wc[0][i] = cmd_c[i]; }   // the own action enters the windows
        if (c.metrics) {  // control_variation accumulator (fixed_wing.py:1109-1114)
            if (E.steps > 0u && W > 1) {   // the previous constrained command is recomputed from the previous raw action in the window
                float pc[3];
                if (c.use_cmd_ring) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) pc[i] = wc[1][i];
                } else {
                    float praw[3], psp[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float r_ = wa[1][i];
                        praw[i] = c.scale_actions ? (V(c).act_to_high[i] - V(c).act_to_low[i]) * (fclampf(r_, V(c).scale_low, V(c).scale_high) - V(c).scale_low) *
                                                        V(c).inv_scale_span + V(c).act_to_low[i]
                                                  : r_;
                    }
                    constrain_commands(c, praw, pc, psp);
                }
                sdcmd_add = fabsf(cmd_c[0] - pc[0]) + fabsf(cmd_c[1] - pc[1]) + fabsf(cmd_c[2] - pc[2]);
            }
        }
        FWG_TL(A, 17);
        E.steps += 1u;
        // lanes in the first steps of an episode: their record 0 is requested now, the padding rows follow at the end of this
        // block (row-log mode without an observer: the partner's tail work instead, tail_rows)
        const bool early_now = c.obs_length > 1 && valid && (int)E.steps <= (c.obs_length - 1) * c.obs_step;
        float rec0[FWG_MAX_OBS];
        if (!tail_rows && __ballot(early_now) != 0ull) {
            if (early_now) {
                int slot0 = 0;
                if (c.obs_log == 0) { slot0 = A.slot_lag - (int)E.steps; slot0 += (slot0 < 0) ? c.L.lag_depth : 0; }
                early_rows_request(c, A, e, rec0, slot0);
            }
        }
        E.sft += 1u;
        if (V(c).steps_max > 0 && E.steps >= (unsigned)V(c).steps_max) { done = true; term = FWG_TERM_STEPS; }
        // an episode that ends at steps_max is known here: everything its episode-end branch reads from memory is requested
        // now -- the prepared draw, the end-error record, the lagged rows of the terminal observation --, so that the round
        // trips run under the rest of this block and the partner's integration
        bool end_in_wave = false;   // (wave-uniform)
        if (c.auto_reset && V(c).steps_max > 0 && __ballot(valid && done) != 0ull) {
            end_in_wave = true;
#ifndef FWG_ABL_NO_END_PRIO
            // a wave that hosts an episode end has the longest way to go of all the launch's waves -- the launch ends when the
            // last of them does: it issues ahead of its SIMD's other wave from here on
            if (SPLIT) FWG_SETPRIO(FWG_PRIO_GYM_END);
#endif
            if (valid && done) {
                pre_end = true;
                if (draw_stage_of(E.flags) == FWG_DRAW_READY) {
                    pre_draw = true;
                    pre_tag = draw_tag(A.S, A.N, e, c);
                    draw_load_final(c, A.S, A.N, e, RD);   // used only if the tag checks out
                }
                if (c.metrics) {
                    int slot = A.slot_end + 1; slot -= (slot >= FWG_END_RING) ? FWG_END_RING : 0;
                    pre_old = CGROUP(A.S, A.N, (L.end_ring >> 2) + slot, e);
                }
                if (c.obs_log > 0 && !partner_rows) log_load_rows(c, A.obs, A.N, e, A.log_win, ob);
            }
        }
        FWG_TL(A, 18);
#pragma unroll
        for (int f = 0; f < FWG_MAX_FACTORS; ++f) {
            fval_action[f] = 0.f;
            if (f >= c.n_factors) continue;
            const DevFactor& F = c.factor[f];
            const DevFactor& FV = V(c).factor[f];   // (its values: fwgym_dev.h cfg_values)
            if (F.cls != FWG_RC_ACTION) continue;
            float val = 0.f;
            if (F.type == FWG_RT_VALUE) val = fabsf(raw[0]) + fabsf(raw[1]) + fabsf(raw[2]);
            else if (F.type == FWG_RT_DELTA) {
                if (E.steps > 1u) {
                    const int m = (int)min(E.steps, (unsigned)F.window);
#pragma unroll
                    for (int k = FWG_MAX_WINDOW - 2; k >= 0; --k) {
                        if (k <= W - 2 && k <= m - 2) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) val += fabsf(wa[k][i] - wa[k + 1][i]);
                        }
                    }
                }
            } else {  // bound
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    val += raw[i] > V(c).act_bound_max[i] ? raw[i] - V(c).act_bound_max[i] : 0.f;
                    val += raw[i] < V(c).act_bound_min[i] ? V(c).act_bound_min[i] - raw[i] : 0.f;
                }
            }
            fval_action[f] = val;
        }
        FWG_TL(A, 19);
#pragma unroll
        for (int j = 0; j < FWG_MAX_OBS; ++j) {
            obs_action[j] = 0.f;
            if (j < c.n_obs && c.obs[j].type == FWG_OBS_ACTION)   // E.steps >= 1 here: never the "no action yet" branch
                obs_action[j] = action_obs_win(c, c.use_cmd_ring ? wc : wa, c.obs[j].src, c.obs[j].window, E.steps);
        }
        FWG_TL(A, 20);
        {   // targets advanced by one step (fixed_wing.py:401-404) for the common case that none is resampled this step
            float keep[3] = {E.tgt[0], E.tgt[1], E.tgt[2]};
            next_targets(c, E);
#pragma unroll
            for (int k = 0; k < 3; ++k) { tgt_next[k] = E.tgt[k]; E.tgt[k] = keep[k]; }
        }
        FWG_TL(A, 21);
        if (c.auto_reset) {   // one piece of the NEXT episode's reset draw, for lanes that do not have it yet.  At most ONE
            // kind of piece per wave and step (the least advanced lanes first): a wave whose lanes sit at different stages
            // would otherwise run all the pieces back to back and outlast the integration it hides behind
            const unsigned stage = draw_stage_of(E.flags);
            // (not in the first steps of an episode, whose lanes have their padding rows to compute in this same interval)
            const bool work = valid && stage < FWG_DRAW_READY && !early_now;
            unsigned long long m = 0ull;
            unsigned pick = FWG_DRAW_READY;
#pragma unroll
            for (unsigned sidx = 0; sidx < FWG_DRAW_READY; ++sidx) {
                const unsigned long long ms = __ballot(work && stage == sidx);
                if (pick == FWG_DRAW_READY && ms != 0ull) { pick = sidx; m = ms; }
            }
#ifdef FWG_ABL_NO_STAGED_DRAW
            m = 0ull;
#endif
#ifndef FWG_ABL_DRAW_AT_END
            if (end_in_wave) m = 0ull;   // (not on top of an episode end's work: the piece waits one step, episodes last hundreds)
#endif
            if (m != 0ull) {
                if (work && stage == pick) {
                    const unsigned ns = draw_stage_step(c, dc, A, e, E.episode, stage, T);
                    E.flags = (E.flags & ~FWG_DRAW_STAGE_MASK) | (ns << FWG_DRAW_STAGE_SHIFT);
                }
            }
        }
        FWG_TL(A, 22);
        if (!tail_rows && __ballot(early_now) != 0ull) {
            if (early_now) { pre_early = true; early_rows_pre(c, A, e, E, ob, rec0, early_noise); }
        }
        // foreseen episode end with a valid prepared draw: the partner installs the new episode (see above); this wave only
        // needs to know that it does.  Both waves decide from the same words (stage, tag, generation, episode)
        if (pre_install) pre_rows = pre_draw && f2u(pre_tag.x) == dc.generation && f2u(pre_tag.y) == E.episode + 1u;
        tail_early = (tail_rows && early_now) ? E.steps : 0u;   // (for the partner: this step's index if its rows need padding)
        // (unpacked here, in the time this wave waits for its partner anyway, not on the chain after the barrier)
        if (c.metrics) { S_prev = fix3_unpack(sum_prev); E.sdcmd += sdcmd_add; }
        FWG_TL(A, 23);
    };
    if (SPLIT && GYM) {
        FWG_TL(A, 1);
        // dense batch: the lagged observation rows (12 KiB per wave, straight from HBM) are requested only NOW and land under the
        // pre-barrier work below, which does not read them.  Requested with the other rows they delay everything: behind a
        // global_load_lds the compiler turns every wait for a vector-memory result into vmcnt(0) ("pending flat" in its counter
        // model), so the gym wave sat through all of them before its first instruction (~6.6k ticks into the kernel instead of
        // ~3.6k) and reached barrier A 1.7k ticks AFTER its partner (tools/timeline.py).  Measured against this, same box:
        // plain streaming loads parked in LDS +0.3 us per step, straight into the record's registers 238 -> 255 VGPRs and
        // spills, requests from inline assembly (not counted by the compiler, explicit partial wait) +0.15 us, windows AND rows
        // by plain loads parked by the wave itself (exact waits, no global_load_lds at all): 256 VGPRs + 8 spills, +0.5 us.
#ifndef FWG_ABL_NO_LAG
        if (c.obs_log == 0) stream_lag_rows(c, A, e, lds + M.lag);
#endif
        FWG_WAVE_SYNC();
        gym_prework();
        if (c.obs_log == 0 && c.obs_length > 1) FWG_DMA_DRAIN();   // (landed by now; drains the rare stores of a draw piece too)
        // a foreseen end's operands, requested above, are pinned in registers here (see FWG_TOUCH: the episode-end branch then
        // runs without a wait); the terminal observation's lagged rows MUST have landed before the barrier -- the partner
        // writes the new window over their planes after it.  (Not a drain: that would sit through the acknowledgement of
        // the stores issued since, e.g. a neighbour lane's draw piece)
        if (__ballot(pre_end) != 0ull) {
            if (pre_end) {   // (only the lanes that requested them: the others' copies are indeterminate)
                touch4(pre_tag); touch4(pre_old);
                if (pre_draw) touch_draw(c, RD);
                if (c.obs_log > 0 && !partner_rows) {
#pragma unroll
                    for (int i = 0; i < FWG_MAX_OBS * FWG_MAX_ROWS; ++i)
                        if (i >= c.n_obs && i < c.obs_dim) { const float v = ob.get(i); FWG_TOUCH(v); }
                }
            }
        }
        FWG_TL(A, 2);
    }
    // ---- the two waves meet (k_step2).  Gym wave: what its partner's tail work needs leaves first (the step's index for the
    // turbulence noise, whose rows need padding, whether the partner installs a foreseen end's next episode -- after the touches
    // above: the terminal observation's lagged rows are in registers before the partner may overwrite their planes); then the
    // candidate state, whose Euler angles this wave computes while the partner evaluates airspeed, incidence and the checks;
    // then those.  Physics wave: it waits for the tail message only, and has nothing else to do by then.
    if (SPLIT && GYM) {
        // (what the pre-hand-over work produced exists before the wait: left alone, arithmetic whose results are needed only after
        // the hand-over is sunk below the polling loop -- onto the chain that starts when the partner's state arrives)
#pragma unroll
        for (int f = 0; f < FWG_MAX_FACTORS; ++f) if (f < c.n_factors && c.factor[f].cls == FWG_RC_ACTION) fwg_pin(fval_action[f]);
#pragma unroll
        for (int j = 0; j < FWG_MAX_OBS; ++j) if (j < c.n_obs && c.obs[j].type == FWG_OBS_ACTION) fwg_pin(obs_action[j]);
        fwg_pin(tgt_next[0], tgt_next[1], tgt_next[2], E.sdcmd, E.steps, E.sft, E.flags);
        if (c.metrics) fwg_pin(S_prev.s[0], S_prev.s[1], S_prev.s[2]);
        fwg_msg_put(tailm, u2f(steps_before), u2f(tail_early), u2f(pre_rows ? 1u : 0u), FWG_TAG_TAIL);
        const float4 e4 = fwg_msg_take(hand + 16, FWG_TAG_STATE);
        FWG_SETPRIO(FWG_PRIO_GYM_POST);
        FWG_TL(A, 25);
        const float4* h4 = reinterpret_cast<const float4*>(hand);
        const float4 a = h4[0], b = h4[1], g = h4[2], ea4 = h4[3];
        FWG_TL(A, 3);
        E.y[4] = a.x; E.y[5] = a.y; E.y[6] = a.z; E.y[7] = a.w; E.y[8] = b.x; E.y[9] = b.y; E.y[10] = b.z; E.y[11] = b.w;
        E.y[12] = g.x; E.y[13] = g.y; E.y[14] = g.z; E.y[15] = g.w;
        const float ea[5] = {ea4.x, ea4.y, ea4.z, ea4.w, e4.x};
        euler_from_args(ea, E.d);
        const float4 r = fwg_msg_take(hand + 20, FWG_TAG_RESULT, 0xFF000000u);
        E.d.Va = r.x; E.d.alpha = r.y; E.d.beta = r.z;
        fail = (int)(f2u(r.w) & 0xFFu);
        if (pre_install) pre_rows = pre_rows && fail == 0;   // (a foreseen end whose last step fails: nothing installed, the unforeseen path)
        if (__ballot(fail != 0) != 0ull) {   // (rare) a failed step: the last valid state and its derived values follow
            const float4 o = fwg_msg_take(hand + 24, FWG_TAG_OLD, 0xFFFFFFFFu, fail != 0);
            if (fail != 0) {
                const float4 a2 = h4[0], b2 = h4[1], g2 = h4[2], d2 = h4[3];
                E.y[4] = a2.x; E.y[5] = a2.y; E.y[6] = a2.z; E.y[7] = a2.w; E.y[8] = b2.x; E.y[9] = b2.y; E.y[10] = b2.z; E.y[11] = b2.w;
                E.y[12] = g2.x; E.y[13] = g2.y; E.y[14] = g2.z; E.y[15] = g2.w;
                E.d.roll = d2.x; E.d.pitch = d2.y; E.d.yaw = d2.z; E.d.Va = d2.w; E.d.alpha = o.x; E.d.beta = o.y;
            }
        }
    }
    // physics wave: what its tail work needs from memory is requested first thing after the hand-over
    unsigned steps_p = 0u;
    bool early_p = false, end_p = false, coop_rows = false;
    unsigned long long end_mask = 0ull;
    float coop_word[FWG_COOP_ENDS] = {};
    float rec0_p[FWG_MAX_OBS];
    ResetDraw RDp;
    if (SPLIT && PHYS) {
        const float4 w = fwg_msg_take(tailm, FWG_TAG_TAIL);
        if (tail_rows || pre_install) {
            steps_p = f2u(w.y);
            end_p = pre_install && valid && f2u(w.z) != 0u && fail == 0;
            early_p = tail_rows && valid && fail == 0 && steps_p != 0u && !end_p;   // (an ending lane's rows are the partner's)
            if (__ballot(early_p) != 0ull) {
                if (early_p) early_rows_request(c, A, e, rec0_p);
            }
            end_mask = __ballot(end_p);
            if (end_mask != 0ull) {
                if (end_p) draw_load_final(c, A.S, A.N, e, RDp);   // (just read by the partner: served from the cache)
                // partner_rows: the lagged rows of the ending lanes' terminal observations, requested now, stored before the new
                // window goes over their planes.  Up to FWG_COOP_ENDS ending lanes: lane i takes word i of each one's rows
                if (partner_rows && A.term_obs != nullptr) {
                    const int lag_words = (c.obs_length - 1) * c.n_obs;
                    coop_rows = lag_words <= FWG_WAVE && __popcll(end_mask) <= FWG_COOP_ENDS;
                    if (coop_rows) {
                        unsigned long long mm = end_mask;
#pragma unroll
                        for (int k = 0; k < FWG_COOP_ENDS; ++k) {
                            if (mm != 0ull) {
                                const long el = env0 + (__ffsll((long long)mm) - 1);
                                mm &= mm - 1ull;
                                if (lane < lag_words) coop_word[k] = A.obs[((A.log_win + 1) * A.N + el) * c.n_obs + (long)(lane / c.n_obs) * A.N * c.n_obs + lane % c.n_obs];
                            }
                        }
                    }
                }
            }
        }
        // hand-shake B, first mark: this wave has read its partner's messages (LDS executes a wave's accesses in order), the
        // partner may re-use the area as the output staging area
        FWG_EMU_WAVE_SYNC();
        if (lane == 0) FWG_FLAG_RAISE(mark, 1);   // (asm with a memory clobber: the reads above stay above)
        if (TURB) {   // the step's four standard normals for the Dryden filter (counters: the step and episode indices BEFORE this step)
            const u4 b = philox4x32((unsigned)(A.env_base + e), f2u(w.x), E.episode, FWG_STREAM_TURB, A.seed_lo, A.seed_hi);
            box_muller(b, n);
        }
    }
    const bool ok = fail == 0;
    if (PHYS) {
        if (TURB && ok) {
            float x_old[FWG_N_DRYDEN];
#pragma unroll
            for (int i = 0; i < FWG_N_DRYDEN; ++i) x_old[i] = E.dry[i];
            dryden_advance(c, E.dry, n);
            if (c.turb_increment) dryden_next_gust(c, x_old, E.dry, E.gust);
        }
#ifndef FWG_ABL_NO_SIMSTORE
        // (foreseen episode end: the old episode's final state went to the partner through LDS; the NEXT episode's rows follow below)
        if (SPLIT && c.store_derived && !(c.con_mask & 0x7u)) {   // the host views' Euler angles: the partner computed its own copy
            float ea[5];
            euler_args(E.y, ea);
            euler_from_args(ea, E.d);
        }
        if (valid && !end_p) store_sim<TURB>(c, A.S, A.N, e, E);
#endif
        FWG_TL(A, 3);
        if (SPLIT && tail_rows && __ballot(early_p) != 0ull) {
            if (early_p) {   // (fail == 0: a failed step's rows are the gym wave's)
                float rn[FWG_MAX_ROWS];
                early_row_noise(c, A, e, steps_p, E.episode, rn);
                const float actuator[3] = {0.5f * (E.y[13] + E.y[14]), 0.5f * (E.y[14] - E.y[13]), E.y[15]};
                early_rows_to_log(c, A, e, steps_p, rec0_p, rn, actuator, A.log_win);
            }
        }
    }
    if (SPLIT && PHYS) {
        // foreseen episode end: the NEXT episode goes into the simulator / cold rows and the row log (the old episode's final
        // state went to the partner through LDS; its terminal observation was requested before barrier A; the lanes are
        // disjoint from the ones the partner re-initialises itself)
        if (pre_install && end_mask != 0ull) {
            if (partner_rows && A.term_obs != nullptr) {   // the old window's rows -> the terminal batch, before the new window lands on them
                if (coop_rows) {
                    const int lag_words = (c.obs_length - 1) * c.n_obs;
                    unsigned long long mm = end_mask;
#pragma unroll
                    for (int k = 0; k < FWG_COOP_ENDS; ++k) {
                        if (mm != 0ull) {
                            const long el = env0 + (__ffsll((long long)mm) - 1);
                            mm &= mm - 1ull;
                            if (lane < lag_words) A.term_obs[el * c.obs_dim + c.n_obs + lane] = coop_word[k];
                        }
                    }
                } else if (end_p) {   // many ending lanes (synchronised episodes): every lane its own rows, one row at a time
                    for (int r = 1; r < c.obs_length; ++r) {
                        const float* src = log_row(c, A.obs, A.N, e, A.log_win + r);
                        float* dst = A.term_obs + e * c.obs_dim + r * c.n_obs;
                        if (((c.obs_dim | c.n_obs) & 3) == 0) {   // (rare: no staging of the groups in a register array)
                            for (int i = 0; 4 * i < c.n_obs; ++i) {
                                const float4 q = reinterpret_cast<const float4*>(src)[i];
                                reinterpret_cast<float4*>(dst)[i] = q;
                            }
                        } else {
                            for (int j = 0; j < c.n_obs; ++j) dst[j] = src[j];
                        }
                    }
                }
            }
            if (end_p) {   // (the partner checked the draw's tag)
                reset_rows_to_log(c, A, e, RDp, aring, A.slot_lag, A.log_win, c.obs_log == 0);
#pragma unroll
                for (int i = 0; i < NY; ++i) E.y[i] = RDp.y[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) E.wind[i] = RDp.wind[i];
                E.episode += 1u;
#pragma unroll
                for (int i = 0; i < FWG_N_DRYDEN; ++i) E.dry[i] = 0.f;
#pragma unroll
                for (int i = 0; i < 6; ++i) E.gust[i] = 0.f;
                E.gust_gain = RDp.gust_gain;
                E.d = RDp.d;   // (store_sim writes the derived host views with store_derived)
                store_cold(c, A.S, A.N, e, E);
                store_sim<TURB>(c, A.S, A.N, e, E);
            }
        }
        // hand-shake B, second mark, one way: everything above is in memory.  The gym wave waits for it only in a wave that
        // re-initialises an env itself (an episode that ended unforeseen: it overwrites this launch's simulator rows, and an
        // early lane's padding rows); this wave waits for nobody
        FWG_DMA_DRAIN();   // s_waitcnt vmcnt(0): covers stores as well
        FWG_WAVE_SYNC();
        if (lane == 0) FWG_FLAG_RAISE(mark, 2);
        return;
    }
    // everything streamed HBM -> LDS at kernel start is needed from here on; the integration above hid its latency
    if (!SPLIT) { dma_wait(); gym_prework(); }
    FWG_TL(A, 4);

    // ---- phase C: gym-side bookkeeping (fixed_wing.py:360-417)
    fill_vars(E, T);
    float err[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < FWG_MAX_TARGETS; ++k)
        if (k < c.n_targets) err[k] = target_error(c.target[k], E.tgt[k], T.get(c.target[k].var));
    float reward = 0.f;
    const unsigned rec = E.steps;  // index of the record this step appends to the episode histories
    // integration_window (fixed_wing.py:708-711, 804-810): the sum of the W errors BEFORE the newest one, from the episode's
    // cumulative sums: S_(t-1) (E.esum has not taken this step's error yet) - S_(t-1-W), padded with the initial error while
    // the episode is younger than the window.  After a failed step the histories are one record shorter: S_(t-2) - S_(t-2-W)
    float wsum[3] = {0.f, 0.f, 0.f};
    if (c.int_window) {
        const int t = (int)E.steps, W = c.int_window;
        if (ok) {
            const Fix3 o = fix3_unpack(int_old);
#pragma unroll
            for (int k = 0; k < 3; ++k) wsum[k] = fix_to_float(fix_wrap(S_prev.s[k] - ((t - 1 - W >= 0) ? o.s[k] : 0ll)));
        } else {   // S_(t-2) = S_(t-1) - the previous record's error (E.perr is the very float that was quantised into it)
            Fix3 o2 = {{0ll, 0ll, 0ll}};
            if (t - 2 - W >= 0) {
                int slot = A.slot_end - (W + 2); slot += (slot < 0) ? FWG_END_RING : 0;
                o2 = fix3_unpack(CGROUP(A.S, A.N, (L.end_ring >> 2) + slot, e));
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) wsum[k] = fix_to_float(fix_wrap(S_prev.s[k] - fix_quant(E.perr[k]) - o2.s[k]));
        }
#pragma unroll
        for (int k = 0; k < FWG_MAX_TARGETS; ++k) {
            if (k < c.n_targets) {
                T.put(FWG_TAB_INT + k, wsum[k] + (float)max(0, W - (t - 1)) * E.e0[k]);
                E.int_reset[k] = wsum[k] + (float)(W + 1) * E.e0[k];   // what a reset after THIS step would show
            }
        }
    }
    if (ok) {
        bool achieved_now = false, resample = false;
        unsigned g = 0u;
        if (c.goal_enabled) {
            g = goal_flags(c, err);
            goal_push(c, E, g, A.bit_goal, rec);
            if (E.sft >= (unsigned)c.streak_req && window_count(E, 3) >= (unsigned)V(c).streak_min_count) {
                achieved_now = !(E.flags & FWG_FLAG_GOAL_ACHIEVED);
                E.flags |= FWG_FLAG_GOAL_ACHIEVED;
                if (c.on_success == FWG_ON_SUCCESS_DONE) { done = true; term = FWG_TERM_SUCCESS; }
                else if (c.on_success == FWG_ON_SUCCESS_NEW) resample = true;
            }
        }
        // ---- reward (fixed_wing.py:674-774)
        float nv[3] = {0.f, 0.f, 0.f}, sh[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < FWG_MAX_FACTORS; ++f) {
            if (f >= c.n_factors) continue;
            const DevFactor& F = c.factor[f];
            const DevFactor& FV = V(c).factor[f];   // (its values: fwgym_dev.h cfg_values)
            float val = 0.f;
            if (F.cls == FWG_RC_ACTION) {
                val = fval_action[f];
            } else if (F.cls == FWG_RC_STATE) {
                if (F.type == FWG_RT_INT_ERROR)
                    val = (F.src == 0 ? wsum[0] : (F.src == 1 ? wsum[1] : wsum[2])) + (float)max(0, c.int_window - (int)E.steps) * E.e0[F.src < 3 ? F.src : 0];
                else
                    val = (F.type == FWG_RT_VALUE) ? T.get(F.src) : (F.src == 0 ? err[0] : (F.src == 1 ? err[1] : err[2]));
            } else if (F.cls == FWG_RC_SUCCESS) {
                val = achieved_now ? (F.value_is_timesteps ? (float)(V(c).steps_max - (int)E.steps) : FV.value) : 0.f;
            } else if (F.cls == FWG_RC_STEP) {
                val = FV.value;
            } else {  // goal
                if (F.type == FWG_RT_PER_STATE) {
#pragma unroll
                    for (int k = 0; k < FWG_MAX_TARGETS; ++k)
                        if (k < c.n_targets && c.target[k].has_bound && ((g >> k) & 1u)) val += FV.value / (float)c.n_targets;
                } else {
                    val = (g & 8u) ? FV.value : 0.f;
                }
            }
            const float inv_scaling = c.randomize_scaling ? E.fscale[f] : FV.inv_scaling;   // per env and episode / fixed
            if (F.fclass == FWG_FC_LINEAR) {
                val = fabsf(val) * inv_scaling;
                if (F.has_max) val = fminf(val, FV.max);
            } else {
                val = val * val * inv_scaling;
            }
            val *= FV.sign;
            const int fc = F.fclass;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                sh[i] += (i == fc && F.shaping) ? val : 0.f;
                nv[i] += (i == fc && !F.shaping) ? val : 0.f;
            }
        }
#pragma unroll
        for (int fc = 0; fc < 3; ++fc) {
            if (c.term_present[fc]) {
                const bool prev_ok = (E.flags >> (FWG_FLAG_PREV_VALID_SHIFT + fc)) & 1u;
                float v;
                if (fc == FWG_FC_EXPONENTIAL) {
                    float arg = nv[fc];
                    if (c.reward_potential) { if (prev_ok) arg += sh[fc] - E.psh[fc]; }
                    else arg += sh[fc];
                    v = expf(arg) - 1.f;
                } else {
                    v = nv[fc];
                    if (c.reward_potential) { if (prev_ok) v += sh[fc] - E.psh[fc]; }
                    else v += sh[fc];
                }
                E.psh[fc] = sh[fc];
                E.flags |= 1u << (FWG_FLAG_PREV_VALID_SHIFT + fc);
                reward += V(c).term_weight[fc] * v;
            }
        }
        // ---- target resampling / propagation (fixed_wing.py:397-404)
        if (resample || (V(c).resample_every > 0 && E.sft >= (unsigned)V(c).resample_every)) {
            sample_targets(c, dc, A, e, E, T, nullptr);
            next_targets(c, E);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) E.tgt[k] = tgt_next[k];
        }
#pragma unroll
        for (int k = 0; k < FWG_MAX_TARGETS; ++k)
            if (k < c.n_targets) err[k] = target_error(c.target[k], E.tgt[k], T.get(c.target[k].var));
        if (c.metrics) {  // streaming form of history["error"] (fixed_wing.py:1095-1157)
#pragma unroll
            for (int k = 0; k < FWG_MAX_TARGETS; ++k) {
                if (k >= c.n_targets) continue;
                const float lo_lim = fabsf(V(c).rise_low * E.e0[k]), hi_lim = fabsf(V(c).rise_high * E.e0[k]);
                const float pa = fabsf(E.perr[k]), ca = fabsf(err[k]);
                if ((E.rise[k] & 0xFFFFu) == 0xFFFFu && pa >= lo_lim && ca < lo_lim) E.rise[k] = (E.rise[k] & 0xFFFF0000u) | (rec - 1u);
                if ((E.rise[k] >> 16) == 0xFFFFu && pa >= hi_lim && ca < hi_lim) E.rise[k] = (E.rise[k] & 0xFFFFu) | ((rec - 1u) << 16);
                E.esum[k] += err[k]; E.eabs[k] += fabsf(err[k]);
                E.emin[k] = fminf(E.emin[k], err[k]); E.emax[k] = fmaxf(E.emax[k], err[k]);
                E.perr[k] = err[k];
            }
            // the ring holds the episode's CUMULATIVE error sums: the sum over the last 50 records is then the difference
            // of two entries (one load at the episode end instead of the whole window)
            Fix3 S_new;
#pragma unroll
            for (int k = 0; k < 3; ++k) S_new.s[k] = S_prev.s[k] + (k < c.n_targets ? fix_quant(err[k]) : 0ll);
            // (a plain store: the NEXT step reads this very slot back -- a streaming store would push it out of the L2 first)
            if (valid) GROUP(A.S, A.N, (L.end_ring >> 2) + A.slot_end, e) = fix3_pack(S_new);
            S_prev = S_new;   // (from here on: the sums through the LAST record of the histories, whether this step appended one or not)
        }
    } else {
        done = true;
        if (c.int_window) E.flags |= FWG_FLAG_LAST_FAILED;
        reward = c.step_fail_timesteps ? (float)((int)E.steps - V(c).steps_max) : V(c).step_fail_value;
        term = FWG_TERM_VAR0 + (unsigned)(fail - 1);
    }
#pragma unroll
    for (int k = 0; k < FWG_MAX_TARGETS; ++k) {
        if (k < c.n_targets) {
            T.put(FWG_TAB_TGT + k, E.tgt[k]);
            T.put(FWG_TAB_ERR + k, err[k]);
        }
    }

    FWG_TL(A, 5);
    // An episode end that was NOT foreseen (failure, success; or a foreseen one whose last step failed) is known here: what its
    // end branch reads from memory is requested and waited for NOW, before the bookkeeping stores go out -- vmcnt counts in
    // issue order, so a load issued after them would sit through the acknowledgement of every one of them (2-3k ticks)
    // where this wait costs one round trip.  The same registers as the foreseen end's prefetches: one end branch for both
    const bool late_end = done && valid && !(pre_end && ok);
    // (overwrites what early_rows_pre prepared; a foreseen end whose lagged rows were left to the partner, which then did not take them)
    const bool reload = c.obs_log > 0 && valid && (pre_end ? (partner_rows && !pre_rows) : (done || !ok));
    if (__ballot(late_end || reload) != 0ull) {
        if (late_end || reload) {
            if (c.metrics && late_end) {
                int slot = ok ? A.slot_end + 1 : A.slot_end;
                slot -= (slot >= FWG_END_RING) ? FWG_END_RING : 0;
                pre_old = CGROUP(A.S, A.N, (L.end_ring >> 2) + slot, e);
            }
            if (c.auto_reset && late_end && !pre_end && draw_stage_of(E.flags) == FWG_DRAW_READY) {
                pre_draw = true;
                pre_tag = draw_tag(A.S, A.N, e, c);
                draw_load_final(c, A.S, A.N, e, RD);   // used only if the tag checks out
            }
            if (reload) log_load_rows(c, A.obs, A.N, e, A.log_win, ob);
            touch4(pre_tag); touch4(pre_old);
            if (pre_draw) touch_draw(c, RD);
            if (c.obs_log > 0 && (reload || (pre_end && !partner_rows))) {
#pragma unroll
                for (int i = 0; i < FWG_MAX_OBS * FWG_MAX_ROWS; ++i)
                    if (i >= c.n_obs && i < c.obs_dim) { const float v = ob.get(i); FWG_TOUCH(v); }
            }
        }
    }
#ifndef FWG_ABL_NO_GYMSTORE
    store_gym(c, A.S, A.N, e, E, A.bit_goal, valid, false);
#endif
    // the step's scalar outputs are final here: issued now, their write latency hides behind the observation build
    // (info["target"] of a lane that ends its episode is the target BEFORE the reset, fixed_wing.py:435)
    if (valid) {
        A.rew[e] = reward;
        A.done[e] = done ? 1 : 0;
        A.term[e] = (uint8_t)term;
        if (A.tgt_out != nullptr) {
#pragma unroll
            for (int k = 0; k < FWG_MAX_TARGETS; ++k)
                if (k < c.n_targets) A.tgt_out[e * c.n_targets + k] = E.tgt[k];
        }
    }
    FWG_TL(A, 6);

    // ---- phase D: observation (fixed_wing.py:776-846)
#ifndef FWG_ABL_NO_LAG
    if (c.obs_log == 0) load_lag_rows(c, lds + M.lag + lane * 4, ob, pre_early ? (int)E.steps : (1 << 30));
#endif
    build_row0(c, A, e, E, T, ob, c.use_cmd_ring ? cring : aring, A.slot_lag, ok && valid && c.obs_log == 0 && !pre_rows, A.slot_act, obs_action);   // (pre_rows: the slot takes the NEW episode's record 0, from the partner)
    FWG_TL(A, 14);
    // row-log mode: the lagged rows stay where they are; only lanes that need the COMPLETE record in registers (episode
    // end: terminal observation) read them back, and only early-episode / failed lanes compute rows of their own
    const bool early = c.obs_length > 1 && (int)E.steps <= (c.obs_length - 1) * c.obs_step;
    const unsigned log_pad_t = E.steps;   // rows with lag >= this are padding (valid for lanes that do not finish)
    const long long log_win = A.log_win;   // wave-uniform
    // (tail_rows: the partner writes the padding rows of lanes that neither fail nor finish; the others build them here)
    if (c.obs_length > 1 && (!ok || (early && (!tail_rows || done)))) fix_lagged_rows(c, A, e, E, T, ob, ok, pre_early && !reload, early_noise);
    FWG_TL(A, 15);
    if (c.obs_noise) add_obs_noise(c, A, e, E, ob);

    FWG_TL(A, 7);
    // ---- phase E: episode end -- metrics block, success reduction, terminal observation, auto-reset
    const unsigned long long done_mask = __ballot(done && valid);
    // waves in which no episode ends have their final observation records here: the moments for an attached rollout head
    // go out before the remaining stores, whose issue then hides the round trip of the atomics
    if (A.acc != nullptr && done_mask == 0ull) step_moments(c, A, ob, reward, done, valid, lane, e, sub, ret_prev, hs);
    // hand-shake B: the output staging area aliases the hand-off areas (first mark, raised right after barrier A), and an env
    // that is re-initialised HERE has rows the physics wave wrote in this launch (second mark: they are in memory).  A
    // foreseen end whose new episode the partner installs needs neither acknowledgement
    if (SPLIT) {
        const bool overwrites = __ballot(done && valid && !pre_rows) != 0ull;
        if (overwrites || c.obs_log == 0 || done_mask != 0ull) FWG_FLAG_WAIT(mark, overwrites ? 2 : 1);
    }
    if (done_mask != 0ull) {
        if (done && valid) {
            if (c.metrics) {
                // the episode's accumulators go into the env's finished-episode record; metrics and success sums are computed
                // from it by k_finish.  end_sum: the sum of the last <= 50 errors = S_last - S_(last-50) over the cumulative
                // sums; S_last is E.esum (this step's record when the step was valid, the previous one otherwise) and the
                // record 50 before it sits in the slot after the last written one of the 51-slot ring
                FinRec R;
                R.steps = E.steps; R.n_rec = ok ? E.steps + 1u : E.steps;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    R.e0[k] = E.e0[k]; R.esum[k] = E.esum[k]; R.eabs[k] = E.eabs[k]; R.emin[k] = E.emin[k]; R.emax[k] = E.emax[k];
                    R.rise[k] = E.rise[k];
                }
                R.settle[0] = E.settle[0]; R.settle[1] = E.settle[1]; R.gcnt[0] = E.gcnt[0]; R.gcnt[1] = E.gcnt[1]; R.sdcmd = E.sdcmd;
                {   // sum of the last <= 50 errors, exactly: S_last - S_(last - 50) in fixed point (the old slot was requested
                    // before the integration for a foreseen end, before the bookkeeping stores otherwise)
                    const Fix3 old = fix3_unpack(pre_old);
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        R.end_sum[k] = fix_to_float(fix_wrap(S_prev.s[k] - (R.n_rec > (unsigned)FWG_END_WINDOW ? old.s[k] : 0ll)));
                }
                if (E.flags & FWG_FLAG_FIN_PENDING) fin_collect_pending(c, A, e);   // (rare) never collected: fold it now
                fin_store(c, A.S, A.N, e, R);
                E.flags |= FWG_FLAG_FIN_PENDING;
                if (!c.auto_reset) GROUP(A.S, A.N, (L.gym >> 2) + 1, e) = make_float4(u2f(E.flags), u2f(E.wcnt), u2f(E.gcnt[0]), u2f(E.gcnt[1]));
            } else {
                atomicAdd(A.reduce, 1ull);   // episodes finished
            }
        }
        FWG_TL(A, 11);
        if (A.term_obs != nullptr) {
            // (partner_rows: of an end whose new episode the partner installs only the newest record -- the partner copies the rest)
            const bool newest_only = partner_rows && done && valid && pre_rows;
            const unsigned long long part_mask = partner_rows ? __ballot(newest_only) : 0ull;
            const unsigned long long full_mask = done_mask & ~part_mask;
            if (part_mask != 0ull) {
                if (newest_only) {
                    float* o = A.term_obs + e * c.obs_dim;
                    if (((c.obs_dim | c.n_obs) & 3) == 0) {
#pragma unroll
                        for (int q = 0; q < FWG_MAX_OBS / 4; ++q)
                            if (q * 4 < c.n_obs) reinterpret_cast<float4*>(o)[q] = make_float4(ob.get(4 * q), ob.get(4 * q + 1), ob.get(4 * q + 2), ob.get(4 * q + 3));
                    } else {
#pragma unroll
                        for (int j = 0; j < FWG_MAX_OBS; ++j)
                            if (j < c.n_obs) o[j] = ob.get(j);
                    }
                }
            }
            // a few ending lanes: each stores its own record; many (synchronised episodes): staged, coalesced
            if (full_mask != 0ull) {
                if ((c.obs_dim & 3) == 0 && __popcll(full_mask) <= 8) {
                    if (done && valid && !newest_only) {
                        float4* o4 = reinterpret_cast<float4*>(A.term_obs + e * c.obs_dim);
#pragma unroll
                        for (int q = 0; q < (FWG_MAX_OBS * FWG_MAX_ROWS) / 4; ++q)
                            if (q * 4 < c.obs_dim) o4[q] = make_float4(ob.get(4 * q), ob.get(4 * q + 1), ob.get(4 * q + 2), ob.get(4 * q + 3));
                    }
                } else {
                    write_obs<ROLE>(c, A.term_obs, env0, A.N, ob, lds + M.stage, lane, full_mask);
                }
            }
        }
        FWG_TL(A, 12);
        FWG_TL(A, 13);
        if (c.auto_reset && done && valid) {
            bool ready = draw_stage_of(E.flags) == FWG_DRAW_READY;
            // prepared in the steps after the previous reset; valid for this configuration generation / episode?  (tag and draw
            // are in registers either way: requested before the integration or before the bookkeeping stores)
            if (ready) {
                ready = pre_draw && f2u(pre_tag.x) == dc.generation && f2u(pre_tag.y) == E.episode + 1u;
                RD.flags = f2u(pre_tag.z);
            }
            RD.episode = E.episode + 1u;
            if (!ready) reset_sample(c, dc, A, e, E.episode, E.flags, T, RD);   // episode ended before its successor's draw was complete
            reset_finish<TURB>(c, A, e, E, T, ob, c.use_cmd_ring ? cring : aring, A.slot_end, A.slot_lag, A.bit_goal, RD, true, pre_rows);
            if (!pre_rows) store_sim<TURB>(c, A.S, A.N, e, E);   // (pre_rows: the partner wrote the new simulator rows)
            store_gym(c, A.S, A.N, e, E, A.bit_goal, true, true);
        }
    }

    if (A.acc != nullptr && done_mask != 0ull) step_moments(c, A, ob, reward, done, valid, lane, e, sub, ret_prev, hs);
    FWG_TL(A, 8);
    // ---- phase F: outputs and the state write-back
#ifndef FWG_ABL_NO_OBSWRITE
    if (c.obs_log == 0) {
        write_obs<ROLE>(c, A.obs, env0, A.N, ob, lds + M.stage, lane, ~__ballot(pre_rows), A.acc != nullptr);   // (pre_rows: the partner wrote the new window)
    } else {
        if (valid && !pre_rows) log_store_row(c, A.obs, A.N, e, log_win, 0, ob);   // the new record: 64 consecutive rows per wave
        if (__ballot((done || (early && !tail_rows)) && valid && !pre_rows) != 0ull) {
#pragma unroll
            for (int r = 1; r < FWG_MAX_ROWS; ++r)
                if (r < c.obs_length && valid && !pre_rows && (done || (early && !tail_rows && r * c.obs_step >= (int)log_pad_t)))
                    log_store_row(c, A.obs, A.N, e, log_win, r, ob);
        }
    }
#else
    if (ob.get(0) == 1.2345e30f) A.obs[e] = ob.get(1) + ob.get(c.obs_dim - 1);
#endif
    FWG_TL(A, 9);
#ifdef FWG_TIMELINE
    if (A.trace != nullptr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); FWG_TL(A, 10); }
#endif
}

template <bool TURB, int SPEC>
__global__ __launch_bounds__(FWG_WAVE, 1) void k_step(const DevCfg* __restrict__ cp, const DynCfg* __restrict__ dp, const KArgs A0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    step_wave<TURB, SPEC, 0>(cp, dp, A0, lds, (int)blockIdx.x);
}

// the same step on two waves per 64 environments (specialised configurations only: the generic kernel keeps its tables
// in lane-private LDS columns of ONE wave)
template <bool TURB, int SPEC>
__global__ __launch_bounds__(2 * FWG_WAVE, 2) void k_step2(const DevCfg* __restrict__ cp, const DynCfg* __restrict__ dp, const KArgs A0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (threadIdx.x < FWG_WAVE) step_wave<TURB, SPEC, 1>(cp, dp, A0, lds, (int)blockIdx.x);
    else step_wave<TURB, SPEC, 2>(cp, dp, A0, lds, (int)blockIdx.x);
}

// One rollout step in ONE launch (fwg_rollout_step): the head (VecNormalize statistics + MlpPolicy + sampling, actor_block of
// fwgym_actor.h) for the 256 environments of the workgroup, then the env step of those same environments under the actions
// just sampled -- examples/train_rl_controller.py:223-231 (VecNormalize(SubprocVecEnv).step inside PPO2's runner) without the
// launch boundary between policy and env.  The only grid-wide dependency of a rollout step -- the batch moments every
// env step contributes to VecNormalize's running statistics -- falls on the boundary BETWEEN launches: a launch's head phase
// folds the moments its predecessor's step phase left (accumulator set c % 3), its step phase adds into set (c + 1) % 3.
// Workgroup = 8 waves: all of them run the head (one 32-environment tile each), then waves 0..3 are the physics waves and
// waves 4..7 the gym waves of four two-wave steps (k_step2's roles; a workgroup's waves are dealt to the four SIMDs in turn,
// so every SIMD hosts one wave of each kind).  LDS: the packed weights during the head phase, the four groups' step areas
// afterwards (aliased), then what outlives the head: updated statistics, the sampled actions.
#define FWG_RO_GROUPS 4
#define FWG_RO_ENVS (FWG_RO_GROUPS * FWG_WAVE)
__host__ __device__ inline int rollout_shared_floats(const DevCfg& c, int hsplit) {
    const int w = actor_weight_floats((c.obs_dim + 15) / 16, hsplit > 1 ? 2 : 1);
    const int st = FWG_RO_GROUPS * lds_map(c.obs_dim, c.n_obs, c.L.window, c.use_cmd_ring, false, c.obs_log, true).total;
    return w > st ? w : st;
}
__host__ __device__ inline int rollout_lds_floats(const DevCfg& c, int hsplit) {
    return rollout_shared_floats(c, hsplit) + actor_scratch_floats() + FWG_RO_ENVS * 4;
}
// (configurations the fused launch exists for: SpecCfg<SPEC>::rollout_ok -- dense observation batch of at most 64 entries, the
// head's limit -- and everything within 160 KiB of LDS, checked by the launcher)
template <bool TURB, int SPEC, int HSPLIT>
__global__ __launch_bounds__(2 * FWG_RO_ENVS, 2) void k_rollout(const DevCfg* __restrict__ cp, const DynCfg* __restrict__ dp, const KArgs A0,
                                                                 const ActorArgs AA) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const DevCfg& c = SpecCfg<SPEC>::get(cp);
    constexpr int NK1 = (SpecCfg<SPEC>::obs_dim + 15) / 16;
    const int shared = rollout_shared_floats(c, HSPLIT);
    ActorLds Z;
    Z.F = reinterpret_cast<frag_t*>(lds);
    Z.mean_s = lds + shared;
    Z.rstd_s = Z.mean_s + FWG_ACT_MAX_OBS;
    Z.misc = Z.rstd_s + FWG_ACT_MAX_OBS;
    Z.bias = Z.misc + 4 + 2 * FWG_ACT_MAX_OBS + 4;
    Z.act_out = lds + shared + actor_scratch_floats();
    const unsigned act_ctr = actor_block<HSPLIT, NK1>(AA, Z, (long)blockIdx.x * FWG_RO_ENVS);
    // actions and statistics are in LDS; the weights are dead: their area becomes the step areas.  (LDS-only barrier: a
    // __syncthreads() would also sit through the acknowledgement of the head's output stores, 2-3k ticks)
    FWG_BLOCK_SYNC_LDS();
    const int wave = threadIdx.x >> 6, group = wave & (FWG_RO_GROUPS - 1);
    const int sub = (int)blockIdx.x * FWG_RO_GROUPS + group;
    const HeadStats hs{Z.mean_s, Z.misc[1], act_ctr + 1u};
    float* area = lds + group * lds_map(c.obs_dim, c.n_obs, c.L.window, c.use_cmd_ring, false, c.obs_log, true).total;
    const float* acts = Z.act_out + group * (FWG_WAVE * 4);
    if (wave < FWG_RO_GROUPS) step_wave<TURB, SPEC, 1, HeadStats>(cp, dp, A0, area, sub, acts, hs);
    else step_wave<TURB, SPEC, 2, HeadStats>(cp, dp, A0, area, sub, acts, hs);
}

// ---------------------------------------------------------------------------------------------------------------------
// reset kernel
// ---------------------------------------------------------------------------------------------------------------------
template <bool TURB, int SPEC>
__global__ __launch_bounds__(FWG_WAVE) void k_reset(const DevCfg* __restrict__ cp, const DynCfg* __restrict__ dp, const KArgs A0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef KernelTypes<SPEC> KT;
    const DevCfg& c = SpecCfg<SPEC>::get(cp);
    const KArgs A = resolve_slots(c, A0);
    DynCfgK& dc = *(DynCfgK*)dp;
    const int lane = threadIdx.x;
    const long env0 = (long)blockIdx.x * FWG_WAVE;
    const bool valid = env0 + lane < A.N;
    const long e = valid ? env0 + lane : A.N - 1;
    const LdsMap M = lds_map(c.obs_dim, c.n_obs, c.L.window, c.use_cmd_ring, KT::generic, c.obs_log);
    const bool sel = valid && (A.mask == nullptr || A.mask[e] != 0);
    const unsigned long long sel_mask = __ballot(sel);
    if (sel_mask == 0ull) return;
    typename KT::Tab T = KT::tab(lds, M, lane);
    typename KT::Obs ob = KT::obs(lds, M, lane);
    Env E;
    {   // only what survives a reset is needed: the sticky goal flag and the episode counter
        const float4 w = CGROUP(A.S, A.N, (c.L.cold >> 2), e), f = CGROUP(A.S, A.N, (c.L.gym >> 2) + 1, e);
        E.episode = f2u(w.w); E.flags = f2u(f.x);
    }
    if (c.int_window && sel) {   // integrator entries of the reset observation: the old episode's sums (see reset_finish)
        const int g0 = c.L.gym >> 2;
        const float4 q0 = CGROUP(A.S, A.N, g0 + 0, e), q2 = CGROUP(A.S, A.N, g0 + 2, e), q3 = CGROUP(A.S, A.N, g0 + 3, e),
                     q4 = CGROUP(A.S, A.N, g0 + 4, e), q7 = CGROUP(A.S, A.N, g0 + 7, e);
        const bool last_failed = (E.flags & FWG_FLAG_LAST_FAILED) != 0u;
        const int steps = (int)(f2u(q0.w) & 0xFFFFu), W = c.int_window;
        const int n = last_failed ? steps : steps + 1;                  // records in the old histories
        const float perr[3] = {q2.w, q3.w, q4.y}, e0[3] = {q7.y, q7.z, q7.w};
        // S_(n-1): the cumulative sums (fixed point, fwgym_env.h Fix3) through the last record, which sits one slot earlier after
        // a failed step; S_(n-2) = S_(n-1) - that record's error; S_(n-W-2): W + 1 records before the last one
        int last = A.slot_end - (last_failed ? 1 : 0); last += (last < 0) ? FWG_END_RING : 0;
        const Fix3 Sl = fix3_unpack(CGROUP(A.S, A.N, (c.L.end_ring >> 2) + last, e));
        Fix3 o = {{0ll, 0ll, 0ll}};
        if (n - W - 2 >= 0) {
            int slot = A.slot_end - (last_failed ? 1 : 0) - (W + 1); slot += (slot < 0) ? FWG_END_RING : 0; slot += (slot < 0) ? FWG_END_RING : 0;
            o = fix3_unpack(CGROUP(A.S, A.N, (c.L.end_ring >> 2) + slot, e));
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) E.int_reset[k] = fix_to_float(fix_wrap(Sl.s[k] - fix_quant(perr[k]) - o.s[k])) + (float)(W + 1) * e0[k];
        (void)q2;
    }
    if (sel) reset_env<TURB>(c, dc, A, e, E, T, ob, lds + M.aring + lane * 4, A.slot_end, A.slot_lag, A.bit_goal);
    if (c.obs_log == 0) {
        write_obs<0>(c, A.obs, env0, A.N, ob, lds + M.stage, lane, sel_mask);
    } else if (sel) {
        const long long win = A.log_win;
#pragma unroll
        for (int r = 0; r < FWG_MAX_ROWS; ++r)
            if (r < c.obs_length) log_store_row(c, A.obs, A.N, e, win, r, ob);
    }
    if (sel) {
        store_sim<TURB>(c, A.S, A.N, e, E);
        store_gym(c, A.S, A.N, e, E, A.bit_goal, true, true);
        if (A.tgt_out != nullptr) {
#pragma unroll
            for (int k = 0; k < FWG_MAX_TARGETS; ++k)
                if (k < c.n_targets) A.tgt_out[e * c.n_targets + k] = E.tgt[k];
        }
    }
}

// simulator.model (fixed_wing.py:532-559): the aircraft parameter table of the NEXT episode of every env whose prepared set
// is stale -- listed parameters drawn around their nominal values (Philox: counter = env id, episode, index in the list),
// the force / moment constants derived from the table by the formulas of lower_config -- into L.aero_next, tagged with
// the episode it is for and the configuration generation.  Launched before every kernel that may reset an env (fwg_step,
// fwg_reset); lanes whose set is current leave at once.  Rare path: loops, the table in a lane-private LDS column.
// (the inertia entries read from the nominal table: PyFly builds its inertia matrix and the gammas once, from the parameter
// file -- the reference's write of a sampled Jx into simulator.params is reported by get_simulator_parameters and moves nothing)
struct LaneColumn {
    const float* base;
    int stride;
    const float* nominal;
    __device__ __forceinline__ float operator[](int i) const {
        return (i == FWG_P_JX || i == FWG_P_JY || i == FWG_P_JZ || i == FWG_P_JXZ) ? nominal[i] : base[i * stride];
    }
};
// the next episode's parameter set of ONE env (tab: this thread's column of the LDS table [FWG_N_PARAMS][stride])
__device__ __forceinline__ void model_draw_env(const DevCfg& c, const DynCfg* dp, const KArgs& A, long e, float* P, int stride, bool model_part = true) {
    const ModelCfg& m = dp->model;
    const unsigned episode_new = f2u(CGROUP(A.S, A.N, (c.L.cold >> 2), e).w) + 1u;
    const unsigned env_id = (unsigned)(A.env_base + e);
    if (c.randomize_scaling) {   // reward.randomize_scaling (fixed_wing.py:330-334): 1 / U(low, high) per listed factor
        const float4 tag = CGROUP(A.S, A.N, (c.L.fscale_next >> 2) + FWG_MAX_FACTORS / 4, e);
        if (!(f2u(tag.x) == episode_new && f2u(tag.y) == dp->generation)) {
            for (int g = 0; g < FWG_MAX_FACTORS / 4; ++g) {
                float v[4];
                for (int i = 0; i < 4; ++i) {
                    const int f = 4 * g + i;
                    v[i] = f < c.n_factors ? V(c).factor[f].inv_scaling : 1.f;
                    if (f < c.n_factors && dp->fs_hi[f] > dp->fs_lo[f]) {
                        const u4 b = philox4x32(env_id, episode_new, (unsigned)f, FWG_STREAM_REWARD_SCALE, A.seed_lo, A.seed_hi);
                        v[i] = 1.f / (dp->fs_lo[f] + (dp->fs_hi[f] - dp->fs_lo[f]) * u01(b.x));
                    }
                }
                GROUP(A.S, A.N, (c.L.fscale_next >> 2) + g, e) = make_float4(v[0], v[1], v[2], v[3]);
            }
            GROUP(A.S, A.N, (c.L.fscale_next >> 2) + FWG_MAX_FACTORS / 4, e) = make_float4(u2f(episode_new), u2f(dp->generation), 0.f, 0.f);
        }
    }
    if (c.model_n <= 0 || !model_part) return;
    const float4 tag = CGROUP(A.S, A.N, (c.L.aero_next >> 2) + FWG_AERO_GROUPS - 1, e);
    if (f2u(tag.y) == episode_new && f2u(tag.z) == dp->generation) return;
    for (int i = 0; i < FWG_N_PARAMS; ++i) P[i * stride] = m.nominal[i];
    for (int i = 0; i < m.n; ++i) {
        const u4 b = philox4x32(env_id, episode_new, (unsigned)i, FWG_STREAM_MODEL, A.seed_lo, A.seed_hi);
        const float nominal = m.nominal[m.idx[i]];
        float x;
        if (m.dist == 0) {   // np_random.normal(loc, scale) then np.clip (numpy's order: max with the lower end first)
            const float z = sqrtf(-2.f * logf(u01(b.x))) * cosf(6.2831853071795865f * u01(b.y));
            x = fminf(fmaxf(nominal + m.var[i] * z, m.lo[i]), m.hi[i]);
        } else {
            x = (nominal - m.var[i]) + 2.f * m.var[i] * u01(b.x);
        }
        P[m.idx[i] * stride] = x;
    }
    for (int g = 0; 4 * g < m.n; ++g) {   // the sampled values themselves, in list order (get_simulator_parameters)
        float r[4];
        for (int i = 0; i < 4; ++i) r[i] = 4 * g + i < m.n ? P[m.idx[4 * g + i] * stride] : 0.f;
        GROUP(A.S, A.N, (c.L.model_raw_next >> 2) + g, e) = make_float4(r[0], r[1], r[2], r[3]);
    }
    Aero a;
    derive_aero<float>(LaneColumn{P, stride, m.nominal}, m.rho, m.g, a);
    float v[4 * FWG_AERO_GROUPS];
    int k = 0;
#define FWG_AERO_PUT(n) v[k++] = a.n;
    FWG_AERO_LIST(FWG_AERO_PUT)
#undef FWG_AERO_PUT
    v[FWG_N_AERO] = u2f(episode_new); v[FWG_N_AERO + 1] = u2f(dp->generation); v[FWG_N_AERO + 2] = 0.f;
#pragma unroll
    for (int g = 0; g < FWG_AERO_GROUPS; ++g)
        GROUP(A.S, A.N, (c.L.aero_next >> 2) + g, e) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}
// every env (fwg_reset, and the first fwg_step after the configuration generation changed): one lane per env
__global__ __launch_bounds__(FWG_WAVE) void k_model_draw(const DevCfg* __restrict__ cp, const DynCfg* __restrict__ dp, const KArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [FWG_N_PARAMS][64]
    const long e = (long)blockIdx.x * FWG_WAVE + threadIdx.x;
    if (e >= A.N) return;
    model_draw_env(*cp, dp, A, e, lds + threadIdx.x, FWG_WAVE);
}
// the envs that were reset by the previous launch: a few workgroups work off their queue (count | env indices), ONE WAVE per
// env -- the listed parameters are sampled by as many lanes side by side (a lane per parameter: one Philox block each), the 49
// constants derived by one lane from the wave's LDS table, the 13 groups stored by 13 lanes -- so that a step of a long run
// (a few dozen resets) pays ~350 dependent instructions instead of the ~2 500 of a lane that does it all by itself, and
// neither pays the full-grid launch whose 65 536 lanes almost all leave after two loads (5-8 us per step).  The queue it
// consumed is emptied by the NEXT launch's instance (which consumes the other one): no grid-wide hand-shake needed.
#define FWG_MQ_WAVES 4
#define FWG_MQ_BLOCKS 32
__global__ __launch_bounds__(64 * FWG_MQ_WAVES) void k_model_draw_q(const DevCfg* __restrict__ cp, const DynCfg* __restrict__ dp, const KArgs A,
                                                                     const unsigned* __restrict__ q, unsigned* __restrict__ q_other) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [FWG_MQ_WAVES][FWG_N_PARAMS + 4 FWG_AERO_GROUPS]
    const DevCfg& c = *cp;
    const ModelCfg& m = dp->model;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) q_other[0] = 0u;   // the queue this launch's step kernel appends to
    float* P = lds + wv * (FWG_N_PARAMS + 4 * FWG_AERO_GROUPS);
    float* V = P + FWG_N_PARAMS;
    const unsigned n = q[0];
    for (unsigned i = blockIdx.x * FWG_MQ_WAVES + wv; i < n; i += FWG_MQ_BLOCKS * FWG_MQ_WAVES) {
        const long e = (long)q[1u + i];
        if (c.randomize_scaling && lane == 0) model_draw_env(c, dp, A, e, P, 1, false);   // (a handful of draws, one lane)
        if (c.model_n <= 0) continue;
        const unsigned episode_new = f2u(CGROUP(A.S, A.N, (c.L.cold >> 2), e).w) + 1u;
        const unsigned env_id = (unsigned)(A.env_base + e);
        const float4 tag = CGROUP(A.S, A.N, (c.L.aero_next >> 2) + FWG_AERO_GROUPS - 1, e);
        if (f2u(tag.y) == episode_new && f2u(tag.z) == dp->generation) continue;   // (wave-uniform)
        FWG_WAVE_SYNC();
        if (lane < FWG_N_PARAMS) P[lane] = m.nominal[lane];
        FWG_WAVE_SYNC();
        if (lane < m.n) {
            const u4 b = philox4x32(env_id, episode_new, (unsigned)lane, FWG_STREAM_MODEL, A.seed_lo, A.seed_hi);
            const float nominal = m.nominal[m.idx[lane]];
            float x;
            if (m.dist == 0) {
                const float z = sqrtf(-2.f * logf(u01(b.x))) * cosf(6.2831853071795865f * u01(b.y));
                x = fminf(fmaxf(nominal + m.var[lane] * z, m.lo[lane]), m.hi[lane]);
            } else {
                x = (nominal - m.var[lane]) + 2.f * m.var[lane] * u01(b.x);
            }
            P[m.idx[lane]] = x;
        }
        FWG_WAVE_SYNC();
        if (4 * lane < m.n) {   // the sampled values themselves, in list order (get_simulator_parameters)
            float r[4];
            for (int k = 0; k < 4; ++k) r[k] = 4 * lane + k < m.n ? P[m.idx[4 * lane + k]] : 0.f;
            GROUP(A.S, A.N, (c.L.model_raw_next >> 2) + lane, e) = make_float4(r[0], r[1], r[2], r[3]);
        }
        if (lane == 0) {
            Aero a;
            derive_aero<float>(LaneColumn{P, 1, m.nominal}, m.rho, m.g, a);
            int k = 0;
#define FWG_AERO_PUT(nm) V[k++] = a.nm;
            FWG_AERO_LIST(FWG_AERO_PUT)
#undef FWG_AERO_PUT
            V[FWG_N_AERO] = u2f(episode_new); V[FWG_N_AERO + 1] = u2f(dp->generation); V[FWG_N_AERO + 2] = 0.f;
        }
        FWG_WAVE_SYNC();
        if (lane < FWG_AERO_GROUPS)
            GROUP(A.S, A.N, (c.L.aero_next >> 2) + lane, e) = make_float4(V[4 * lane], V[4 * lane + 1], V[4 * lane + 2], V[4 * lane + 3]);
    }
}

__global__ void k_check_nan(const float* __restrict__ a, long n, int* flag) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != a[i]) atomicOr(flag, 1);
}

// fwg_finish_episodes: finished-episode records not yet collected -> metrics block + success sums (per-wave reduction by
// shuffles, one atomic per value per wave: the per-GPU part of the success reduction of examples/train_rl_controller.py:51-66,
// 80-85); the pending mark is cleared.  One lane per env; a wave without a pending record leaves after one 16-byte load.
__device__ __forceinline__ unsigned long long finish_wave(const DevCfg& c, const KArgs& A, long e, int lane, bool pending, float4 fl) {
    float red[FWG_N_REDUCE];
#pragma unroll
    for (int i = 0; i < FWG_N_REDUCE; ++i) red[i] = 0.f;
    if (pending) {
        FinRec R;
        fin_load(c, A.S, A.N, e, R);
        float mt[FWG_N_METRICS];
        finish_metrics(c, R, mt, red);
        if (A.metrics != nullptr) {
#pragma unroll
            for (int i = 0; i < FWG_N_METRICS; ++i) A.metrics[(unsigned)i * (unsigned)A.N + (unsigned)e] = mt[i];
        }
        GROUP(A.S, A.N, (c.L.gym >> 2) + 1, e) = make_float4(u2f(f2u(fl.x) & ~FWG_FLAG_FIN_PENDING), fl.y, fl.z, fl.w);
    }
    float v32[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v32[i] = i < FWG_N_REDUCE ? red[i] : 0.f;
    const float tot = wave_totals32(v32, lane);   // lane l holds the total of value l & 31
    unsigned long long old = 0ull;   // (returned: the caller's ticket waits for the atomics to have been performed)
    if (lane < FWG_N_REDUCE && tot != 0.f) old = atomicAdd(A.reduce + lane, reduce_fixed(lane, tot));
    return old;
}
// TAKE (fwg_reduce_success_device, FWGYM_TAKE=ticket): the same launch also hands the sums out and clears them -- the block that
// finishes LAST (a device-wide ticket, A.reduce[FWG_N_REDUCE]) exchanges the 16 accumulators for zero and writes the floats.
// Ordering without a fence: a block's accumulating atomics RETURN their old values and the ticket is taken only after they
// have come back (performed at the device's coherence point), so the last ticket holder's exchanges see every contribution.
// (First form of round 6: __threadfence() before the ticket -- on this part an agent-scope release writes the XCD's L2 back:
// 21.6 us per launch under rocprofv3 against ~4.5 + ~2 for the two launches it replaced; profiles/r06_finish_take.txt.)
template <bool TAKE>
__global__ __launch_bounds__(FWG_WAVE) void k_finish(const DevCfg* __restrict__ cp, const KArgs A, float* __restrict__ take_out) {
    const DevCfg& c = *cp;
    const int lane = threadIdx.x;
    const long e0 = (long)blockIdx.x * FWG_WAVE + lane;
    const bool valid = e0 < A.N;
    const long e = valid ? e0 : A.N - 1;
    const float4 fl = CGROUP(A.S, A.N, (c.L.gym >> 2) + 1, e);
    const bool pending = valid && (f2u(fl.x) & FWG_FLAG_FIN_PENDING);
    unsigned long long seen = 0ull;
    if (__ballot(pending) != 0ull) seen = finish_wave(c, A, e, lane, pending, fl);
    if (TAKE) {
        FWG_TOUCH((unsigned)seen);   // (the returned values in registers: the wave's accumulating atomics have been performed)
        FWG_TOUCH((unsigned)(seen >> 32));
        unsigned ticket = 0u;
        if (lane == 0) ticket = atomicAdd(reinterpret_cast<unsigned*>(A.reduce + FWG_N_REDUCE), 1u);
        const bool last = __ballot(lane == 0 && ticket == gridDim.x - 1u) != 0ull;
        if (last) {
            if (lane < FWG_N_REDUCE) {
                const long long q = (long long)atomicExch(A.reduce + lane, 0ull);
                take_out[lane] = lane < 5 ? (float)q : (float)((double)q / (double)FWG_ACC_SCALE);
            }
            if (lane == 0) *reinterpret_cast<unsigned*>(A.reduce + FWG_N_REDUCE) = 0u;   // (the next launch is stream-ordered behind this one)
        }
    }
}
// the two-launch form of fwg_reduce_success_device (default): fixed-point sums -> floats, accumulators cleared
__global__ void k_reduce_take(unsigned long long* __restrict__ acc, float* __restrict__ out) {
    const int i = threadIdx.x;
    if (i < FWG_N_REDUCE) {
        const long long q = (long long)acc[i];
        out[i] = i < 5 ? (float)q : (float)((double)q / (double)FWG_ACC_SCALE);
        acc[i] = 0ull;
    }
}

// fwg_gae: generalised advantage estimation over a rollout stored step-major ([T][N], as fwg_rollout_step / fwg_actor_act
// fill it) -- the backward loop of PPO2's runner behind examples/train_rl_controller.py:231-232 (`PPO2(...).learn`):
//   delta_t = r_t + gamma V_(t+1) (1 - done_t) - V_t,  A_t = delta_t + gamma lambda (1 - done_t) A_(t+1),  R_t = A_t + V_t
// with done_t the flag the env returned for step t (the observation behind V_(t+1) then belongs to the next episode) and V_T
// the value of the observation after the last step.
// HBM-bound: 17 B per stored transition (4 + 4 + 1 read, 4 + 4 written).  The recurrence is AFFINE in A_(t+1), so it splits over
// time: a workgroup of four waves takes 64 envs x up to 128 steps, wave w the steps [w L, (w + 1) L), L = 32.  Every lane requests
// its 32 x 3 values at once (nothing in the first pass depends on another load: 4 096 waves x 96 loads in flight at 65 536 x 128),
// scans its segment backwards from A = 0 keeping per step the partial advantage A0_t and the coefficient product a_t
// (A_t = A0_t + a_t B, B = the advantage at the first step AFTER the segment), publishes (A0, a) of its first step through LDS,
// folds the later segments' summaries into its B after ONE barrier, and writes A and A + V from registers -- every byte is read
// once.  Rollouts longer than 128 steps are walked in 128-step spans from the end, the boundary carried in a register.
// (first build of round 6: one lane per env walking all T steps with 8 steps of loads in flight -- 48 us for 65 536 x 128,
// 0.37 of the roofline: 1 024 waves cannot keep 8 TB/s busy)
#define FWG_GAE_L 32
__global__ __launch_bounds__(256) void k_gae(const float* __restrict__ rew, const float* __restrict__ val, const unsigned char* __restrict__ done,
                                             const float* __restrict__ last_value, float gamma, float lam, float* __restrict__ adv,
                                             float* __restrict__ ret, long T, long N) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [4 waves][64 lanes][2]: (A0, a) of each segment's first step
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long e0 = (long)blockIdx.x * 64 + lane;
    const bool valid = e0 < N;
    const long e = valid ? e0 : N - 1;
    float carry = 0.f;                                            // advantage at the first step of the span handled before (later in time)
    for (long hi = T; hi > 0; hi -= 4 * FWG_GAE_L) {              // span [lo, hi), walked from the end of the rollout
        const long lo = hi - 4 * FWG_GAE_L > 0 ? hi - 4 * FWG_GAE_L : 0;
        const long s0 = lo + (long)w * FWG_GAE_L;                 // this wave's segment [s0, s1)
        const long s1 = s0 + FWG_GAE_L < hi ? s0 + FWG_GAE_L : hi;
        float r[FWG_GAE_L], v[FWG_GAE_L], A0[FWG_GAE_L], ac[FWG_GAE_L];
        unsigned char d[FWG_GAE_L];
#pragma unroll
        for (int k = 0; k < FWG_GAE_L; ++k) {
            const long t = s0 + k;
            if (t < s1) { r[k] = rew[t * N + e]; v[k] = val[t * N + e]; d[k] = done[t * N + e]; }
        }
        float vnext = 0.f;
        if (s0 < s1) vnext = s1 < T ? val[s1 * N + e] : last_value[e];
        float run = 0.f, prod = 1.f;
#pragma unroll
        for (int k = FWG_GAE_L - 1; k >= 0; --k) {
            if (s0 + k < s1) {
                const float nt = d[k] ? 0.f : 1.f;
                const float delta = r[k] + gamma * vnext * nt - v[k];
                const float c = gamma * lam * nt;
                run = delta + c * run;
                prod = c * prod;
                A0[k] = run; ac[k] = prod;
                vnext = v[k];
            }
        }
        lds[(w * 64 + lane) * 2] = run;                            // (an empty segment publishes the identity: A0 = 0, a = 1)
        lds[(w * 64 + lane) * 2 + 1] = prod;
        __syncthreads();
        float B = carry, first = carry;                            // B: boundary of THIS wave's segment; first: advantage at step lo
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            if (j == w) B = first;
            first = lds[(j * 64 + lane) * 2] + lds[(j * 64 + lane) * 2 + 1] * first;
        }
        carry = first;
#pragma unroll
        for (int k = 0; k < FWG_GAE_L; ++k) {
            const long t = s0 + k;
            if (t < s1 && valid) {
                const float a = A0[k] + ac[k] * B;
                __builtin_nontemporal_store(a, adv + t * N + e);
                __builtin_nontemporal_store(a + v[k], ret + t * N + e);
            }
        }
        __syncthreads();                                           // the summaries are read before the next span overwrites them
    }
}

// known-answer hook for the device Philox4x32-10 (fwg_selftest_philox): in[i] = counter[4] | key[2]
__global__ void k_selftest_philox(const unsigned* __restrict__ in, unsigned* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u4 r = philox4x32(in[6 * i], in[6 * i + 1], in[6 * i + 2], in[6 * i + 3], in[6 * i + 4], in[6 * i + 5]);
    out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

// row-log observations -> dense [N][length * n_obs] batch (fwg_obs_gather): one thread per 16-byte piece of the output
__global__ void k_obs_gather(const float* __restrict__ log, float* __restrict__ out, const StepSlots* slots, long long win_host,
                             long N, int n_obs, int length) {
    const long long win = slots != nullptr ? slots->log_win : win_host;
    const int q_per_row = n_obs >> 2, q_per_env = q_per_row * length;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * q_per_env) return;
    const long e = i / q_per_env;
    const int q = (int)(i - e * q_per_env), r = q / q_per_row, j4 = q - r * q_per_row;
    reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(log + ((win + r) * N + e) * n_obs)[j4];
}

// =====================================================================================================================
// host side: config lowering, handle, C ABI
// =====================================================================================================================
static thread_local std::string g_err;
static int fail_with(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(x)                                                                                         \
    do {                                                                                                   \
        hipError_t _e = (x);                                                                               \
        if (_e != hipSuccess) return fail_with(FWG_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct fwg_handle {
    fwg_config cfg;
    DevCfg h;
    DynCfg hd;
    int spec;  // index of the frozen configuration the lowered DevCfg equals, or -1
    DevCfg* d_cfg;
    DynCfg* d_dyn;
    unsigned long long* d_reduce;
    int* d_flag;
    float* arena;
    int64_t n_envs;
    int64_t env_base;
    int device;
    uint64_t seed;
    int64_t gstep;  // number of env steps taken so far (drives the ring slots)
    int graph_mode; // the ring positions live on the device (d_slots[2], double-buffered by the parity of the host count)
    int64_t gstep_at_capture;
    unsigned generation_at_capture;   // configuration generation the captured launch sequences belong to
    int spec_at_capture;              // ... and the kernel INSTANCE they hold (a frozen kernel has the configuration's values folded in)
    StepSlots* d_slots;
    size_t lds_bytes;
    float* last_metrics_out;      // metrics block of the last fwg_step (what fwg_reduce_success* collect into)
    unsigned* d_mq;               // simulator.model / randomize_scaling: two reset queues [2][1 + N] (by the parity of the step count)
    int model_all_stale;          // every env needs a new prepared set (start, fwg_update_config, fwg_seed): full-grid draw next
    struct fwg_actor* observer;   // attached rollout head (fwg_attach_observer) or null
    int rollout_lds_granted;      // the one-launch rollout step's dynamic LDS has been asked for (both head variants)
    int split;                    // specialised configurations: the step runs as k_step2 (two waves per 64 envs)
#ifdef FWG_TIMELINE
    long long* trace;
#endif
};

static int compute_layout(const fwg_config& c, fwg_layout* L, std::string* why) {
    if (c.n_actions != 3) { *why = "n_actions must be 3 (elevator, aileron, throttle)"; return -1; }
    if (c.n_targets < 1 || c.n_targets > FWG_MAX_TARGETS) { *why = "n_targets out of range"; return -1; }
    if (c.n_obs < 1 || c.n_obs > FWG_MAX_OBS) { *why = "n_obs out of range"; return -1; }
    if (c.obs_length < 1 || c.obs_length > FWG_MAX_ROWS || c.obs_step < 1) { *why = "observation length/step out of range"; return -1; }
    if (c.n_factors < 0 || c.n_factors > FWG_MAX_FACTORS) { *why = "too many reward factors"; return -1; }
    if (c.streak_req > FWG_MAX_STREAK) { *why = "success_streak_req > 128"; return -1; }
    if (c.steps_max >= 65535) { *why = "steps_max must be < 65535"; return -1; }
    if (c.n_substeps < 1) { *why = "n_substeps < 1"; return -1; }
    int window = 2;
    int use_cmd = 0;
    for (int j = 0; j < c.n_obs; ++j)
        if (c.obs[j].type == FWG_OBS_ACTION) {
            if (c.obs[j].window > window) window = c.obs[j].window;
            if (!c.scale_actions) use_cmd = 1;
            if (c.obs[j].src < 0 || c.obs[j].src > 2) { *why = "action observation source"; return -1; }
        }
    for (int f = 0; f < c.n_factors; ++f)
        if (c.factor[f].cls == FWG_RC_ACTION && c.factor[f].type == FWG_RT_DELTA && c.factor[f].window > window)
            window = c.factor[f].window;
    if (window > FWG_MAX_WINDOW) { *why = "action window_size > 8"; return -1; }
    // all offsets are in 32-bit words and multiples of 4: the arena is addressed in 16-byte groups [group][env]
    int o = 0;
    L->sim = o; o += 32;                 // y[18] | dryden[8] | gust[6] (increment turbulence)
    L->cold = o; o += 8;                 // wind[3] episode | e0[3] pad  (per-episode constants, written at reset)
    L->derived = o; o += 8;              // roll pitch yaw Va | alpha beta pad pad
    L->gym = o; o += 36;                 // 9 bookkeeping groups (see load_gym)
    L->tprop = o; o += 4 * FWG_MAX_TARGETS;
    L->goal = o; o += FWG_MAX_STREAK / 8; // goal-window ring: plain word rows, 8 positions x 4 flags per word
    L->act_ring = o; o += window * 4;    // raw actions, one group per slot, slot = global_step % window
    L->cmd_ring = o; o += use_cmd ? window * 4 : 0;
    L->end_ring = o; o += FWG_END_RING * 4;
    if (c.obs_log_rows != 0) {
        if (c.obs_length < 2) { *why = "obs_log_rows needs a lagged (matrix) observation"; return -1; }
        if (c.obs_noise) { *why = "obs_log_rows: observation noise re-draws every row each step, use the dense batch"; return -1; }
        if (c.obs_log_rows < 2 * (c.obs_length - 1) || c.obs_log_rows > 4096) { *why = "obs_log_rows must be in [2 (obs_length - 1), 4096]"; return -1; }
    }
    // dense batch: ring of the last (length - 1) * step + 1 records; row log: one slot holding the episode's record 0
    L->lag_depth = c.obs_length > 1 ? (c.obs_log_rows != 0 ? 1 : (c.obs_length - 1) * c.obs_step + 1) : 0;
    L->lag_groups = (c.n_obs + 3) / 4;
    L->lag_ring = o; o += L->lag_depth * L->lag_groups * 4;
    {   // the next episode's reset draw (fwgym_env.h "draw_stage_step"): 11 groups, + 3 with dynamic target classes
        bool dynamic = false;
        for (int k = 0; k < c.n_targets; ++k) dynamic = dynamic || c.target[k].cls >= FWG_TGT_LINEAR;
        L->draw = o; o += dynamic ? 56 : 44;
    }
    if (c.model_n < 0 || c.model_n > FWG_N_PARAMS) { *why = "model_n out of range"; return -1; }
    L->aero = o; o += c.model_n > 0 ? FWG_AERO_GROUPS * 4 : 0;        // per-env force/moment constants: this episode's ...
    L->aero_next = o; o += c.model_n > 0 ? FWG_AERO_GROUPS * 4 : 0;   // ... and the next one's (k_model_draw)
    L->fscale = o; o += c.randomize_scaling ? FWG_MAX_FACTORS : 0;             // per-env 1 / scaling of the reward factors ...
    L->fscale_next = o; o += c.randomize_scaling ? FWG_MAX_FACTORS + 4 : 0;    // ... and the next episode's, + tag group
    L->model_raw = o; o += ((c.model_n + 3) / 4) * 4;                          // sampled values of the listed parameters ...
    L->model_raw_next = o; o += ((c.model_n + 3) / 4) * 4;                     // ... and the next episode's
    L->fin = o; o += c.metrics ? 28 : 0;                                       // finished-episode record (fwgym_env.h FinRec)
    L->window = window;
    L->rows = o;
    return use_cmd;
}

static float f32(double x) { return (float)x; }
// exp(t * [[a, b], [c, d]]) by scaling-and-squaring with a Taylor series (double)
static void expm2(double a, double b, double c, double d, double t, double out[4]) {
    double m[4] = {a * t, b * t, c * t, d * t};
    double nrm = fabs(m[0]) + fabs(m[1]) + fabs(m[2]) + fabs(m[3]);
    int s = 0;
    while (nrm > 0.25) { nrm *= 0.5; ++s; }
    const double sc = ldexp(1.0, -s);
    for (int i = 0; i < 4; ++i) m[i] *= sc;
    double e[4] = {1, 0, 0, 1}, term[4] = {1, 0, 0, 1};
    for (int k = 1; k < 20; ++k) {
        const double t0 = (term[0] * m[0] + term[1] * m[2]) / k, t1 = (term[0] * m[1] + term[1] * m[3]) / k;
        const double t2 = (term[2] * m[0] + term[3] * m[2]) / k, t3 = (term[2] * m[1] + term[3] * m[3]) / k;
        term[0] = t0; term[1] = t1; term[2] = t2; term[3] = t3;
        for (int i = 0; i < 4; ++i) e[i] += term[i];
    }
    for (int q = 0; q < s; ++q) {
        const double r0 = e[0] * e[0] + e[1] * e[2], r1 = e[0] * e[1] + e[1] * e[3];
        const double r2 = e[2] * e[0] + e[3] * e[2], r3 = e[2] * e[1] + e[3] * e[3];
        e[0] = r0; e[1] = r1; e[2] = r2; e[3] = r3;
    }
    for (int i = 0; i < 4; ++i) out[i] = e[i];
}
static float lim32(double x, bool is_min) {
    if (std::isnan(x)) return is_min ? -INFINITY : INFINITY;
    return (float)x;
}

static int lower_config(const fwg_config& c, DevCfg* d, DynCfg* dy, std::string* why) {
    memset(d, 0, sizeof(DevCfg));
    memset(dy, 0, sizeof(DynCfg));
    const int use_cmd = compute_layout(c, &d->L, why);
    if (use_cmd < 0) return -1;
    const double* P = c.param;
    d->dt = f32(c.dt);
    d->nsub = c.n_substeps;
    const double h = c.dt / c.n_substeps;
    d->h = f32(h); d->half_h = f32(0.5 * h); d->h_sixth = f32(h / 6.0);
    d->turbulence = c.turbulence;
    if (c.turbulence_output != FWG_TURB_FILTER && c.turbulence_output != FWG_TURB_INCREMENT) { *why = "turbulence_output"; return -1; }
    d->turb_increment = c.turbulence && c.turbulence_output == FWG_TURB_INCREMENT;
    {   // force / moment constants from the parameter table (derive_aero: the device runs the same formulas per env when
        // simulator.model re-samples the table)
        AeroT<double> ad;
        derive_aero<double>(P, c.rho, c.g, ad);
#define FWG_AERO_LOWER(n) d->n = f32(ad.n);
        FWG_AERO_LIST(FWG_AERO_LOWER)
#undef FWG_AERO_LOWER
    }
    d->model_n = c.model_n;
    dy->model.n = c.model_n; dy->model.dist = c.model_dist;
    dy->model.rho = f32(c.rho); dy->model.g = f32(c.g);
    for (int i = 0; i < FWG_N_PARAMS; ++i) dy->model.nominal[i] = f32(P[i]);
    for (int i = 0; i < c.model_n; ++i) {
        if (c.model_idx[i] < 0 || c.model_idx[i] >= FWG_N_PARAMS) { *why = "model_idx out of range"; return -1; }
        if (c.model_dist != 0 && c.model_dist != 1) { *why = "model_dist must be 0 (gaussian) or 1 (uniform)"; return -1; }
        dy->model.idx[i] = c.model_idx[i];
        dy->model.var[i] = f32(c.model_var[i]);
        dy->model.lo[i] = lim32(c.model_clip_lo[i], true); dy->model.hi[i] = lim32(c.model_clip_hi[i], false);
    }
    d->randomize_scaling = c.randomize_scaling ? 1 : 0;
    if (c.sk_n_intensity < 0 || c.sk_n_intensity > 4 || c.sk_n_turbulence < 0 || c.sk_n_turbulence > 2) { *why = "sk_n_* out of range"; return -1; }
    if (c.integration_window < 0 || c.integration_window > FWG_END_WINDOW - 1) { *why = "integration_window out of range"; return -1; }
    d->int_window = c.integration_window;
    d->has_int_obs = 0;
    for (int j = 0; j < c.n_obs; ++j) d->has_int_obs |= c.obs[j].type == FWG_OBS_TARGET_INTEGRATOR ? 1 : 0;
    if ((d->has_int_obs || d->int_window) && !c.metrics) { *why = "integration_window needs the metric accumulators (metrics = 1)"; return -1; }
    if (d->has_int_obs && d->int_window == 0) { *why = "integrator observations need integration_window > 0"; return -1; }
    d->sim_keys = (c.sk_n_intensity > 0 || c.sk_n_turbulence > 0) ? 1 : 0;
    if (d->sim_keys && !c.turbulence) { *why = "sampled turbulence keys need the turbulence kernels (turbulence = 1)"; return -1; }
    dy->sk_n_int = c.sk_n_intensity; dy->sk_n_turb = c.sk_n_turbulence; dy->sk_idx_int = c.sk_index_intensity; dy->sk_idx_turb = c.sk_index_turbulence;
    for (int i = 0; i < 4; ++i) { dy->sk_cum_int[i] = f32(c.sk_cum_intensity[i]); dy->sk_gain_int[i] = f32(c.sk_gain_intensity[i]); }
    for (int i = 0; i < 2; ++i) { dy->sk_cum_turb[i] = f32(c.sk_cum_turbulence[i]); dy->sk_on_turb[i] = f32(c.sk_on_turbulence[i]); }
    dy->sk_base_gain = f32(c.sk_base_gain);
    for (int f = 0; f < FWG_MAX_FACTORS; ++f) {
        const bool listed = c.randomize_scaling && f < c.n_factors;
        dy->fs_lo[f] = f32(listed ? c.factor_scaling_low[f] : 1.0); dy->fs_hi[f] = f32(listed ? c.factor_scaling_high[f] : 1.0);
        if (listed && !(c.factor_scaling_low[f] > 0.0 && c.factor_scaling_high[f] >= c.factor_scaling_low[f])) {
            *why = "factor_scaling_low/high must satisfy 0 < low <= high"; return -1;
        }
    }
    d->con_mask = 0;
    for (int v = 0; v < FWG_N_VARS; ++v) {
        d->con_min[v] = lim32(c.con_min[v], true); d->con_max[v] = lim32(c.con_max[v], false);
        d->val_min[v] = lim32(c.val_min[v], true); d->val_max[v] = lim32(c.val_max[v], false);
        dy->init_min[v] = std::isnan(c.init_min[v]) ? 0.f : f32(c.init_min[v]);
        dy->init_max[v] = std::isnan(c.init_max[v]) ? 0.f : f32(c.init_max[v]);
        if (std::isfinite(c.con_min[v]) || std::isfinite(c.con_max[v])) d->con_mask |= 1u << v;
    }
    if (c.actuator_microsteps < 2 * c.n_substeps || c.actuator_microsteps % (2 * c.n_substeps) != 0) {
        *why = "actuator_microsteps must be a positive multiple of 2*n_substeps"; return -1;
    }
    d->act_per_half = c.actuator_microsteps / (2 * c.n_substeps);
    const double hm = c.dt / c.actuator_microsteps;
    for (int i = 0; i < 2; ++i) {
        double phi[4];
        expm2(0.0, 1.0, -c.elevon_omega0[i] * c.elevon_omega0[i], -2.0 * c.elevon_zeta[i] * c.elevon_omega0[i], hm, phi);
        for (int j = 0; j < 4; ++j) d->act_phi[i][j] = f32(phi[j]);
        d->dot_max[i] = std::isfinite(c.elevon_dot_max[i]) ? f32(c.elevon_dot_max[i]) : INFINITY;
        d->act_travel[i] = std::isfinite(c.elevon_dot_max[i]) ? f32(c.elevon_dot_max[i] * hm) : INFINITY;
    }
    d->act_ethr = f32(exp(-hm / c.throttle_tau));
    for (int i = 0; i < FWG_N_DRYDEN * FWG_N_DRYDEN; ++i) d->dryA[i] = f32(c.dryden_A[i]);
    for (int i = 0; i < FWG_N_DRYDEN * 4; ++i) d->dryB[i] = f32(c.dryden_B[i]);
    for (int i = 0; i < 6 * FWG_N_DRYDEN; ++i) d->dryC[i] = f32(c.dryden_C[i]);

    d->steps_max = c.steps_max; d->obs_length = c.obs_length; d->obs_step = c.obs_step; d->n_obs = c.n_obs;
    d->obs_dim = c.obs_length * c.n_obs;
    d->obs_noise = c.obs_noise; d->obs_noise_mean = f32(c.obs_noise_mean); d->obs_noise_std = f32(c.obs_noise_std);
    for (int j = 0; j < c.n_obs; ++j) {
        const fwg_obs_desc& o = c.obs[j];
        if (o.type == FWG_OBS_STATE && (o.src < 0 || o.src >= FWG_N_VARS)) { *why = "observation state source"; return -1; }
        if ((o.type == FWG_OBS_TARGET_RELATIVE || o.type == FWG_OBS_TARGET_ABSOLUTE) && (o.src < 0 || o.src >= c.n_targets)) {
            *why = "observation target source"; return -1;
        }
        d->obs[j] = DevObs{o.type, o.src, o.window < 1 ? 1 : o.window, (c.obs_normalize && o.norm) ? 1 : 0, f32(o.mean), f32(1.0 / o.var)};
    }
    d->scale_actions = c.scale_actions;
    d->scale_low = f32(c.scale_low); d->scale_high = f32(c.scale_high);
    d->inv_scale_span = c.scale_actions ? f32(1.0 / (c.scale_high - c.scale_low)) : 0.f;
    for (int i = 0; i < 3; ++i) {
        d->act_to_low[i] = f32(c.act_to_low[i]); d->act_to_high[i] = f32(c.act_to_high[i]);
        d->inv_act_span[i] = f32(1.0 / (c.act_to_high[i] - c.act_to_low[i]));
        d->act_bound_min[i] = c.has_action_bounds ? f32(c.act_bound_min[i]) : -INFINITY;
        d->act_bound_max[i] = c.has_action_bounds ? f32(c.act_bound_max[i]) : INFINITY;
    }
    d->has_action_bounds = c.has_action_bounds;
    d->n_targets = c.n_targets; d->resample_every = c.resample_every; d->streak_req = c.streak_req;
    d->on_success = c.on_success; d->goal_enabled = c.streak_req > 0;
    {   // smallest count whose float64 mean over the window reaches the fraction (np.mean(...) >= fraction)
        int mc = 0;
        while (mc <= c.streak_req && !((double)mc / (double)(c.streak_req > 0 ? c.streak_req : 1) >= c.streak_fraction)) ++mc;
        d->streak_min_count = mc;
    }
    d->any_dynamic_target = 0;
    for (int k = 0; k < c.n_targets; ++k) {
        const fwg_target_desc& t = c.target[k];
        if (t.var < 0 || t.var >= FWG_N_VARS) { *why = "target variable"; return -1; }
        if (t.cls >= FWG_TGT_LINEAR) d->any_dynamic_target = 1;
        if (t.cls == FWG_TGT_COMPENSATE && t.var != FWG_V_VA) { *why = "class compensate is only defined for Va"; return -1; }
        d->target[k] = DevTarget{t.var, t.cls, t.wrap, t.has_delta, t.has_bound, f32(t.bound)};
        dy->target[k] = DynTarget{f32(t.low), f32(t.high), f32(t.delta), f32(t.slope_low), f32(t.slope_high),
                                  f32(t.amplitude_low), f32(t.amplitude_high), f32(t.period_low), f32(t.period_high)};
    }
    d->reward_potential = c.reward_potential; d->step_fail_timesteps = c.step_fail_timesteps;
    d->step_fail_value = f32(c.step_fail_value);
    for (int i = 0; i < 3; ++i) { d->term_present[i] = c.term_present[i]; d->term_weight[i] = f32(c.term_weight[i]); }
    d->n_factors = c.n_factors;
    for (int f = 0; f < c.n_factors; ++f) {
        const fwg_factor_desc& F = c.factor[f];
        if (F.fclass < 0 || F.fclass > 2 || !c.term_present[F.fclass]) { *why = "reward factor without a matching term"; return -1; }
        if (F.cls == FWG_RC_ACTION && F.type == FWG_RT_BOUND && !c.has_action_bounds) { *why = "action bound factor needs bounds_multiplier"; return -1; }
        d->factor[f] = DevFactor{F.cls, F.type, F.src, F.fclass, F.shaping, F.window < 1 ? 1 : F.window, F.has_max,
                                 F.value_is_timesteps, f32(F.sign > 0 ? 1.0 : (F.sign < 0 ? -1.0 : 0.0)), f32(1.0 / F.scaling),
                                 f32(F.max), f32(F.value)};
    }
    d->metrics = c.metrics; d->auto_reset = c.auto_reset; d->use_cmd_ring = use_cmd; d->store_derived = c.store_derived;
    d->obs_log = c.obs_log_rows;
    d->rise_low = f32(c.rise_low); d->rise_high = f32(c.rise_high);
    return 0;
}

// frozen configurations compiled into this library (host copies for matching)
#define FWG_SPEC_HOST(i) &kSpecWords##i,
static const SpecWords* const kSpecTable[] = {FWG_SPEC_LIST(FWG_SPEC_HOST) nullptr};
static int match_shape(const DevCfg& d) {
    // a shape instance whose structure is this configuration's?  (merging d's values into the instance gives d back bit for bit)
#ifndef FWG_EMU   /* (the host emulation reads V(c) as c: frozen and generic kernels only) */
#define FWG_SHAPE_MATCH(i) { const DevCfg m = merge_values(kSpec##i, &d); if (memcmp(&m, &d, sizeof(DevCfg)) == 0) return FWG_SHAPE_BASE + i; }
    FWG_SHAPE_LIST(FWG_SHAPE_MATCH)
#undef FWG_SHAPE_MATCH
#endif
    return -1;
}
static int match_spec(const DevCfg& d) {
    // FWGYM_SHAPE=0: never a shape instance (the generic kernel instead, for A/B measurements); =force: the shape instance
    // even where a frozen configuration matches (tests: the two must agree)
    const char* env = getenv("FWGYM_SHAPE");
    if (env != nullptr && env[0] == 'f') { const int s = match_shape(d); if (s >= 0) return s; }
    for (int i = 0; kSpecTable[i] != nullptr; ++i)
        if (memcmp(kSpecTable[i], &d, sizeof(DevCfg)) == 0) return i;
    if (env != nullptr && env[0] == '0') return -1;
    return match_shape(d);
}

template <bool IS_STEP>
static void launch(const fwg_handle* h, const KArgs& A, hipStream_t stream);

extern "C" {

int fwg_abi_version(void) { return FWG_ABI_VERSION; }
const char* fwg_last_error(void) { return g_err.c_str(); }

int fwg_get_layout(const fwg_config* cfg, fwg_layout* out) {
    if (!cfg || !out) return fail_with(FWG_ERR_INVALID, "null argument");
    if (cfg->abi_version != FWG_ABI_VERSION || cfg->struct_bytes != sizeof(fwg_config))
        return fail_with(FWG_ERR_ABI, "fwg_config version/size mismatch");
    std::string why;
    if (compute_layout(*cfg, out, &why) < 0) return fail_with(FWG_ERR_INVALID, why);
    return FWG_OK;
}

int fwg_create(const fwg_config* cfg, int64_t n_envs, int device, void* state_arena, int64_t env_id_base, fwg_handle** out) {
    if (!cfg || !out || !state_arena || n_envs < 1) return fail_with(FWG_ERR_INVALID, "null/invalid argument");
    if (cfg->abi_version != FWG_ABI_VERSION || cfg->struct_bytes != sizeof(fwg_config))
        return fail_with(FWG_ERR_ABI, "fwg_config version/size mismatch");
    fwg_handle* h = new fwg_handle();
    h->cfg = *cfg;
    std::string why;
    if (lower_config(*cfg, &h->h, &h->hd, &why) != 0) { delete h; return fail_with(FWG_ERR_INVALID, why); }
    h->spec = match_spec(h->h);
    h->spec_at_capture = h->spec;
    {   // FWGYM_SPLIT=0 keeps the one-wave kernel (A/B measurements)
        const char* env = getenv("FWGYM_SPLIT");
        h->split = !(env != nullptr && env[0] == '0');
    }
    h->n_envs = n_envs; h->env_base = env_id_base; h->device = device; h->seed = 0; h->gstep = 0;
    h->arena = (float*)state_arena;
    if ((int64_t)h->h.L.rows * n_envs >= (int64_t)1 << 30) { delete h; return fail_with(FWG_ERR_INVALID, "rows*n_envs must be < 2^30"); }
    if (lds_map(h->h.obs_dim, h->h.n_obs, h->h.L.window, h->h.use_cmd_ring, true).total * sizeof(float) > 64 * 1024) {
        delete h; return fail_with(FWG_ERR_INVALID, "observation too large for the LDS scratch");
    }
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc((void**)&h->d_cfg, sizeof(DevCfg)));
    HIP_TRY(hipMalloc((void**)&h->d_dyn, sizeof(DynCfg)));
    HIP_TRY(hipMemcpy(h->d_dyn, &h->hd, sizeof(DynCfg), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void**)&h->d_reduce, sizeof(unsigned long long) * (FWG_N_REDUCE + 1)));   // (+ the ticket of k_finish<true>)
    HIP_TRY(hipMalloc((void**)&h->d_flag, sizeof(int)));
    HIP_TRY(hipMalloc((void**)&h->d_slots, 2 * sizeof(StepSlots)));
    HIP_TRY(hipMemset(h->d_slots, 0, 2 * sizeof(StepSlots)));
    h->graph_mode = 0;
    HIP_TRY(hipMemcpy(h->d_cfg, &h->h, sizeof(DevCfg), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(h->d_reduce, 0, sizeof(unsigned long long) * (FWG_N_REDUCE + 1)));
    HIP_TRY(hipMemset(h->d_flag, 0, sizeof(int)));
    h->d_mq = nullptr;
    h->model_all_stale = 1;
    if (h->h.model_n > 0 || h->h.randomize_scaling) {
        HIP_TRY(hipMalloc((void**)&h->d_mq, 2 * (size_t)(1 + n_envs) * sizeof(unsigned)));
        HIP_TRY(hipMemset(h->d_mq, 0, 2 * (size_t)(1 + n_envs) * sizeof(unsigned)));
    }
    *out = h;
    return FWG_OK;
}

int fwg_destroy(fwg_handle* h) {
    if (!h) return FWG_OK;
    (void)hipSetDevice(h->device);
    (void)hipFree(h->d_cfg); (void)hipFree(h->d_dyn); (void)hipFree(h->d_reduce); (void)hipFree(h->d_flag); (void)hipFree(h->d_slots); if (h->d_mq) (void)hipFree(h->d_mq);
    delete h;
    return FWG_OK;
}

int fwg_update_config(fwg_handle* h, const fwg_config* cfg) {
    if (!h || !cfg) return fail_with(FWG_ERR_INVALID, "null argument");
    if (cfg->abi_version != FWG_ABI_VERSION || cfg->struct_bytes != sizeof(fwg_config))
        return fail_with(FWG_ERR_ABI, "fwg_config version/size mismatch");
    DevCfg d;
    DynCfg dy;
    std::string why;
    if (lower_config(*cfg, &d, &dy, &why) != 0) return fail_with(FWG_ERR_INVALID, why);
    if (memcmp(&d.L, &h->h.L, sizeof(fwg_layout)) != 0 || d.obs_dim != h->h.obs_dim)
        return fail_with(FWG_ERR_INVALID, "fwg_update_config must not change the state layout");
    dy.generation = h->hd.generation + 1u;   // ranges may have changed: prepared reset draws are stale
    h->model_all_stale = 1;
    h->cfg = *cfg; h->h = d; h->hd = dy;
    h->spec = match_spec(h->h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpy(h->d_cfg, &h->h, sizeof(DevCfg), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_dyn, &h->hd, sizeof(DynCfg), hipMemcpyHostToDevice));
    return FWG_OK;
}

int fwg_seed(fwg_handle* h, uint64_t seed) {
    if (!h) return fail_with(FWG_ERR_INVALID, "null handle");
    if (seed != h->seed) {   // prepared reset draws belong to the old streams
        h->hd.generation += 1u;
        h->model_all_stale = 1;
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipMemcpy(h->d_dyn, &h->hd, sizeof(DynCfg), hipMemcpyHostToDevice));
    }
    h->seed = seed;
    return FWG_OK;
}

int64_t fwg_global_step(const fwg_handle* h) { return h ? h->gstep : -1; }

static StepSlots host_slots(const fwg_handle* h, int64_t g) {
    const DevCfg& d = h->h;
    return make_slots(d.obs_step, d.obs_log, d.obs_length, d.L.window, d.L.lag_depth, d.streak_req, (long long)g);
}
static void fill_slots(const fwg_handle* h, int64_t g, KArgs* A) {
    const StepSlots s = host_slots(h, g);
    A->gnow = s.gnow; A->log_win = s.log_win; A->log_wrap_now = s.log_wrap_now;
    A->slot_act = s.slot_act; A->slot_end = s.slot_end; A->slot_lag = s.slot_lag; A->bit_goal = s.bit_goal;
    for (int r = 0; r < FWG_MAX_ROWS; ++r) A->lag_slots[r] = s.lag_slots[r];
}

static void base_args(const fwg_handle* h, KArgs* A) {
    memset(A, 0, sizeof(KArgs));
    A->S = h->arena; A->N = h->n_envs; A->env_base = h->env_base; A->reduce = h->d_reduce;
    A->seed_lo = (unsigned)(h->seed & 0xFFFFFFFFull); A->seed_hi = (unsigned)(h->seed >> 32);
}

static void observer_args(fwg_handle* h, KArgs* A);   // defined with the rollout head below
static int launch_rollout(fwg_handle* h, fwg_actor* a, const KArgs& A, const ActorArgs& AA, hipStream_t stream, bool probe, bool grant = false);
// simulator.model: before any launch that may reset an env, every env has the parameter set of its next episode prepared
static void launch_model_draw(fwg_handle* h, const KArgs& A, hipStream_t stream, bool all) {
    if (h->h.model_n <= 0 && !h->h.randomize_scaling) return;
    if (all || h->model_all_stale) {
        hipLaunchKernelGGL(k_model_draw, dim3((unsigned)((h->n_envs + FWG_WAVE - 1) / FWG_WAVE)), dim3(FWG_WAVE), FWG_N_PARAMS * FWG_WAVE * sizeof(float), stream, h->d_cfg, h->d_dyn, A);
        // (the queues hold nothing this draw has not covered; the one the coming kernel appends to starts empty)
        (void)hipMemsetAsync(h->d_mq, 0, sizeof(unsigned), stream);
        (void)hipMemsetAsync(h->d_mq + (size_t)(1 + h->n_envs), 0, sizeof(unsigned), stream);
        h->model_all_stale = 0;
    } else {   // the envs reset by the previous launch (queue of the other parity)
        hipLaunchKernelGGL(k_model_draw_q, dim3(FWG_MQ_BLOCKS), dim3(64 * FWG_MQ_WAVES), FWG_MQ_WAVES * (FWG_N_PARAMS + 4 * FWG_AERO_GROUPS) * sizeof(float), stream, h->d_cfg, h->d_dyn, A,
                           h->d_mq + (size_t)((h->gstep - 1) & 1) * (size_t)(1 + h->n_envs), h->d_mq + (size_t)(h->gstep & 1) * (size_t)(1 + h->n_envs));
    }
}

int fwg_reset(fwg_handle* h, const uint8_t* mask, const float* init_state, const float* init_target, float* obs_out, void* stream) {
    if (!h || !obs_out) return fail_with(FWG_ERR_INVALID, "null argument");
    KArgs A;
    base_args(h, &A);
    A.mask = mask; A.init_state = init_state; A.init_target = init_target; A.obs = obs_out;
    fill_slots(h, h->gstep - 1, &A);  // initial records take the ring position of the last completed step
    if (h->graph_mode) { A.slots_in = h->d_slots + (h->gstep & 1); A.reset_launch = 1; }
    if (h->d_mq != nullptr) A.mq = h->d_mq + (size_t)((h->gstep - 1) & 1) * (size_t)(1 + h->n_envs);   // as if part of the last step
    launch_model_draw(h, A, (hipStream_t)stream, true);
    launch<false>(h, A, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return FWG_OK;
}

int fwg_step(fwg_handle* h, const float* actions, float* obs_out, float* reward_out, uint8_t* done_out, uint8_t* term_code_out,
             float* terminal_obs_out, float* metrics_out, float* target_out, void* stream) {
    if (!h || !actions || !obs_out || !reward_out || !done_out || !term_code_out) return fail_with(FWG_ERR_INVALID, "null argument");
    KArgs A;
    base_args(h, &A);
    A.actions = actions; A.obs = obs_out; A.rew = reward_out; A.done = done_out; A.term = term_code_out;
    A.term_obs = terminal_obs_out; A.metrics = metrics_out; A.tgt_out = target_out;
    h->last_metrics_out = metrics_out;
    fill_slots(h, h->gstep, &A);
    if (h->graph_mode) { A.slots_in = h->d_slots + (h->gstep & 1); A.slots_out = h->d_slots + ((h->gstep + 1) & 1); }
    observer_args(h, &A);
#ifdef FWG_TIMELINE
    A.trace = h->trace;
#endif
    launch_model_draw(h, A, (hipStream_t)stream, false);
    if (h->d_mq != nullptr) A.mq = h->d_mq + (size_t)(h->gstep & 1) * (size_t)(1 + h->n_envs);
    launch<true>(h, A, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    h->gstep += 1;
    return FWG_OK;
}

int64_t fwg_obs_log_floats(const fwg_config* cfg, int64_t n_envs) {
    if (!cfg || cfg->obs_log_rows <= 0) return 0;
    return (int64_t)cfg->obs_step * cfg->obs_log_rows * n_envs * cfg->n_obs;
}

int fwg_obs_window(const fwg_handle* h, int64_t* plane) {
    if (!h || !plane) return fail_with(FWG_ERR_INVALID, "fwg_obs_window: null argument");
    if (h->h.obs_log <= 0) return fail_with(FWG_ERR_INVALID, "fwg_obs_window: the env writes the dense observation batch");
    *plane = host_slots(h, h->gstep - 1).log_win;   // last completed step
    return FWG_OK;
}

int fwg_obs_gather(const fwg_handle* h, const float* obs_log, float* obs_out, void* stream) {
    if (!h || !obs_log || !obs_out) return fail_with(FWG_ERR_INVALID, "fwg_obs_gather: null argument");
    if (h->h.obs_log <= 0) return fail_with(FWG_ERR_INVALID, "fwg_obs_gather: the env writes the dense observation batch");
    if (h->h.n_obs & 3) return fail_with(FWG_ERR_INVALID, "fwg_obs_gather: n_obs must be a multiple of 4");
    const long total = (long)h->n_envs * (h->h.n_obs >> 2) * h->h.obs_length;
    // graph mode: the window of the last completed step is read from the device-resident positions (replay-safe)
    const StepSlots* slots = h->graph_mode ? h->d_slots + ((h->gstep - 1) & 1) : nullptr;
    hipLaunchKernelGGL(k_obs_gather, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obs_log, obs_out, slots,
                       host_slots(h, h->gstep - 1).log_win, (long)h->n_envs, h->h.n_obs, h->h.obs_length);
    HIP_TRY(hipGetLastError());
    return FWG_OK;
}

int fwg_gae(int64_t n_steps, int64_t n_envs, const float* rewards, const float* values, const uint8_t* dones, const float* last_value,
            float gamma, float lam, float* adv_out, float* ret_out, void* stream) {
    if (!rewards || !values || !dones || !last_value || !adv_out || !ret_out) return fail_with(FWG_ERR_INVALID, "fwg_gae: null argument");
    if (n_steps < 1 || n_envs < 1) return fail_with(FWG_ERR_INVALID, "fwg_gae: n_steps and n_envs must be positive");
    hipLaunchKernelGGL(k_gae, dim3((unsigned)((n_envs + 63) / 64)), dim3(256), 4 * 64 * 2 * sizeof(float), (hipStream_t)stream, rewards, values, dones, last_value,
                       gamma, lam, adv_out, ret_out, (long)n_steps, (long)n_envs);
    HIP_TRY(hipGetLastError());
    return FWG_OK;
}

int fwg_selftest_philox(const uint32_t* ctr_key_dev, uint32_t* out_dev, int64_t n, void* stream) {
    if (!ctr_key_dev || !out_dev || n < 1) return fail_with(FWG_ERR_INVALID, "fwg_selftest_philox: bad argument");
    hipLaunchKernelGGL(k_selftest_philox, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ctr_key_dev, out_dev, (long)n);
    HIP_TRY(hipGetLastError());
    return FWG_OK;
}

int fwg_check_actions(fwg_handle* h, const float* actions, void* stream) {
    if (!h || !actions) return fail_with(FWG_ERR_INVALID, "null argument");
    const long n = (long)h->n_envs * 3;
    HIP_TRY(hipMemsetAsync(h->d_flag, 0, sizeof(int), (hipStream_t)stream));
    hipLaunchKernelGGL(k_check_nan, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, actions, n, h->d_flag);
    int flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (flag) return fail_with(FWG_ERR_NAN_ACTION, "NaN in actions");
    return FWG_OK;
}

static void launch_finish(fwg_handle* h, float* metrics_out, hipStream_t stream) {
    if (!h->h.metrics) return;
    KArgs A;
    base_args(h, &A);
    A.metrics = metrics_out;
    hipLaunchKernelGGL(k_finish<false>, dim3((unsigned)((h->n_envs + FWG_WAVE - 1) / FWG_WAVE)), dim3(FWG_WAVE), 0, stream, h->d_cfg, A, (float*)nullptr);
}
int fwg_finish_episodes(fwg_handle* h, float* metrics_out, void* stream) {
    if (!h) return fail_with(FWG_ERR_INVALID, "null handle");
    launch_finish(h, metrics_out, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return FWG_OK;
}

static inline float reduce_to_float(int i, unsigned long long q) {
    return i < 5 ? (float)(long long)q : (float)((double)(long long)q / (double)FWG_ACC_SCALE);
}
int fwg_reduce_success(fwg_handle* h, float* out_host, void* stream) {
    if (!h || !out_host) return fail_with(FWG_ERR_INVALID, "null argument");
    unsigned long long q[FWG_N_REDUCE];
    launch_finish(h, h->last_metrics_out, (hipStream_t)stream);
    HIP_TRY(hipMemcpyAsync(q, h->d_reduce, sizeof(q), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipMemsetAsync(h->d_reduce, 0, sizeof(q), (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < FWG_N_REDUCE; ++i) out_host[i] = reduce_to_float(i, q[i]);
    return FWG_OK;
}

int fwg_reduce_success_device(fwg_handle* h, float* out_dev, void* stream) {
    if (!h || !out_dev) return fail_with(FWG_ERR_INVALID, "null argument");
    // default: collection, then a one-wave launch that hands the sums out and clears them.  FWGYM_TAKE=ticket: ONE launch
    // (k_finish<true>: the block that finishes last does it) -- see k_finish for what each costs
    static const bool ticket = [] { const char* e = getenv("FWGYM_TAKE"); return e != nullptr && e[0] == 't'; }();
    if (ticket) {
        KArgs A;
        base_args(h, &A);
        A.metrics = h->h.metrics ? h->last_metrics_out : nullptr;
        hipLaunchKernelGGL(k_finish<true>, dim3((unsigned)((h->n_envs + FWG_WAVE - 1) / FWG_WAVE)), dim3(FWG_WAVE), 0, (hipStream_t)stream,
                           h->d_cfg, A, out_dev);
    } else {
        launch_finish(h, h->last_metrics_out, (hipStream_t)stream);
        hipLaunchKernelGGL(k_reduce_take, dim3(1), dim3(FWG_WAVE), 0, (hipStream_t)stream, h->d_reduce, out_dev);
    }
    HIP_TRY(hipGetLastError());
    return FWG_OK;
}

int fwg_spec_index(const fwg_handle* h) { return h ? h->spec : -1; }
#ifdef FWG_TIMELINE
int fwg_debug_set_trace(fwg_handle* h, long long* trace_dev) { if (!h) return FWG_ERR_INVALID; h->trace = trace_dev; return FWG_OK; }
int fwg_debug_set_actor_trace(struct fwg_actor* a, long long* trace_dev);
#endif

int fwg_set_graph_mode(fwg_handle* h, int enable, void* stream) {
    if (!h) return fail_with(FWG_ERR_INVALID, "null handle");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (enable) {   // the launch for step g reads copy [g & 1]; the other copy still describes step g - 1 (what a reader
        StepSlots s[2];   // of the LAST completed step's window wants, see fwg_actor_set_obs_log)
        s[h->gstep & 1] = host_slots(h, h->gstep);
        s[(h->gstep + 1) & 1] = host_slots(h, h->gstep - 1);
        HIP_TRY(hipMemcpy(h->d_slots, s, sizeof(s), hipMemcpyHostToDevice));
    } else if (h->graph_mode) {
        StepSlots s[2];
        HIP_TRY(hipMemcpy(s, h->d_slots, sizeof(s), hipMemcpyDeviceToHost));
        h->gstep = s[0].gnow > s[1].gnow ? s[0].gnow : s[1].gnow;
    }
    h->graph_mode = enable ? 1 : 0;
    return FWG_OK;
}

int fwg_capture_begin(fwg_handle* h) {
    if (!h || !h->graph_mode) return fail_with(FWG_ERR_INVALID, "fwg_capture_begin needs graph mode");
    // simulator.model / randomize_scaling: whether a step launches the full-grid draw (every prepared set stale) or the queue
    // kernel is decided on the HOST when the launch is issued -- a sequence captured while the sets are stale would replay
    // the full-grid draw for ever, and one captured before a generation bump would never run it (fwg_replay_check)
    if (h->d_mq != nullptr && h->model_all_stale)
        return fail_with(FWG_ERR_INVALID, "fwg_capture_begin: the per-env parameter sets are stale (start, fwg_update_config, fwg_seed): "
                                          "issue fwg_reset or an even number of direct fwg_step calls before capturing");
    h->gstep_at_capture = h->gstep;
    h->generation_at_capture = h->hd.generation;
    h->spec_at_capture = h->spec;
    return FWG_OK;
}
int fwg_capture_end(fwg_handle* h) {   // the captured calls did not execute: take the host count back
    if (!h || !h->graph_mode) return fail_with(FWG_ERR_INVALID, "fwg_capture_end needs graph mode");
    if ((h->gstep - h->gstep_at_capture) % 2 != 0) return fail_with(FWG_ERR_INVALID, "captured an odd number of steps");
    h->gstep = h->gstep_at_capture;
    return FWG_OK;
}

int fwg_note_replayed_steps(fwg_handle* h, int64_t n_steps) {
    if (!h || n_steps < 0) return fail_with(FWG_ERR_INVALID, "bad argument");
    if (!h->graph_mode) return fail_with(FWG_ERR_INVALID, "fwg_note_replayed_steps needs graph mode");
    if (n_steps % 2 != 0) return fail_with(FWG_ERR_INVALID, "a replayed launch sequence must hold an even number of steps");
    h->gstep += n_steps;
    return FWG_OK;
}
int fwg_capture_parity(const fwg_handle* h) { return h ? (int)(h->gstep_at_capture & 1) : 0; }
int fwg_replay_check(const fwg_handle* h, int capture_parity) {
    if (!h || !h->graph_mode) return fail_with(FWG_ERR_INVALID, "fwg_replay_check needs graph mode");
    if (h->d_mq != nullptr && (h->hd.generation != h->generation_at_capture || h->model_all_stale))
        return fail_with(FWG_ERR_INVALID, "hipGraph captured before fwg_update_config / fwg_seed: its launches would keep using per-env "
                                          "parameter sets drawn under the old ranges (issue two direct fwg_step calls, then capture again)");
    if (h->spec != h->spec_at_capture)
        return fail_with(FWG_ERR_INVALID, "hipGraph captured before an fwg_update_config that moved the configuration to another kernel instance "
                                          "(a frozen configuration's kernel has its values folded in: set_curriculum_level on a preset leaves "
                                          "it for the shape instance): its launches would keep computing with the old values -- capture again");
    if ((int)(h->gstep & 1) != (capture_parity & 1))
        return fail_with(FWG_ERR_INVALID, "hipGraph captured at the other step parity: its launches would read the stale copy of the ring "
                                          "positions (run an even number of direct steps between capture and replay, or capture again)");
    return FWG_OK;
}
int fwg_num_specs(void) {
    int n = 0;
    while (kSpecTable[n] != nullptr) ++n;
    return n;
}

int fwg_config_instance(const fwg_config* cfg) {
    if (!cfg) return fail_with(FWG_ERR_INVALID, "null argument");
    if (cfg->abi_version != FWG_ABI_VERSION || cfg->struct_bytes != sizeof(fwg_config))
        return fail_with(FWG_ERR_ABI, "fwg_config version/size mismatch");
    DevCfg d;
    DynCfg dy;
    std::string why;
    if (lower_config(*cfg, &d, &dy, &why) != 0) return fail_with(FWG_ERR_INVALID, why);
    const int i = match_spec(d);
    return i < 0 ? FWG_INSTANCE_GENERIC : i;
}

int fwg_dump_spec(const fwg_config* cfg, uint32_t* words_out, int64_t capacity) {
    if (!cfg || !words_out) return fail_with(FWG_ERR_INVALID, "null argument");
    if (cfg->abi_version != FWG_ABI_VERSION || cfg->struct_bytes != sizeof(fwg_config))
        return fail_with(FWG_ERR_ABI, "fwg_config version/size mismatch");
    DevCfg d;
    DynCfg dy;
    std::string why;
    if (lower_config(*cfg, &d, &dy, &why) != 0) return fail_with(FWG_ERR_INVALID, why);
    const int64_t n = (int64_t)(sizeof(DevCfg) / 4);
    if (capacity < n) return fail_with(FWG_ERR_INVALID, "fwg_dump_spec: buffer too small");
    memcpy(words_out, &d, sizeof(DevCfg));
    return (int)n;
}

}  // extern "C"

template <bool IS_STEP, bool TURB, int SPEC>
static void launch_one(const fwg_handle* h, const KArgs& A, hipStream_t stream) {
    const dim3 grid((unsigned)((h->n_envs + FWG_WAVE - 1) / FWG_WAVE)), block(FWG_WAVE);
    const size_t lds_bytes = (size_t)lds_map(h->h.obs_dim, h->h.n_obs, h->h.L.window, h->h.use_cmd_ring, SPEC < 0, h->h.obs_log).total * sizeof(float);
    if (IS_STEP) {
#ifndef FWG_NO_SPLIT
        if (SPEC >= 0 && (h->split || SPEC >= FWG_SHAPE_BASE)) {   // (shape instances exist as the two-wave kernel only)
            const size_t lds2 = (size_t)lds_map(h->h.obs_dim, h->h.n_obs, h->h.L.window, h->h.use_cmd_ring, false, h->h.obs_log, true).total * sizeof(float);
            hipLaunchKernelGGL((k_step2<TURB, (SPEC >= 0 ? SPEC : 0)>), grid, dim3(2 * FWG_WAVE), lds2, stream, h->d_cfg, h->d_dyn, A);
            return;
        }
#endif
#ifndef FWG_DEV_FAST_BUILD   /* tools/isa.py -DFWG_DEV_FAST_BUILD: only the two-wave / fused kernels of the frozen configuration */
        if constexpr (SPEC < FWG_SHAPE_BASE) hipLaunchKernelGGL((k_step<TURB, SPEC>), grid, block, lds_bytes, stream, h->d_cfg, h->d_dyn, A);
#endif
    } else hipLaunchKernelGGL((k_reset<TURB, SPEC>), grid, block, lds_bytes, stream, h->d_cfg, h->d_dyn, A);
}

template <bool IS_STEP>
static void launch(const fwg_handle* h, const KArgs& A, hipStream_t stream) {
    switch (h->spec) {
#define FWG_SPEC_CASE(i) \
    case i: launch_one<IS_STEP, (kSpec##i.turbulence != 0), i>(h, A, stream); return;
        FWG_SPEC_LIST(FWG_SPEC_CASE)
#define FWG_SHAPE_CASE(i) \
    case FWG_SHAPE_BASE + i: launch_one<IS_STEP, (kSpec##i.turbulence != 0), FWG_SHAPE_BASE + i>(h, A, stream); return;
        FWG_SHAPE_LIST(FWG_SHAPE_CASE)
        default: break;
    }
#ifndef FWG_DEV_FAST_BUILD
    if (h->h.turbulence) launch_one<IS_STEP, true, -1>(h, A, stream);
    else launch_one<IS_STEP, false, -1>(h, A, stream);
#endif
}

// =====================================================================================================================
// rollout head (include/fwgym.h "Rollout head"): VecNormalize statistics + MlpPolicy on the matrix cores
// =====================================================================================================================

struct fwg_actor {
    int device;
    int64_t n_envs, env_base;
    int D, act_dim, nk1;
    float gamma, clip_obs, clip_rew, eps;
    int training, precise, parity;
    uint64_t seed;
    ActorStats* d_stats;   // [2]
    unsigned long long* d_acc;   // [FWG_ACC_SETS][FWG_ACC_SHARDS][acc_cols] fixed-point batch moments (rotation: fwgym_actor.h)
    int acc_cols;
    frag_t* d_frags;
    float* d_bias;         // [2][FWG_ACT_BIAS_FLOATS]
    float* d_log_std;
    float* d_ret;
    size_t lds_act[2];     // dynamic LDS of k_actor_act<1>, <3>
    const fwg_handle* log_env;   // fwg_actor_set_obs_log: `obs` arguments are this env's row log
#ifdef FWG_TIMELINE
    long long* trace;
#endif
};

static size_t actor_lds_bytes(int nk1, int parts) {
    return (size_t)(actor_weight_floats(nk1, parts) + actor_scratch_floats()) * sizeof(float);
}

static ActorArgs actor_args(const fwg_actor* a) {
    ActorArgs A;
    memset(&A, 0, sizeof(A));
    A.ret = a->d_ret; A.stats = a->d_stats; A.frags = a->d_frags; A.bias = a->d_bias; A.log_std = a->d_log_std;
    A.acc = a->d_acc; A.acc_cols = a->acc_cols;
    A.N = (long)a->n_envs; A.env_base = (long)a->env_base;
    A.D = a->D; A.nk1 = a->nk1; A.act_dim = a->act_dim; A.parity = a->parity; A.training = a->training;
    A.gamma = a->gamma; A.clip_obs = a->clip_obs; A.clip_rew = a->clip_rew; A.eps = a->eps;
    A.seed_lo = (unsigned)(a->seed & 0xFFFFFFFFull); A.seed_hi = (unsigned)(a->seed >> 32);
#ifdef FWG_TIMELINE
    A.trace = a->trace;
#endif
    if (a->log_env != nullptr) {   // the window of the env's LAST completed step; read on the device in graph mode
        const fwg_handle* h = a->log_env;
        A.obs_n = h->h.n_obs;
        A.obs_win = host_slots(h, h->gstep - 1).log_win;
        A.obs_slots = h->graph_mode ? h->d_slots + ((h->gstep - 1) & 1) : nullptr;
    }
    return A;
}

static void observer_args(fwg_handle* h, KArgs* A) {
    fwg_actor* a = h->observer;
    if (!a) return;
    A->acc = a->d_acc;   // three sets; the step adds into set (act counter of the current statistics copy) % 3
    A->acc_ctr = &a->d_stats[a->parity].act_counter;
    A->acc_mean = a->d_stats[a->parity].mean; A->acc_ret_mean = &a->d_stats[a->parity].ret_mean;
    A->acc_ret = a->d_ret; A->acc_gamma = a->gamma; A->acc_cols = a->acc_cols;
}

extern "C" {

int fwg_attach_observer(fwg_handle* h, fwg_actor* a) {
    if (!h) return fail_with(FWG_ERR_INVALID, "fwg_attach_observer: null env handle");
    if (a && h->h.obs_log > 0)
        return fail_with(FWG_ERR_INVALID, "fwg_attach_observer: row-log observations keep the lagged rows out of the step kernel; use fwg_actor_observe");
    if (a && (a->n_envs != h->n_envs || a->D != h->h.obs_dim || a->device != h->device))
        return fail_with(FWG_ERR_INVALID, "fwg_attach_observer: the head was created for another batch size / observation size / device");
    h->observer = a;
    // the one-launch rollout step's dynamic LDS (both head variants), asked for here and now: attach is never captured
    h->rollout_lds_granted = 0;
    if (a && h->spec >= 0 && h->split && launch_rollout(h, a, KArgs(), ActorArgs(), nullptr, true) == 0)
        h->rollout_lds_granted = launch_rollout(h, a, KArgs(), ActorArgs(), nullptr, false, true) == 0 ? 1 : 0;
    return FWG_OK;
}

#ifdef FWG_TIMELINE
int fwg_debug_set_actor_trace(fwg_actor* a, long long* trace_dev) { if (!a) return FWG_ERR_INVALID; a->trace = trace_dev; return FWG_OK; }
#endif
int fwg_actor_set_obs_log(fwg_actor* a, const fwg_handle* env) {
    if (!a) return fail_with(FWG_ERR_INVALID, "fwg_actor_set_obs_log: null actor");
    if (env != nullptr) {
        if (env->h.obs_log <= 0) return fail_with(FWG_ERR_INVALID, "fwg_actor_set_obs_log: the env writes the dense observation batch");
        if (env->n_envs != a->n_envs || env->h.obs_dim != a->D || env->device != a->device || (env->h.n_obs & 3))
            return fail_with(FWG_ERR_INVALID, "fwg_actor_set_obs_log: batch size / observation size / device mismatch (n_obs must be a multiple of 4)");
    }
    a->log_env = env;
    return FWG_OK;
}

int fwg_actor_create(int device, int64_t n_envs, int obs_dim, int act_dim, float gamma, float clip_obs, float clip_reward,
                     float epsilon, fwg_actor** out) {
    if (!out) return fail_with(FWG_ERR_INVALID, "fwg_actor_create: null argument");
    if (n_envs < 1) return fail_with(FWG_ERR_INVALID, "fwg_actor_create: n_envs < 1");
    if (obs_dim < 1 || obs_dim > FWG_ACT_MAX_OBS) return fail_with(FWG_ERR_INVALID, "fwg_actor_create: obs_dim must be in [1, 64]");
    if (act_dim < 1 || act_dim > FWG_ACT_MAX_ACT) return fail_with(FWG_ERR_INVALID, "fwg_actor_create: act_dim must be in [1, 4]");
    HIP_TRY(hipSetDevice(device));
    fwg_actor* a = new fwg_actor();
    memset(a, 0, sizeof(*a));
    a->device = device; a->n_envs = n_envs; a->D = obs_dim; a->act_dim = act_dim; a->nk1 = (obs_dim + 15) / 16;
    a->gamma = gamma; a->clip_obs = clip_obs; a->clip_rew = clip_reward; a->eps = epsilon;
    a->training = 1; a->precise = 1;
    const size_t nfrag = (size_t)2 * 2 * actor_frags(a->nk1) * 64;
    HIP_TRY(hipMalloc((void**)&a->d_stats, 2 * sizeof(ActorStats)));
    a->acc_cols = acc_cols_for(obs_dim);
    const size_t acc_bytes = (size_t)FWG_ACC_SETS * FWG_ACC_SHARDS * a->acc_cols * sizeof(unsigned long long);
    HIP_TRY(hipMalloc((void**)&a->d_acc, acc_bytes));
    HIP_TRY(hipMemset(a->d_acc, 0, acc_bytes));
    HIP_TRY(hipMalloc((void**)&a->d_frags, nfrag * sizeof(frag_t)));
    HIP_TRY(hipMalloc((void**)&a->d_bias, 2 * FWG_ACT_BIAS_FLOATS * sizeof(float)));
    HIP_TRY(hipMemset(a->d_bias, 0, 2 * FWG_ACT_BIAS_FLOATS * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&a->d_log_std, FWG_ACT_MAX_ACT * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&a->d_ret, (size_t)n_envs * sizeof(float)));
    HIP_TRY(hipMemset(a->d_frags, 0, nfrag * sizeof(frag_t)));
    HIP_TRY(hipMemset(a->d_log_std, 0, FWG_ACT_MAX_ACT * sizeof(float)));
    HIP_TRY(hipMemset(a->d_ret, 0, (size_t)n_envs * sizeof(float)));
    ActorStats s[2];
    memset(s, 0, sizeof(s));
    for (int p = 0; p < 2; ++p) {   // RunningMeanStd(epsilon=1e-4): mean 0, var 1, count 1e-4
        for (int f = 0; f < FWG_ACT_MAX_OBS; ++f) s[p].var[f] = 1.f;
        s[p].count = 1e-4f; s[p].ret_var = 1.f; s[p].ret_count = 1e-4f;
    }
    HIP_TRY(hipMemcpy(a->d_stats, s, sizeof(s), hipMemcpyHostToDevice));
    a->lds_act[0] = actor_lds_bytes(a->nk1, 1);
    a->lds_act[1] = actor_lds_bytes(a->nk1, 2);
    // more than 64 KiB of dynamic LDS per workgroup has to be asked for (gfx950: 160 KiB per CU)
    {
        const void* k1[4] = {(const void*)k_actor_act<1, 1>, (const void*)k_actor_act<1, 2>, (const void*)k_actor_act<1, 3>, (const void*)k_actor_act<1, 4>};
        const void* k3[4] = {(const void*)k_actor_act<3, 1>, (const void*)k_actor_act<3, 2>, (const void*)k_actor_act<3, 3>, (const void*)k_actor_act<3, 4>};
        HIP_TRY(hipFuncSetAttribute(k1[a->nk1 - 1], hipFuncAttributeMaxDynamicSharedMemorySize, (int)a->lds_act[0]));
        HIP_TRY(hipFuncSetAttribute(k3[a->nk1 - 1], hipFuncAttributeMaxDynamicSharedMemorySize, (int)a->lds_act[1]));
    }
    *out = a;
    return FWG_OK;
}

void fwg_actor_destroy(fwg_actor* a) {
    if (!a) return;
    (void)hipFree(a->d_stats); (void)hipFree(a->d_acc); (void)hipFree(a->d_frags); (void)hipFree(a->d_bias); (void)hipFree(a->d_log_std); (void)hipFree(a->d_ret);
    delete a;
}

int fwg_actor_set_weights(fwg_actor* a, const fwg_actor_weights* w) {
    if (!a || !w) return fail_with(FWG_ERR_INVALID, "fwg_actor_set_weights: null argument");
    const float* const need[] = {w->pi_w0, w->pi_b0, w->pi_w1, w->pi_b1, w->pi_w2, w->pi_b2,
                                 w->vf_w0, w->vf_b0, w->vf_w1, w->vf_b1, w->vf_w2, w->vf_b2, w->log_std};
    for (const float* p : need) if (!p) return fail_with(FWG_ERR_INVALID, "fwg_actor_set_weights: null weight array");
    HIP_TRY(hipSetDevice(a->device));
    std::vector<unsigned> all;
    std::vector<float> biases;
    for (int net = 0; net < 2; ++net) {
        std::vector<unsigned> hi, lo;
        const float* w0 = net ? w->vf_w0 : w->pi_w0; const float* b0 = net ? w->vf_b0 : w->pi_b0;
        const float* w1 = net ? w->vf_w1 : w->pi_w1; const float* b1 = net ? w->vf_b1 : w->pi_b1;
        const float* w2 = net ? w->vf_w2 : w->pi_w2; const float* b2 = net ? w->vf_b2 : w->pi_b2;
        const int out = net ? 1 : a->act_dim;
        actor_pack_layer(hi, lo, w0, 64, a->D, 2, a->nk1, false, FWG_ACT_PRESCALE);
        actor_pack_layer(hi, lo, w1, 64, 64, 2, 4, true, FWG_ACT_PRESCALE);
        actor_pack_layer(hi, lo, w2, out, 64, 1, 4, true, 1.f);
        all.insert(all.end(), hi.begin(), hi.end());
        all.insert(all.end(), lo.begin(), lo.end());
        actor_pack_bias(biases, b0, b1, b2, out);
    }
    HIP_TRY(hipMemcpy(a->d_bias, biases.data(), biases.size() * sizeof(float), hipMemcpyHostToDevice));
    const size_t want = (size_t)2 * 2 * actor_frags(a->nk1) * 64 * 4;
    if (all.size() != want) return fail_with(FWG_ERR_INVALID, "fwg_actor_set_weights: internal packing size mismatch");
    HIP_TRY(hipMemcpy(a->d_frags, all.data(), all.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    float ls[FWG_ACT_MAX_ACT] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < a->act_dim; ++i) ls[i] = w->log_std[i];
    HIP_TRY(hipMemcpy(a->d_log_std, ls, sizeof(ls), hipMemcpyHostToDevice));
    return FWG_OK;
}

int fwg_actor_set_stats(fwg_actor* a, const fwg_actor_stats* s, void* stream) {
    if (!a || !s) return fail_with(FWG_ERR_INVALID, "fwg_actor_set_stats: null argument");
    HIP_TRY(hipSetDevice(a->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    ActorStats cur;
    HIP_TRY(hipMemcpy(&cur, a->d_stats + a->parity, sizeof(cur), hipMemcpyDeviceToHost));
    memcpy(cur.mean, s->obs_mean, sizeof(cur.mean));
    memcpy(cur.var, s->obs_var, sizeof(cur.var));
    cur.count = s->obs_count; cur.ret_mean = s->ret_mean; cur.ret_var = s->ret_var; cur.ret_count = s->ret_count;
    HIP_TRY(hipMemcpy(a->d_stats + a->parity, &cur, sizeof(cur), hipMemcpyHostToDevice));
    return FWG_OK;
}

int fwg_actor_get_stats(fwg_actor* a, fwg_actor_stats* s, void* stream) {
    if (!a || !s) return fail_with(FWG_ERR_INVALID, "fwg_actor_get_stats: null argument");
    HIP_TRY(hipSetDevice(a->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    ActorStats cur;
    HIP_TRY(hipMemcpy(&cur, a->d_stats + a->parity, sizeof(cur), hipMemcpyDeviceToHost));
    memcpy(s->obs_mean, cur.mean, sizeof(cur.mean));
    memcpy(s->obs_var, cur.var, sizeof(cur.var));
    s->obs_count = cur.count; s->ret_mean = cur.ret_mean; s->ret_var = cur.ret_var; s->ret_count = cur.ret_count;
    return FWG_OK;
}

int fwg_actor_configure(fwg_actor* a, int training, int precise) {
    if (!a) return fail_with(FWG_ERR_INVALID, "fwg_actor_configure: null handle");
    a->training = training ? 1 : 0;
    a->precise = precise ? 1 : 0;
    return FWG_OK;
}

int fwg_actor_seed(fwg_actor* a, uint64_t seed, int64_t env_id_base) {
    if (!a) return fail_with(FWG_ERR_INVALID, "fwg_actor_seed: null handle");
    a->seed = seed; a->env_base = env_id_base;
    return FWG_OK;
}

int fwg_actor_observe(fwg_actor* a, const float* obs, const float* reward, const uint8_t* done, void* stream) {
    if (!a) return fail_with(FWG_ERR_INVALID, "fwg_actor_observe: null handle");
    if (!obs && !reward) return FWG_OK;
    HIP_TRY(hipSetDevice(a->device));
    ActorArgs A = actor_args(a);
    A.obs = obs; A.rew = reward; A.done = done;
    const dim3 grid((unsigned)((a->n_envs + FWG_ACT_BLOCK - 1) / FWG_ACT_BLOCK)), block(FWG_ACT_BLOCK);
    hipLaunchKernelGGL(k_actor_stats, grid, block, 0, (hipStream_t)stream, A);
    HIP_TRY(hipGetLastError());
    return FWG_OK;
}

int fwg_actor_act(fwg_actor* a, const float* obs, const float* reward, const uint8_t* done, float* norm_obs_out,
                  float* action_out, float* value_out, float* logp_out, float* norm_reward_out, uint8_t* done_out,
                  int deterministic, void* stream) {
    if (!a || !obs) return fail_with(FWG_ERR_INVALID, "fwg_actor_act: null argument");
    HIP_TRY(hipSetDevice(a->device));
    ActorArgs A = actor_args(a);
    A.obs = obs; A.rew = reward; A.done = done;
    A.norm_obs = norm_obs_out; A.action = action_out; A.value = value_out; A.logp = logp_out; A.norm_rew = norm_reward_out;
    A.done_out = done_out; A.deterministic = deterministic ? 1 : 0;
    const dim3 grid((unsigned)((a->n_envs + FWG_ACT_ENVS - 1) / FWG_ACT_ENVS)), block(64 * FWG_ACT_WAVES);
    hipStream_t st = (hipStream_t)stream;
#define FWG_ACT_LAUNCH(NK)                                                                                \
    case NK:                                                                                              \
        if (a->precise) hipLaunchKernelGGL((k_actor_act<3, NK>), grid, block, a->lds_act[1], st, A);     \
        else hipLaunchKernelGGL((k_actor_act<1, NK>), grid, block, a->lds_act[0], st, A);                \
        break;
    switch (a->nk1) { FWG_ACT_LAUNCH(1) FWG_ACT_LAUNCH(2) FWG_ACT_LAUNCH(3) FWG_ACT_LAUNCH(4) default: break; }
#undef FWG_ACT_LAUNCH
    HIP_TRY(hipGetLastError());
    a->parity ^= 1;
    return FWG_OK;
}

}  // extern "C"
// ---- the head and the env step in one launch
// grant = true (fwg_attach_observer, never inside a stream capture): asks for the dynamic LDS of BOTH head variants -- more than
// 64 KiB per workgroup has to be requested per kernel; done eagerly and recorded in the handle, so that no launch (possibly the
// first one of a variant inside a capture, after fwg_actor_configure) ever has to
template <bool TURB, int SPEC>
static int launch_rollout_one(fwg_handle* h, fwg_actor* a, const KArgs& A, const ActorArgs& AA, hipStream_t stream, bool grant) {
    if constexpr (SPEC >= 0) {
        if constexpr (SpecCfg<SPEC>::rollout_ok) {
            if (grant) {
                const size_t b3 = (size_t)rollout_lds_floats(h->h, 3) * sizeof(float), b1 = (size_t)rollout_lds_floats(h->h, 1) * sizeof(float);
                if (b3 > 160 * 1024 || b1 > 160 * 1024) return 1;
                if (hipFuncSetAttribute((const void*)k_rollout<TURB, SPEC, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b3) != hipSuccess) return 1;
                if (hipFuncSetAttribute((const void*)k_rollout<TURB, SPEC, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b1) != hipSuccess) return 1;
                return 0;
            }
            const int hsplit = a->precise ? 3 : 1;
            const size_t bytes = (size_t)rollout_lds_floats(h->h, hsplit) * sizeof(float);
            if (bytes > 160 * 1024 || !h->rollout_lds_granted) return 1;
            const dim3 grid((unsigned)((h->n_envs + FWG_RO_ENVS - 1) / FWG_RO_ENVS)), block(2 * FWG_RO_ENVS);
            if (a->precise) hipLaunchKernelGGL((k_rollout<TURB, SPEC, 3>), grid, block, bytes, stream, h->d_cfg, h->d_dyn, A, AA);
            else hipLaunchKernelGGL((k_rollout<TURB, SPEC, 1>), grid, block, bytes, stream, h->d_cfg, h->d_dyn, A, AA);
            return 0;
        }
    }
    return 1;
}
static int launch_rollout(fwg_handle* h, fwg_actor* a, const KArgs& A, const ActorArgs& AA, hipStream_t stream, bool probe, bool grant) {
    switch (h->spec) {
#define FWG_SPEC_RO(i) \
    case i: if (probe) return (SpecCfg<i>::rollout_ok && (size_t)rollout_lds_floats(kSpec##i, 3) * sizeof(float) <= 160 * 1024) ? 0 : 1; \
            return launch_rollout_one<(kSpec##i.turbulence != 0), i>(h, a, A, AA, stream, grant);
        FWG_SPEC_LIST(FWG_SPEC_RO)
#undef FWG_SPEC_RO
        default: break;
    }
    return 1;
}

extern "C" {

int fwg_rollout_available(const fwg_handle* h, const fwg_actor* a) {
    if (!h || !a || h->observer != a || h->spec < 0 || !h->split || h->h.obs_log > 0) return 0;
    if (h->h.model_n > 0 || h->h.randomize_scaling) return 0;   // (the per-env parameter queue launch sits between head and step)
    if (a->act_dim != 3 || a->log_env != nullptr) return 0;
    if (!h->rollout_lds_granted) return 0;
    return launch_rollout(const_cast<fwg_handle*>(h), const_cast<fwg_actor*>(a), KArgs(), ActorArgs(), nullptr, true) == 0 ? 1 : 0;
}

int fwg_rollout_step(fwg_handle* h, fwg_actor* a, float* norm_obs_out, float* action_out, float* value_out, float* logp_out,
                     float* norm_reward_out, uint8_t* done_prev_out, float* obs_io, float* reward_io, uint8_t* done_io,
                     uint8_t* term_code_out, float* terminal_obs_out, float* metrics_out, int deterministic, void* stream) {
    if (!h || !a || !obs_io || !reward_io || !done_io || !term_code_out) return fail_with(FWG_ERR_INVALID, "fwg_rollout_step: null argument");
    if (!fwg_rollout_available(h, a))
        return fail_with(FWG_ERR_INVALID, "fwg_rollout_step: needs a build-time specialised configuration with the dense observation batch "
                                          "and this head attached (fwg_attach_observer); use fwg_step + fwg_actor_act otherwise");
    HIP_TRY(hipSetDevice(a->device));
    // head: on the observation / reward / done flags the env's last step left in obs_io / reward_io / done_io
    ActorArgs AA = actor_args(a);
    AA.obs = obs_io; AA.rew = reward_io; AA.done = done_io;
    AA.norm_obs = norm_obs_out; AA.action = action_out; AA.value = value_out; AA.logp = logp_out; AA.norm_rew = norm_reward_out;
    AA.done_out = done_prev_out; AA.deterministic = deterministic ? 1 : 0;
    // env step: as fwg_step, the actions come from the head phase (LDS)
    KArgs A;
    base_args(h, &A);
    A.actions = nullptr; A.obs = obs_io; A.rew = reward_io; A.done = done_io; A.term = term_code_out;
    A.term_obs = terminal_obs_out; A.metrics = metrics_out; A.tgt_out = nullptr;
    h->last_metrics_out = metrics_out;
    fill_slots(h, h->gstep, &A);
    if (h->graph_mode) { A.slots_in = h->d_slots + (h->gstep & 1); A.slots_out = h->d_slots + ((h->gstep + 1) & 1); }
    a->parity ^= 1;          // the step phase belongs to the statistics copy the head phase publishes ...
    observer_args(h, &A);    // (... its means and counter are taken from LDS in the kernel; acc / ret / gamma from here)
#ifdef FWG_TIMELINE
    A.trace = h->trace;
#endif
    if (launch_rollout(h, a, A, AA, (hipStream_t)stream, false) != 0) {
        a->parity ^= 1;
        return fail_with(FWG_ERR_INVALID, "fwg_rollout_step: no fused launch for this configuration");
    }
    HIP_TRY(hipGetLastError());
    h->gstep += 1;
    return FWG_OK;
}

}  // extern "C"
