// fwgym_actor.h -- the rollout head between two env steps: VecNormalize statistics + MlpPolicy forward + sampling
// (include/fwgym.h "Rollout head"; reference call sites examples/train_rl_controller.py:223-231,
// examples/evaluate_controller.py:93-100).  Two kernels:
//   k_actor_stats : batch moments of the observation (and discounted-return) batch -> global accumulators
//   k_actor_act   : folds the moments into the running statistics, normalises, runs pi and vf on the matrix cores,
//                   samples, writes the rollout-buffer slices
// MLP on MFMA, transposed formulation H^T = W * X^T so that the accumulator registers of one layer ARE the B operand
// of the next (no cross-lane movement between layers):
//   v_mfma_f32_32x32x16_bf16: D[i][j] += sum_k A[i][k] B[k][j]; lane l supplies row i = l & 31 of A and column
//   j = l & 31 of B, eight k-slots (half = l >> 5, t = 0..7); lane l receives D[(r & 3) + 8 (r >> 2) + 4 half][j],
//   r = 0..15.  Columns j = environments, rows i = output units.  A lane's 16 results of a 32-row tile are 2 x 8
//   k-slots of the next layer, provided the weight fragments are packed with the matching k permutation
//   (k_chained below) -- done once on the host.  Biases ride in one extra k-block against a constant "ones" operand.
// fp32 accuracy from bf16 matrix cores: every operand is split x = hi + lo (both bf16), and
// A B ~ A_hi B_hi + A_lo B_hi + A_hi B_lo (error ~2^-17 relative per product).
#pragma once
#include "fwgym_env.h"

#define FWG_STREAM_POLICY 6u
#define FWG_ACT_BLOCK 256          // k_actor_stats: threads = environments per block
#define FWG_ACT_WAVES 8            // k_actor_act: waves per block, one 32-environment tile each
#define FWG_ACT_ENVS (32 * FWG_ACT_WAVES)
// hidden-layer weights and biases are packed pre-multiplied by 2 log2(e), so that tanh(z) = 1 - 2 / (2^acc + 1)
#define FWG_ACT_PRESCALE 2.885390081777927f
#define FWG_ACT_MAX_OBS 64
#define FWG_ACT_MAX_ACT 4

// running statistics + the act counter (Philox counter of the sampling noise), double-buffered by parity: every
// k_actor_act reads copy [parity] and publishes copy [parity ^ 1], so captured launch sequences replay correctly
struct ActorStats { float mean[FWG_ACT_MAX_OBS], var[FWG_ACT_MAX_OBS]; float count, ret_mean, ret_var, ret_count; unsigned act_counter, pad_[3]; };
// Batch moments reach k_actor_act through the fixed-point accumulators described in fwgym_env.h (acc_*), filled either
// by k_actor_stats or by the env step kernel itself (fwg_attach_observer).  THREE sets, rotating with the act counter c of
// the statistics copy a launch reads: producers (k_actor_stats, the env step) add into set c % 3, the act with counter c
// consumes set c % 3, publishes c + 1 and clears set (c + 2) % 3 -- the set the producers AFTER THE NEXT act will add into.
// No launch ever clears a set that the same launch reads or adds to, so the act and the env step that follows it can be
// ONE launch (k_rollout: its blocks consume set c % 3, add into (c + 1) % 3 and block 0 clears (c + 2) % 3), and a captured
// launch sequence replays whatever its length (the rotation lives on the device, not in the captured arguments).

struct alignas(16) frag_t { unsigned x, y, z, w; };   // 8 bf16 = the A or B operand of one lane

struct ActorArgs {
    const float* obs; const float* rew; const uint8_t* done;
    // obs_n > 0: `obs` is an env's observation ROW LOG (fwgym_env.h): feature f = r obs_n + j of env e sits at
    // obs[((win + r) N + e) obs_n + j], win = obs_slots->log_win (graph mode: read on the device) or obs_win (host value)
    const StepSlots* obs_slots; long long obs_win; int obs_n;
#ifdef FWG_TIMELINE
    long long* trace;   // measurement builds: [block][wave][16] s_memtime stamps
#endif
    float* ret;
    ActorStats* stats;                  // [2], indexed by parity
    unsigned long long* acc; int acc_cols;   // [FWG_ACC_SETS][FWG_ACC_SHARDS][acc_cols]
    const frag_t* frags;                // [net 2][part hi/lo][frag][64 lanes]
    const float* bias;                  // [net 2][FWG_ACT_BIAS_FLOATS] fp32 biases (hidden layers pre-multiplied by 2 log2 e)
    const float* log_std;
    float* norm_obs; float* action; float* value; float* logp; float* norm_rew; uint8_t* done_out;
    long N; long env_base;
    int D, nk1, act_dim, parity, training, deterministic;
    float gamma, clip_obs, clip_rew, eps;
    unsigned seed_lo, seed_hi;
};

// frags per network and part: layer 1: 2 row tiles x nk1, layer 2: 2 x 4, layer 3: 1 x 4 (the biases are kept apart, in fp32)
__host__ __device__ inline int actor_frags(int nk1) { return 2 * nk1 + 2 * 4 + 4; }
#define FWG_ACT_BIAS_FLOATS 160   /* per network: layer 1 (64) | layer 2 (64) | layer 3 (32, act_dim / 1 used) */
// logical input index of k-slot (kk, half, t): first layer = features in order; later layers = the accumulator
// registers of the previous layer in register order (see the header comment)
__host__ __device__ inline int k_input(int kk, int half, int t) { return 16 * kk + 8 * half + t; }
__host__ __device__ inline int k_chained(int kk, int half, int t) {
    return 32 * (kk >> 1) + 8 * (2 * (kk & 1) + (t >> 2)) + 4 * half + (t & 3);
}

__host__ __device__ inline unsigned bf16_rne(float x) {   // round-to-nearest-even bf16 bits (finite inputs)
    unsigned u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(x);
#else
    memcpy(&u, &x, 4);
#endif
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__host__ __device__ inline float bf16_to_f32(unsigned h) {
    const unsigned u = h << 16;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}

// address of features [f0, f0 + 4) of env e (f0 a multiple of 4; in log mode obs_n is a multiple of 4 too)
__device__ __forceinline__ const float* actor_obs_at(const ActorArgs& A, long long win, long e, int f0) {
    if (A.obs_n == 0) return A.obs + e * A.D + f0;
    const int r = f0 / A.obs_n, j = f0 - r * A.obs_n;
    return A.obs + ((win + r) * A.N + e) * A.obs_n + j;
}
__device__ __forceinline__ long long actor_obs_win(const ActorArgs& A) { return A.obs_slots != nullptr ? A.obs_slots->log_win : A.obs_win; }

#ifdef FWG_TIMELINE
#define FWG_ATL(A, i) do { if ((A).trace != nullptr) { const long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); \
        if ((threadIdx.x & 63) == 0) (A).trace[(blockIdx.x * FWG_ACT_WAVES + (threadIdx.x >> 6)) * 16 + (i)] = t_; } } while (0)
#else
#define FWG_ATL(A, i) do { } while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifdef FWG_EMU
#define fwg_mfma_bf16(a, b, c) emu_mfma_f32_32x32x16_bf16((a), (b), (c))
#define fwg_exp2(x) exp2f(x)
#else
typedef __bf16 fwg_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 fwg_mfma_bf16(const frag_t& a, const frag_t& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fwg_bf16x8, a), __builtin_bit_cast(fwg_bf16x8, b), c, 0, 0, 0);
}
#define fwg_exp2(x) __builtin_amdgcn_exp2f(x)
#endif

// eight fp32 values -> bf16 hi and lo operands (round-to-nearest-even both; v_cvt_pk_bf16_f32 on the device)
#ifdef FWG_EMU
__device__ __forceinline__ void split8(const float (&x)[8], frag_t& hi, frag_t& lo) {
    unsigned h[8], l[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        h[t] = bf16_rne(x[t]);
        l[t] = bf16_rne(x[t] - bf16_to_f32(h[t]));
    }
    hi = frag_t{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
    lo = frag_t{l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
}
#else
typedef __bf16 fwg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float fwg_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    const fwg_f32x2 v = {a, b};
    const fwg_bf16x2 h = __builtin_convertvector(v, fwg_bf16x2);
#ifdef FWG_ABL_NO_LO   /* measurement only: no low part */
    hi = __builtin_bit_cast(unsigned, h); lo = 0u; return;
#endif
    const fwg_f32x2 r = v - __builtin_convertvector(h, fwg_f32x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, fwg_bf16x2));
}
__device__ __forceinline__ void split8(const float (&x)[8], frag_t& hi, frag_t& lo) {
    split2(x[0], x[1], hi.x, lo.x); split2(x[2], x[3], hi.y, lo.y);
    split2(x[4], x[5], hi.z, lo.z); split2(x[6], x[7], hi.w, lo.w);
}
#endif

// tanh of a pre-activation that arrives multiplied by 2 log2(e): 1 - 2 / (e^{2z} + 1), absolute error ~1e-7
__device__ __forceinline__ float tanh_prescaled(float a) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(fwg_exp2(a) + 1.f);
}

// acc += A * B over the split parts (a_lo unused for single products)
template <int SPLIT>
__device__ __forceinline__ f32x16 mma3(const frag_t& a_hi, const frag_t& a_lo, const frag_t& b_hi, const frag_t& b_lo, f32x16 acc) {
    if (SPLIT > 1) {
        acc = fwg_mfma_bf16(a_lo, b_hi, acc);
        acc = fwg_mfma_bf16(a_hi, b_lo, acc);
    }
    return fwg_mfma_bf16(a_hi, b_hi, acc);
}
// the same k-block into TWO accumulator tiles, alternating between them: an MFMA never follows the one whose result it
// accumulates onto (back-to-back dependent MFMAs wait for the whole pipe depth: 130 instead of 32 ticks each, measured)
template <int SPLIT>
__device__ __forceinline__ void mma3x2(const frag_t& a0_hi, const frag_t& a0_lo, const frag_t& a1_hi, const frag_t& a1_lo, const frag_t& b_hi,
                                       const frag_t& b_lo, f32x16& acc0, f32x16& acc1) {
    if (SPLIT > 1) {
        acc0 = fwg_mfma_bf16(a0_lo, b_hi, acc0); acc1 = fwg_mfma_bf16(a1_lo, b_hi, acc1);
        acc0 = fwg_mfma_bf16(a0_hi, b_lo, acc0); acc1 = fwg_mfma_bf16(a1_hi, b_lo, acc1);
    }
    acc0 = fwg_mfma_bf16(a0_hi, b_hi, acc0); acc1 = fwg_mfma_bf16(a1_hi, b_hi, acc1);
}
// one k-block into the three partial accumulators of a single output tile (layer 3: one tile, so the three products of the
// split get an accumulator each; added up at the end)
template <int SPLIT>
__device__ __forceinline__ void mma3s(const frag_t& a_hi, const frag_t& a_lo, const frag_t& b_hi, const frag_t& b_lo, f32x16& acc, f32x16& acc_a,
                                      f32x16& acc_b) {
    if (SPLIT > 1) {
        acc_a = fwg_mfma_bf16(a_lo, b_hi, acc_a);
        acc_b = fwg_mfma_bf16(a_hi, b_lo, acc_b);
    }
    acc = fwg_mfma_bf16(a_hi, b_hi, acc);
}
// NF consecutive weight fragments of one network from LDS (F = [part][nfw][64]) into registers: requested ahead of the MFMAs
// that consume them, so that no MFMA waits for an LDS round trip
template <int SPLIT, int NF>
__device__ __forceinline__ void load_frags(const frag_t* F, int nfw, int base, int l, frag_t (&hi)[NF], frag_t (&lo)[NF]) {
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        hi[i] = F[(base + i) * 64 + l];
        if (SPLIT > 1) lo[i] = F[(nfw + base + i) * 64 + l];
    }
}
// accumulator tile `it` (rows 32 it ... 32 it + 31) initialised with the layer's bias: lane l receives rows
// (r & 3) + 8 (r >> 2) + 4 half of every column, i.e. four 16-byte pieces of the bias vector (fp32, exact -- the bias used to
// ride in an extra k-block against a constant "ones" operand: 10 of the 104 MFMAs per tile)
__device__ __forceinline__ f32x16 bias_tile(const float* b, int it, int half) {
    f32x16 a;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(b + 32 * it + 8 * q + 4 * half);
        a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
    return a;
}

// eight accumulator values (one k-block of the next layer: registers 8 kb ... 8 kb + 7 of tile a) -> tanh -> B operand
__device__ __forceinline__ void hidden_block(const f32x16& a, int kb, frag_t& hi, frag_t& lo) {
    float x[8];
#if defined(FWG_ABL_TANH_ID)   /* measurement only: no tanh */
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = a[8 * kb + t] * 0.01f;
#elif !defined(FWG_EMU)
#pragma unroll
    for (int t = 0; t < 8; t += 2) {   // the add and the fma of 1 - 2 / (2^a + 1) on pairs (v_pk_add_f32 / v_pk_fma_f32)
        const float u0 = a[8 * kb + t], u1 = a[8 * kb + t + 1];
        const fwg_f32x2 d = fwg_f32x2{fwg_exp2(u0), fwg_exp2(u1)} + fwg_f32x2{1.f, 1.f};
        const fwg_f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        const fwg_f32x2 y = __builtin_elementwise_fma(r, fwg_f32x2{-2.f, -2.f}, fwg_f32x2{1.f, 1.f});
        x[t] = y[0]; x[t + 1] = y[1];
    }
#else
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = tanh_prescaled(a[8 * kb + t]);
#endif
    split8(x, hi, lo);
}
// two accumulator tiles (64 hidden units) -> the four k-blocks of the next layer's B operand
__device__ __forceinline__ void hidden_to_b(const f32x16& a0, const f32x16& a1, frag_t (&hi)[4], frag_t (&lo)[4]) {
    hidden_block(a0, 0, hi[0], lo[0]); hidden_block(a0, 1, hi[1], lo[1]);
    hidden_block(a1, 0, hi[2], lo[2]); hidden_block(a1, 1, hi[3], lo[3]);
}

// instruction-scheduling directives (LLVM SchedGroupMask): `n` groups of {1 MFMA, `valu` VALU / transcendental instructions}
#ifdef FWG_EMU
#define FWG_SCHED_MFMA_VALU(n, valu) do { } while (0)
#define FWG_SCHED_FENCE() do { } while (0)
#else
#define FWG_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define FWG_SCHED_MFMA_VALU(n, valu)                                        \
    do {                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               \
            __builtin_amdgcn_sched_group_barrier(0x002, (valu), 0);          \
        }                                                                    \
    } while (0)
#endif

// Both networks (obs -> 64 -> 64 -> out) for the 32 environments of this wave's tile, as ONE software pipeline: while the
// matrix pipe works through a layer of one network, the vector units turn the other network's previous layer into the next
// B operand (tanh + bf16 hi / lo split: ~180 VALU per layer and network, against 24 MFMAs = 768 pipe cycles), and every
// weight fragment is in registers before the MFMA that takes it is issued.  FP / FV: the networks' fragments in LDS
// ([part][nfw][64], nfw = 2 NK1 + 12: layer 1 tiles 0 / 1, layer 2 tiles 0 / 1, layer 3), bP / bV their biases (fp32: 64 | 64 |
// 32, hidden layers pre-multiplied by 2 log2 e like the weights).  Result rows 0..3 in out[0..3] of the lanes with half == 0.
template <int SPLIT, int NK1>
__device__ __forceinline__ void mlp_pair(const ActorArgs& A, const frag_t* FP, const frag_t* FV, const float* bP, const float* bV, int l,
                                         const frag_t (&bx_hi)[NK1], const frag_t (&bx_lo)[NK1], f32x16& o_pi, f32x16& o_vf) {
    constexpr int nfw = 2 * NK1 + 12;
    const int half = l >> 5;
    // ---- layer 1 of both networks
    frag_t w1p_hi[2 * NK1], w1p_lo[2 * NK1], w1v_hi[2 * NK1], w1v_lo[2 * NK1];
    load_frags<SPLIT, 2 * NK1>(FP, nfw, 0, l, w1p_hi, w1p_lo);
    load_frags<SPLIT, 2 * NK1>(FV, nfw, 0, l, w1v_hi, w1v_lo);
    f32x16 p0 = bias_tile(bP, 0, half), p1 = bias_tile(bP, 1, half), v0 = bias_tile(bV, 0, half), v1 = bias_tile(bV, 1, half);
    frag_t w2_hi[8], w2_lo[8];
    load_frags<SPLIT, 8>(FP, nfw, 2 * NK1, l, w2_hi, w2_lo);   // (layer 2 of pi: lands under the layer-1 MFMAs)
#pragma unroll
    for (int kk = 0; kk < NK1; ++kk) mma3x2<SPLIT>(w1p_hi[kk], w1p_lo[kk], w1p_hi[NK1 + kk], w1p_lo[NK1 + kk], bx_hi[kk], bx_lo[kk], p0, p1);
#pragma unroll
    for (int kk = 0; kk < NK1; ++kk) mma3x2<SPLIT>(w1v_hi[kk], w1v_lo[kk], w1v_hi[NK1 + kk], w1v_lo[NK1 + kk], bx_hi[kk], bx_lo[kk], v0, v1);
    // ---- pi: hidden 1 -> B (vector units; the vf layer-1 MFMAs above are still in the pipe)
    frag_t bp_hi[4], bp_lo[4], bv_hi[4], bv_lo[4];
    FWG_ATL(A, 8);
    hidden_to_b(p0, p1, bp_hi, bp_lo);
    FWG_SCHED_FENCE();
    FWG_ATL(A, 9);
    // ---- pi layer 2 (matrix pipe) || vf: hidden 1 -> B (vector units), one k-block of each per round: the round is its own
    // scheduling region, so the {1 MFMA, n VALU} groups below are met exactly
    f32x16 h0 = bias_tile(bP + 64, 0, half), h1 = bias_tile(bP + 64, 1, half);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        mma3x2<SPLIT>(w2_hi[kk], w2_lo[kk], w2_hi[4 + kk], w2_lo[4 + kk], bp_hi[kk], bp_lo[kk], h0, h1);
        hidden_block(kk < 2 ? v0 : v1, kk & 1, bv_hi[kk], bv_lo[kk]);
        FWG_SCHED_MFMA_VALU(SPLIT > 1 ? 6 : 2, SPLIT > 1 ? 8 : 24);
        FWG_SCHED_FENCE();
    }
    p0 = h0; p1 = h1;
    FWG_ATL(A, 10);
    // ---- vf layer 2 || pi: hidden 2 -> B
    load_frags<SPLIT, 8>(FV, nfw, 2 * NK1, l, w2_hi, w2_lo);
    h0 = bias_tile(bV + 64, 0, half); h1 = bias_tile(bV + 64, 1, half);
    FWG_SCHED_FENCE();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        mma3x2<SPLIT>(w2_hi[kk], w2_lo[kk], w2_hi[4 + kk], w2_lo[4 + kk], bv_hi[kk], bv_lo[kk], h0, h1);
        hidden_block(kk < 2 ? p0 : p1, kk & 1, bp_hi[kk], bp_lo[kk]);
        FWG_SCHED_MFMA_VALU(SPLIT > 1 ? 6 : 2, SPLIT > 1 ? 8 : 24);
        FWG_SCHED_FENCE();
    }
    v0 = h0; v1 = h1;
    FWG_ATL(A, 11);
    // ---- pi layer 3 || vf: hidden 2 -> B
    frag_t w3_hi[4], w3_lo[4];
    load_frags<SPLIT, 4>(FP, nfw, 2 * NK1 + 8, l, w3_hi, w3_lo);
    o_pi = bias_tile(bP + 128, 0, half);
    f32x16 oa = {0.f}, ob = {0.f};
    FWG_SCHED_FENCE();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        mma3s<SPLIT>(w3_hi[kk], w3_lo[kk], bp_hi[kk], bp_lo[kk], o_pi, oa, ob);
        hidden_block(kk < 2 ? v0 : v1, kk & 1, bv_hi[kk], bv_lo[kk]);
        FWG_SCHED_MFMA_VALU(SPLIT > 1 ? 3 : 1, SPLIT > 1 ? 16 : 48);
        FWG_SCHED_FENCE();
    }
    FWG_ATL(A, 12);
    // ---- vf layer 3
    load_frags<SPLIT, 4>(FV, nfw, 2 * NK1 + 8, l, w3_hi, w3_lo);
    o_vf = bias_tile(bV + 128, 0, half);
    f32x16 va = {0.f}, vb = {0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) mma3s<SPLIT>(w3_hi[kk], w3_lo[kk], bv_hi[kk], bv_lo[kk], o_vf, va, vb);
    if (SPLIT > 1) {   // (only rows 0..3 are outputs)
#pragma unroll
        for (int i = 0; i < 4; ++i) { o_pi[i] += oa[i] + ob[i]; o_vf[i] += va[i] + vb[i]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// batch moments of an observation / reward batch that did not come out of an attached env step (the first observation
// after reset(), evaluation loops): one thread = one environment, same accumulation as the tail of k_step
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FWG_ACT_BLOCK) void k_actor_stats(const ActorArgs A) {
    const int tid = threadIdx.x, lane = tid & 63;
    const long e = (long)blockIdx.x * FWG_ACT_BLOCK + tid;
    const bool valid = e < A.N;
    const long wave = e >> 6;
    if ((wave << 6) >= A.N) return;   // whole wave out of range (wave-uniform)
    const ActorStats& S = A.stats[A.parity];
    const int D = A.D;
    const bool has_obs = A.obs != nullptr, has_ret = A.rew != nullptr;
    float dr = 0.f;
    if (has_ret && valid) {   // VecNormalize.step_wait: ret = ret * gamma + r; ret_rms.update(ret); ret[done] = 0
        const float r = A.ret[e] * A.gamma + A.rew[e];
        A.ret[e] = (A.done != nullptr && A.done[e]) ? 0.f : r;
        dr = r - S.ret_mean;
    }
    unsigned long long* acc = A.acc + (size_t)(S.act_counter % FWG_ACC_SETS) * FWG_ACC_SHARDS * A.acc_cols;
    const long long win = actor_obs_win(A);
    const long er = valid ? e : 0;
#define FWG_OBS_AT(k) (actor_obs_at(A, win, er, (k) & ~3)[(k) & 3])
    for (int chunk = 0; 32 * chunk < 2 * D + 4; ++chunk) {
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = FWG_ACC_COLUMN(32 * chunk + i, D, valid, FWG_OBS_AT, S.mean, dr, has_obs, has_ret);
        acc_flush(acc, A.acc_cols, (int)(wave & (FWG_ACC_SHARDS - 1)), chunk, lane, wave_totals32(v, lane));
    }
#undef FWG_OBS_AT
}

// ---------------------------------------------------------------------------------------------------------------------
// statistics update + normalisation + pi/vf forward + sampling for the 256 environments [env_first, env_first + 256) of one
// 8-wave workgroup (two waves per SIMD, so that one wave's MFMAs overlap the other's tanh/conversion work; a wave = one
// 32-environment tile); the packed weights of both networks sit in LDS once per workgroup (76 KiB for obs_dim <= 16).
// Shared by k_actor_act (the head as a launch of its own) and k_rollout (fwgym.hip: the head AND the env step that takes its
// actions in one launch -- `act_out` then receives the sampled actions, [256][4] floats in LDS).
// ---------------------------------------------------------------------------------------------------------------------
struct ActorLds {
    frag_t* F;        // [net][part][nf][64] packed weights
    float* mean_s;    // [64] updated running mean
    float* rstd_s;    // [64] 1 / sqrt(updated running variance + eps)
    float* misc;      // [0] 1 / sqrt(ret_var + eps), [1] updated running mean of the returns, [4 ...] batch sums
    float* act_out;   // nullable: sampled actions of the workgroup's environments, [256][4]
    float* bias;      // [2][FWG_ACT_BIAS_FLOATS] the networks' biases
};
__host__ __device__ inline int actor_weight_floats(int nk1, int parts) { return 2 * parts * actor_frags(nk1) * 64 * 4; }
__host__ __device__ inline int actor_scratch_floats() { return 2 * FWG_ACT_MAX_OBS + 4 + 2 * FWG_ACT_MAX_OBS + 4 + 2 * FWG_ACT_BIAS_FLOATS; }

template <int SPLIT, int NK1>
__device__ __forceinline__ unsigned actor_block(const ActorArgs& A, const ActorLds& Z, long env_first) {
    const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6, j = l & 31, half = l >> 5;
    constexpr int nf = 2 * NK1 + 12;
    constexpr int PARTS = SPLIT > 1 ? 2 : 1;
    const bool first_block = env_first == 0;
    FWG_ATL(A, 0);
    frag_t* F = Z.F;
    float* mean_s = Z.mean_s;
    float* rstd_s = Z.rstd_s;
    float* misc = Z.misc;
    // packed weights L2 -> registers now, -> LDS after the statistics are folded (below): every wave moves its share of the 1 KiB
    // pieces, 16 bytes per lane and piece.  (Round 3 streamed them with global_load_lds: the LDS-DMA path delivers ~25 GB/s per
    // CU whatever the number of waves issuing -- 56 KiB landed 5k ticks after the request, 1.7k of them exposed; plain loads
    // come in at the L1 rate and the ds_write_b128 cost 13 cycles each)
    constexpr int n_pieces = PARTS == 2 ? 2 * 2 * nf : 2 * nf;
    constexpr int per_wave = (n_pieces + FWG_ACT_WAVES - 1) / FWG_ACT_WAVES;
    float4 wreg[per_wave];
#ifndef FWG_ABL_ACT_NO_STAGE
#pragma unroll
    for (int i = 0; i < per_wave; ++i) {
        const int q = wv + i * FWG_ACT_WAVES;
        wreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < n_pieces) {
            // split products: source and destination have the same layout [net][part][nf]; single products: only the hi parts
            const int src = PARTS == 2 ? q : ((q / nf) * 2 * nf + (q % nf));
            wreg[i] = reinterpret_cast<const float4*>(A.frags + src * 64)[l];
        }
    }
#endif
    // raw observation entries of this lane (k-slots of the first layer), requested before the statistics are folded
    const long e = env_first + wv * 32 + j;
    const bool valid = e < A.N;
    const bool vec4 = (A.D & 3) == 0 && (A.obs_n & 3) == 0;
    const long long win = actor_obs_win(A);
    float raw_x[NK1][8];
#pragma unroll
    for (int kk = 0; kk < NK1; ++kk) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f0 = k_input(kk, half, 4 * q);
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (valid && f0 < A.D) {
                if (vec4) {
                    const float4 g = *reinterpret_cast<const float4*>(actor_obs_at(A, win, e, f0));
                    v[0] = g.x; v[1] = g.y; v[2] = g.z; v[3] = g.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (f0 + i < A.D) v[i] = A.obs[e * A.D + f0 + i];   // dense only (log: obs_n % 4 == 0)
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) raw_x[kk][4 * q + i] = v[i];
        }
    }
    // what the sampling / output phase reads is requested here as well, BEFORE this kernel's first store: a load issued after
    // the normalised-observation stores waits for their acknowledgement (vmcnt counts in issue order), and a uniform load
    // after a store is no longer a scalar load
    const unsigned act_ctr = A.stats[A.parity].act_counter;
    const float bias_in = tid < 2 * FWG_ACT_BIAS_FLOATS ? A.bias[tid] : 0.f;   // (parked in LDS with the batch sums below)
    float ls_in[FWG_ACT_MAX_ACT];
#pragma unroll
    for (int i = 0; i < FWG_ACT_MAX_ACT; ++i) ls_in[i] = i < A.act_dim ? A.log_std[i] : 0.f;
    const float rew_in = (A.rew != nullptr && valid) ? A.rew[e] : 0.f;
    const uint8_t done_in = (A.done != nullptr && valid) ? A.done[e] : (uint8_t)0;
    FWG_ATL(A, 1);
    {   // add the accumulator shards (integers: exact, order-free) and fold the batch into the running statistics (the
        // parallel-variance update of VecNormalize's RunningMeanStd).  Every block computes the same values; the first
        // block publishes them and clears the accumulator set the producers after the NEXT act will add into
        const ActorStats& S0 = A.stats[A.parity];
        ActorStats& S1 = A.stats[A.parity ^ 1];
        const int D = A.D, cols = 2 * D + 4;
        const size_t set_words = (size_t)FWG_ACC_SHARDS * A.acc_cols;
        const unsigned long long* acc = A.acc + (size_t)(act_ctr % FWG_ACC_SETS) * set_words;
        float* tot = misc + 4;   // [cols] batch sums, column layout of fwgym_env.h
        if (tid < cols) {
            long long sum = 0;
#pragma unroll
            for (int sh = 0; sh < FWG_ACC_SHARDS; ++sh) sum += (long long)acc[(long)sh * A.acc_cols + tid];
            tot[tid] = (tid == 2 || tid == 3) ? (float)sum : (float)sum * (1.f / FWG_ACC_SCALE);
        }
        if (tid < 2 * FWG_ACT_BIAS_FLOATS) Z.bias[tid] = bias_in;
        // (LDS-only barriers here: __syncthreads() drains vmcnt, i.e. it would wait for the whole weight staging above --
        // the fold and the normalisation below are meant to run UNDER it; dma_wait() before the first MFMA is the drain)
        FWG_BLOCK_SYNC_LDS();
        const float n_obs = A.training ? tot[2] : 0.f, n_ret = A.training ? tot[3] : 0.f;
        if (tid < FWG_ACT_MAX_OBS) {
            const int f = tid;
            float m = 0.f, rs = 0.f, v = 1.f;
            if (f < D) {
                m = S0.mean[f]; v = S0.var[f];
                if (n_obs > 0.f) {
                    const float cnt = S0.count, tt = cnt + n_obs;
                    const float s1 = tot[4 + 2 * f] / n_obs, s2 = tot[5 + 2 * f] / n_obs;
                    const float bvar = fmaxf(s2 - s1 * s1, 0.f);
                    const float mm = v * cnt + bvar * n_obs + s1 * s1 * (cnt * n_obs / tt);
                    m += s1 * (n_obs / tt);
                    v = mm / tt;
                }
                rs = 1.f / sqrtf(v + A.eps);
            }
            mean_s[f] = m; rstd_s[f] = rs;
            if (first_block) { S1.mean[f] = m; S1.var[f] = v; }
        }
        if (tid == FWG_ACT_MAX_OBS) {
            float rm = S0.ret_mean, rv = S0.ret_var, rc = S0.ret_count;
            if (n_ret > 0.f) {
                const float tt = rc + n_ret;
                const float s1 = tot[0] / n_ret, s2 = tot[1] / n_ret;
                const float bvar = fmaxf(s2 - s1 * s1, 0.f);
                const float mm = rv * rc + bvar * n_ret + s1 * s1 * (rc * n_ret / tt);
                rm += s1 * (n_ret / tt);
                rv = mm / tt;
                rc = tt;
            }
            misc[0] = 1.f / sqrtf(rv + A.eps);
            misc[1] = rm;
            if (first_block) {
                S1.count = S0.count + n_obs; S1.ret_mean = rm; S1.ret_var = rv; S1.ret_count = rc;
                S1.act_counter = act_ctr + 1u;
            }
        }
        if (first_block) {
            unsigned long long* nxt = A.acc + (size_t)((act_ctr + 2u) % FWG_ACC_SETS) * set_words;
            for (int i = tid; i < FWG_ACC_SHARDS * A.acc_cols; i += 64 * FWG_ACT_WAVES) nxt[i] = 0ull;
        }
    }
    FWG_BLOCK_SYNC_LDS();
    FWG_ATL(A, 2);

    frag_t bx_hi[NK1], bx_lo[NK1];
#pragma unroll
    for (int kk = 0; kk < NK1; ++kk) {
        float x[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f0 = k_input(kk, half, 4 * q);
            float v[4] = {raw_x[kk][4 * q], raw_x[kk][4 * q + 1], raw_x[kk][4 * q + 2], raw_x[kk][4 * q + 3]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // features >= D: mean 0, rstd 0 -> 0
                v[i] = fminf(fmaxf((v[i] - mean_s[f0 + i]) * rstd_s[f0 + i], -A.clip_obs), A.clip_obs);
                x[4 * q + i] = v[i];
            }
            if (A.norm_obs != nullptr && valid && f0 < A.D) {
                if (vec4) *reinterpret_cast<float4*>(A.norm_obs + e * A.D + f0) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (f0 + i < A.D) A.norm_obs[e * A.D + f0 + i] = v[i];
                }
            }
        }
        split8(x, bx_hi[kk], bx_lo[kk]);
    }
    FWG_ATL(A, 3);
#ifndef FWG_ABL_ACT_NO_STAGE
#pragma unroll
    for (int i = 0; i < per_wave; ++i) {
        const int q = wv + i * FWG_ACT_WAVES;
        if (q < n_pieces) reinterpret_cast<float4*>(F + q * 64)[l] = wreg[i];
    }
#endif
    // the sampling noise does not depend on the networks: drawn here (Philox + Box-Muller + exp(log_std)), under the wait for the
    // other waves' weight pieces, instead of after the MLP at the end of the head phase
    float noise_n[4] = {0.f, 0.f, 0.f, 0.f}, noise_s[FWG_ACT_MAX_ACT] = {0.f, 0.f, 0.f, 0.f}, lp = 0.f;
    if (half == 0) {
        if (!A.deterministic) {
            const u4 b = philox4x32((unsigned)(A.env_base + e), act_ctr, 0u, FWG_STREAM_POLICY, A.seed_lo, A.seed_hi);
            box_muller(b, noise_n);
        }
#pragma unroll
        for (int i = 0; i < FWG_ACT_MAX_ACT; ++i) {
            if (i < A.act_dim) {
                const float ls = ls_in[i];
                noise_s[i] = expf(ls) * noise_n[i];
                lp += -0.5f * noise_n[i] * noise_n[i] - ls - 0.9189385332046727f;
            }
        }
    }
    FWG_BLOCK_SYNC_LDS();
    FWG_ATL(A, 4);
    // both networks in one instruction stream: their MFMA chains and tanh phases are independent and interleave
#ifdef FWG_ABL_ACT_NO_MLP   // measurement only (tools/ablate.py)
    f32x16 o_pi = {0.f}, o_vf = {0.f};
    o_pi[0] = __uint_as_float(bx_hi[0].x ^ F[l].x); o_vf[0] = __uint_as_float(bx_lo[0].y);
#else
    f32x16 o_pi, o_vf;
    // (a static s_setprio 1 for the second-dispatched half of the waves around this call: measured, no effect)
    mlp_pair<SPLIT, NK1>(A, F, F + PARTS * nf * 64, Z.bias, Z.bias + FWG_ACT_BIAS_FLOATS, l, bx_hi, bx_lo, o_pi, o_vf);
#endif
    FWG_ATL(A, 5);
    const float res[2][FWG_ACT_MAX_ACT] = {{o_pi[0], o_pi[1], o_pi[2], o_pi[3]}, {o_vf[0], o_vf[1], o_vf[2], o_vf[3]}};
    if (half == 0) {
        float act[FWG_ACT_MAX_ACT] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < FWG_ACT_MAX_ACT; ++i) {
            if (i < A.act_dim) {
                act[i] = res[0][i] + noise_s[i];
                if (A.action != nullptr && valid) A.action[e * A.act_dim + i] = act[i];
            }
        }
        // (k_rollout) the env step of this same launch takes the actions from here
        if (Z.act_out != nullptr) *reinterpret_cast<float4*>(Z.act_out + (wv * 32 + j) * 4) = make_float4(act[0], act[1], act[2], act[3]);
        if (valid) {
            if (A.logp != nullptr) A.logp[e] = lp;
            if (A.value != nullptr) A.value[e] = res[1][0];
            if (A.norm_rew != nullptr && A.rew != nullptr)
                A.norm_rew[e] = fminf(fmaxf(rew_in * misc[0], -A.clip_rew), A.clip_rew);
            if (A.done_out != nullptr && A.done != nullptr) A.done_out[e] = done_in;
        }
    }
    FWG_ATL(A, 6);
    return act_ctr;
}

template <int SPLIT, int NK1>
__global__ __launch_bounds__(64 * FWG_ACT_WAVES) void k_actor_act(const ActorArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PARTS = SPLIT > 1 ? 2 : 1;
    ActorLds Z;
    Z.F = reinterpret_cast<frag_t*>(lds);
    Z.mean_s = lds + actor_weight_floats(NK1, PARTS);
    Z.rstd_s = Z.mean_s + FWG_ACT_MAX_OBS;
    Z.misc = Z.rstd_s + FWG_ACT_MAX_OBS;
    Z.act_out = nullptr;
    Z.bias = Z.misc + 4 + 2 * FWG_ACT_MAX_OBS + 4;
    actor_block<SPLIT, NK1>(A, Z, (long)blockIdx.x * FWG_ACT_ENVS);
}

// ---------------------------------------------------------------------------------------------------------------------
// host: weight packing into MFMA operand fragments
// ---------------------------------------------------------------------------------------------------------------------
// one layer: W [out][in] row-major; nit row tiles, nk k-blocks -> hi/lo words appended (biases: actor_pack_bias)
static void actor_pack_layer(std::vector<unsigned>& hi, std::vector<unsigned>& lo, const float* W, int out,
                             int in, int nit, int nk, bool chained, float scale) {
    for (int it = 0; it < nit; ++it)
        for (int kk = 0; kk < nk; ++kk)
            for (int l = 0; l < 64; ++l) {
                unsigned wh[4] = {0, 0, 0, 0}, wl[4] = {0, 0, 0, 0};
                const int i = 32 * it + (l & 31), half = l >> 5;
                for (int t = 0; t < 8; ++t) {
                    float v = 0.f;
                    const int k = chained ? k_chained(kk, half, t) : k_input(kk, half, t);
                    if (i < out && k < in) v = scale * W[(size_t)i * in + k];
                    const unsigned h16 = bf16_rne(v), l16 = bf16_rne(v - bf16_to_f32(h16));
                    wh[t >> 1] |= h16 << (16 * (t & 1));
                    wl[t >> 1] |= l16 << (16 * (t & 1));
                }
                hi.insert(hi.end(), wh, wh + 4);
                lo.insert(lo.end(), wl, wl + 4);
            }
}
// the three bias vectors of one network: 64 | 64 | 32 floats (hidden layers pre-scaled like their weights)
static void actor_pack_bias(std::vector<float>& out, const float* b0, const float* b1, const float* b2, int n_out) {
    for (int i = 0; i < 64; ++i) out.push_back(FWG_ACT_PRESCALE * b0[i]);
    for (int i = 0; i < 64; ++i) out.push_back(FWG_ACT_PRESCALE * b1[i]);
    for (int i = 0; i < 32; ++i) out.push_back(i < n_out ? b2[i] : 0.f);
}
