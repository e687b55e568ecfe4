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
#include "fwgym_physics.h"

#define FWG_STREAM_POLICY 6u
#define FWG_ACT_BLOCK 256
#define FWG_ACT_MAX_OBS 64
#define FWG_ACT_MAX_ACT 4

// running statistics + the act counter (Philox counter of the sampling noise), double-buffered by parity: every
// k_actor_act reads copy [parity] and publishes copy [parity ^ 1], so captured launch sequences replay correctly
struct ActorStats { float mean[FWG_ACT_MAX_OBS], var[FWG_ACT_MAX_OBS]; float count, ret_mean, ret_var, ret_count; unsigned act_counter, pad_[3]; };
// batch moments about the running mean and the batch sizes they were taken over (0 = nothing observed)
struct ActorAcc { float s1[FWG_ACT_MAX_OBS], s2[FWG_ACT_MAX_OBS]; float r1, r2, n_obs, n_ret; };
#define FWG_ACT_NACC (2 * FWG_ACT_MAX_OBS + 4)

struct alignas(16) frag_t { unsigned x, y, z, w; };   // 8 bf16 = the A or B operand of one lane

struct ActorArgs {
    const float* obs; const float* rew; const uint8_t* done;
    float* ret;
    ActorStats* stats; ActorAcc* acc;   // [2] each, indexed by parity
    const frag_t* frags;                // [net 2][part hi/lo][frag][64 lanes]
    const float* log_std;
    float* norm_obs; float* action; float* value; float* logp; float* norm_rew; uint8_t* done_out;
    long N; long env_base;
    int D, nk1, act_dim, parity, training, deterministic;
    float gamma, clip_obs, clip_rew, eps;
    unsigned seed_lo, seed_hi;
};

// frags per network and part: layer 1: 2 row tiles x (nk1 + bias), layer 2: 2 x (4 + bias), layer 3: 1 x (4 + bias)
__host__ __device__ inline int actor_frags(int nk1) { return 2 * (nk1 + 1) + 2 * 5 + 5; }
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

// eight fp32 values -> bf16 hi and lo operands
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

__device__ __forceinline__ float fast_tanh(float x) {   // 1 - 2 / (e^{2x} + 1), absolute error ~1e-7
    const float t = fwg_exp2(2.885390081777927f * x);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
}

// acc += A[idx] * B over the split parts; F = this network's fragments in LDS [part][nf][64]
template <int SPLIT>
__device__ __forceinline__ f32x16 mma(const frag_t* F, int nf, int idx, int l, const frag_t& b_hi, const frag_t& b_lo, f32x16 acc) {
    const frag_t a_hi = F[idx * 64 + l];
    if (SPLIT > 1) {
        const frag_t a_lo = F[(nf + idx) * 64 + l];
        acc = fwg_mfma_bf16(a_lo, b_hi, acc);
        acc = fwg_mfma_bf16(a_hi, b_lo, acc);
    }
    return fwg_mfma_bf16(a_hi, b_hi, acc);
}

// acc += bias (the extra k-block against the constant ones operand)
template <int SPLIT>
__device__ __forceinline__ f32x16 mma_bias(const frag_t* F, int nf, int idx, int l, const frag_t& ones, f32x16 acc) {
    if (SPLIT > 1) acc = fwg_mfma_bf16(F[(nf + idx) * 64 + l], ones, acc);
    return fwg_mfma_bf16(F[idx * 64 + l], ones, acc);
}

// two accumulator tiles (64 hidden units) -> tanh -> the four k-blocks of the next layer's B operand
__device__ __forceinline__ void hidden_to_b(const f32x16& a0, const f32x16& a1, frag_t (&hi)[4], frag_t (&lo)[4]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        float x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = fast_tanh(kk < 2 ? a0[8 * kk + t] : a1[8 * (kk - 2) + t]);
        split8(x, hi[kk], lo[kk]);
    }
}

// one network (obs -> 64 -> 64 -> out) for the 32 environments of this half-wave tile; result rows 0..3 in out[0..3]
// of the lanes with half == 0
template <int SPLIT>
__device__ __forceinline__ f32x16 mlp_forward(const frag_t* F, int nf, int nk1, int l, const frag_t (&bx_hi)[4], const frag_t (&bx_lo)[4]) {
    const frag_t ones = frag_t{(l >> 5) == 0 ? 0x3F80u : 0u, 0u, 0u, 0u};   // 1.0 in k-slot (half 0, t 0)
    f32x16 a[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        f32x16 acc = {0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            if (kk < nk1) acc = mma<SPLIT>(F, nf, it * (nk1 + 1) + kk, l, bx_hi[kk], bx_lo[kk], acc);
        a[it] = mma_bias<SPLIT>(F, nf, it * (nk1 + 1) + nk1, l, ones, acc);
    }
    frag_t bh_hi[4], bh_lo[4];
    hidden_to_b(a[0], a[1], bh_hi, bh_lo);
    const int base2 = 2 * (nk1 + 1);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        f32x16 acc = {0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = mma<SPLIT>(F, nf, base2 + it * 5 + kk, l, bh_hi[kk], bh_lo[kk], acc);
        a[it] = mma_bias<SPLIT>(F, nf, base2 + it * 5 + 4, l, ones, acc);
    }
    hidden_to_b(a[0], a[1], bh_hi, bh_lo);
    const int base3 = base2 + 10;
    f32x16 out = {0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) out = mma<SPLIT>(F, nf, base3 + kk, l, bh_hi[kk], bh_lo[kk], out);
    return mma_bias<SPLIT>(F, nf, base3 + 4, l, ones, out);
}

__device__ __forceinline__ float actor_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// batch moments about the running mean -> acc[parity]; discounted returns
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FWG_ACT_BLOCK) void k_actor_stats(const ActorArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const long e = (long)blockIdx.x * FWG_ACT_BLOCK + tid;
    const bool valid = e < A.N;
    const ActorStats& S = A.stats[A.parity];
    for (int i = tid; i < FWG_ACT_NACC; i += FWG_ACT_BLOCK) lds[i] = 0.f;
    __syncthreads();
    if (A.obs != nullptr) {
        for (int f0 = 0; f0 < A.D; f0 += 4) {
            float x[4] = {0.f, 0.f, 0.f, 0.f};
            if (valid) {
                if ((A.D & 3) == 0) {
                    const float4 q = *reinterpret_cast<const float4*>(A.obs + e * A.D + f0);
                    x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (f0 + i < A.D) x[i] = A.obs[e * A.D + f0 + i];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (f0 + i < A.D) {   // wave-uniform
                    const float d = valid ? x[i] - S.mean[f0 + i] : 0.f;
                    const float s1 = actor_wave_sum(d), s2 = actor_wave_sum(d * d);
                    if ((tid & 63) == 0) { atomicAdd(&lds[f0 + i], s1); atomicAdd(&lds[FWG_ACT_MAX_OBS + f0 + i], s2); }
                }
            }
        }
    }
    if (A.rew != nullptr) {   // VecNormalize.step_wait: ret = ret * gamma + r; ret_rms.update(ret); ret[done] = 0
        float d = 0.f;
        if (valid) {
            const float r = A.ret[e] * A.gamma + A.rew[e];
            d = r - S.ret_mean;
            A.ret[e] = (A.done != nullptr && A.done[e]) ? 0.f : r;
        }
        const float s1 = actor_wave_sum(d), s2 = actor_wave_sum(d * d);
        if ((tid & 63) == 0) { atomicAdd(&lds[2 * FWG_ACT_MAX_OBS], s1); atomicAdd(&lds[2 * FWG_ACT_MAX_OBS + 1], s2); }
    }
    if (tid == 0) {   // batch sizes
        const long left = A.N - (long)blockIdx.x * FWG_ACT_BLOCK;
        const float cnt = (float)(left < FWG_ACT_BLOCK ? left : FWG_ACT_BLOCK);
        if (A.obs != nullptr) lds[2 * FWG_ACT_MAX_OBS + 2] = cnt;
        if (A.rew != nullptr) lds[2 * FWG_ACT_MAX_OBS + 3] = cnt;
    }
    __syncthreads();
    float* acc = reinterpret_cast<float*>(&A.acc[A.parity]);
    for (int i = tid; i < FWG_ACT_NACC; i += FWG_ACT_BLOCK)
        if (lds[i] != 0.f) atomicAdd(acc + i, lds[i]);
}

// ---------------------------------------------------------------------------------------------------------------------
// statistics update + normalisation + pi/vf forward + sampling.  Block = 4 waves, a wave = 64 envs in two 32-env tiles
// ---------------------------------------------------------------------------------------------------------------------
template <int SPLIT>
__global__ __launch_bounds__(FWG_ACT_BLOCK, 1) void k_actor_act(const ActorArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6, j = l & 31, half = l >> 5;
    const int nk1 = A.nk1, nf = actor_frags(nk1);
    constexpr int PARTS = SPLIT > 1 ? 2 : 1;
    frag_t* F = reinterpret_cast<frag_t*>(lds);                     // [net][part][nf][64]
    float* mean_s = lds + 2 * PARTS * nf * 64 * 4;                  // [64]
    float* rstd_s = mean_s + FWG_ACT_MAX_OBS;                       // [64]
    float* misc = rstd_s + FWG_ACT_MAX_OBS;                         // [0] = 1/sqrt(ret_var + eps)
    for (int net = 0; net < 2; ++net)
        for (int part = 0; part < PARTS; ++part)
            for (int i = tid; i < nf * 64; i += FWG_ACT_BLOCK)
                F[(net * PARTS + part) * nf * 64 + i] = A.frags[(net * 2 + part) * nf * 64 + i];
    {   // fold the accumulated batch moments into the running statistics (parallel-variance update of VecNormalize's
        // RunningMeanStd); every block computes the same values, block 0 publishes them for the next launch
        const ActorStats& S0 = A.stats[A.parity];
        ActorStats& S1 = A.stats[A.parity ^ 1];
        const ActorAcc& C = A.acc[A.parity];
        if (tid < FWG_ACT_MAX_OBS) {
            const float n = C.n_obs;
            const int f = tid;
            float m = 0.f, rs = 0.f, v = 1.f;
            if (f < A.D) {
                m = S0.mean[f]; v = S0.var[f];
                if (A.training && n > 0.f) {
                    const float cnt = S0.count, tot = cnt + n;
                    const float s1 = C.s1[f] / n, s2 = C.s2[f] / n;
                    const float bvar = fmaxf(s2 - s1 * s1, 0.f);
                    const float m2 = v * cnt + bvar * n + s1 * s1 * (cnt * n / tot);
                    m += s1 * (n / tot);
                    v = m2 / tot;
                }
                rs = 1.f / sqrtf(v + A.eps);
            }
            mean_s[f] = m; rstd_s[f] = rs;
            if (blockIdx.x == 0) { S1.mean[f] = m; S1.var[f] = v; }
        }
        if (tid == FWG_ACT_MAX_OBS) {
            float rm = S0.ret_mean, rv = S0.ret_var, rc = S0.ret_count, oc = S0.count;
            if (A.training) {
                oc += C.n_obs;
                const float n = C.n_ret;
                if (n > 0.f) {
                    const float tot = rc + n;
                    const float s1 = C.r1 / n, s2 = C.r2 / n;
                    const float bvar = fmaxf(s2 - s1 * s1, 0.f);
                    const float m2 = rv * rc + bvar * n + s1 * s1 * (rc * n / tot);
                    rm += s1 * (n / tot);
                    rv = m2 / tot;
                    rc = tot;
                }
            }
            misc[0] = 1.f / sqrtf(rv + A.eps);
            if (blockIdx.x == 0) {
                S1.count = oc; S1.ret_mean = rm; S1.ret_var = rv; S1.ret_count = rc;
                S1.act_counter = S0.act_counter + 1u;
            }
        }
        if (blockIdx.x == 0) {   // nobody reads the other parity's accumulators during this launch
            float* nxt = reinterpret_cast<float*>(&A.acc[A.parity ^ 1]);
            for (int i = tid; i < FWG_ACT_NACC; i += FWG_ACT_BLOCK) nxt[i] = 0.f;
        }
    }
    __syncthreads();

    const float log_2pi_half = 0.9189385332046727f;
#pragma unroll 1
    for (int tile = 0; tile < 2; ++tile) {
        const long e = (long)blockIdx.x * FWG_ACT_BLOCK + wv * 64 + tile * 32 + j;
        const bool valid = e < A.N;
        frag_t bx_hi[4], bx_lo[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < nk1) {
                float x[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int f0 = k_input(kk, half, 4 * q);
                    float v[4] = {0.f, 0.f, 0.f, 0.f};
                    if (valid && f0 < A.D) {
                        if ((A.D & 3) == 0) {
                            const float4 g = *reinterpret_cast<const float4*>(A.obs + e * A.D + f0);
                            v[0] = g.x; v[1] = g.y; v[2] = g.z; v[3] = g.w;
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) if (f0 + i < A.D) v[i] = A.obs[e * A.D + f0 + i];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int f = f0 + i;   // < 64
                        v[i] = fminf(fmaxf((v[i] - mean_s[f]) * rstd_s[f], -A.clip_obs), A.clip_obs);
                        x[4 * q + i] = v[i];
                    }
                    if (A.norm_obs != nullptr && valid && f0 < A.D) {
                        if ((A.D & 3) == 0) *reinterpret_cast<float4*>(A.norm_obs + e * A.D + f0) = make_float4(v[0], v[1], v[2], v[3]);
                        else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) if (f0 + i < A.D) A.norm_obs[e * A.D + f0 + i] = v[i];
                        }
                    }
                }
                split8(x, bx_hi[kk], bx_lo[kk]);
            }
        }
        const f32x16 pi = mlp_forward<SPLIT>(F, nf, nk1, l, bx_hi, bx_lo);
        const f32x16 vf = mlp_forward<SPLIT>(F + PARTS * nf * 64, nf, nk1, l, bx_hi, bx_lo);
        if (half == 0 && valid) {
            const float mean[FWG_ACT_MAX_ACT] = {pi[0], pi[1], pi[2], pi[3]};
            float n[4] = {0.f, 0.f, 0.f, 0.f};
            if (!A.deterministic) {
                const u4 b = philox4x32((unsigned)(A.env_base + e), A.stats[A.parity].act_counter, 0u, FWG_STREAM_POLICY,
                                        A.seed_lo, A.seed_hi);
                box_muller(b, n);
            }
            float lp = 0.f;
#pragma unroll
            for (int i = 0; i < FWG_ACT_MAX_ACT; ++i) {
                if (i < A.act_dim) {
                    const float ls = A.log_std[i];
                    if (A.action != nullptr) A.action[e * A.act_dim + i] = mean[i] + expf(ls) * n[i];
                    lp += -0.5f * n[i] * n[i] - ls - log_2pi_half;
                }
            }
            if (A.logp != nullptr) A.logp[e] = lp;
            if (A.value != nullptr) A.value[e] = vf[0];
            if (A.norm_rew != nullptr && A.rew != nullptr)
                A.norm_rew[e] = fminf(fmaxf(A.rew[e] * misc[0], -A.clip_rew), A.clip_rew);
            if (A.done_out != nullptr && A.done != nullptr) A.done_out[e] = A.done[e];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host: weight packing into MFMA operand fragments
// ---------------------------------------------------------------------------------------------------------------------
// one layer: W [out][in] row-major, b [out]; nit row tiles, nk k-blocks (+1 bias block) -> hi/lo words appended
static void actor_pack_layer(std::vector<unsigned>& hi, std::vector<unsigned>& lo, const float* W, const float* b, int out,
                             int in, int nit, int nk, bool chained) {
    for (int it = 0; it < nit; ++it)
        for (int kk = 0; kk <= nk; ++kk)
            for (int l = 0; l < 64; ++l) {
                unsigned wh[4] = {0, 0, 0, 0}, wl[4] = {0, 0, 0, 0};
                const int i = 32 * it + (l & 31), half = l >> 5;
                for (int t = 0; t < 8; ++t) {
                    float v = 0.f;
                    if (kk < nk) {
                        const int k = chained ? k_chained(kk, half, t) : k_input(kk, half, t);
                        if (i < out && k < in) v = W[(size_t)i * in + k];
                    } else if (half == 0 && t == 0 && i < out) {
                        v = b[i];
                    }
                    const unsigned h16 = bf16_rne(v), l16 = bf16_rne(v - bf16_to_f32(h16));
                    wh[t >> 1] |= h16 << (16 * (t & 1));
                    wl[t >> 1] |= l16 << (16 * (t & 1));
                }
                hi.insert(hi.end(), wh, wh + 4);
                lo.insert(lo.end(), wl, wl + 4);
            }
}
