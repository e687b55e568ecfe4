// TEST-ONLY host emulation of the tiny slice of the HIP programming model that csrc/fwgym.hip uses, so that the kernel
// LOGIC (not its performance) can be exercised by `pytest -m "not gpu"` in a container without a GPU:
//   g++ -x c++ -I tests/emu ... csrc/fwgym.hip -> tests/emu/libfwgym_emu.so
// One workgroup = 64 std::threads in lock-step at __syncthreads()/__ballot()/__shfl_xor(); "device memory" is host
// memory; global_load_lds is an immediate copy.  Never loaded by the product (gym_fixed_wing/_native.py only loads
// libfwgym.so); see DESIGN.md "Testing without a GPU".
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <ucontext.h>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define ext_vector_type(n) vector_size(4 * (n))
#define __shared__
#define FWG_DMA_DRAIN() ((void)0)
#define FWG_EMU 1
#ifdef FWG_EMU_OMP   /* bench-only CPU baseline: workgroups spread over OpenMP threads, per-thread emulation state */
#include <omp.h>
#define EMU_TLS thread_local
#else
#define EMU_TLS
#endif

struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct emu_idx { unsigned x, y, z; };
static EMU_TLS emu_idx threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n); return 0; }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }

// ---- block-wide lock-step primitives: the 64 lanes of a workgroup are fibers (ucontext) of ONE thread; a lane runs
// until it reaches a barrier (or returns), then the next lane runs; a full round-robin pass = one barrier phase ----
struct emu_lane { ucontext_t ctx; char* stack; int state; };  // state: 0 runnable, 1 at barrier, 2 finished
static EMU_TLS emu_lane emu_lanes[1024];
static EMU_TLS ucontext_t emu_sched_ctx;
static EMU_TLS unsigned emu_cur = 0;
static EMU_TLS uint32_t emu_xchg[1024];
alignas(16) EMU_TLS float lds[160 * 1024 / 4];  // the block's dynamic LDS (`extern __shared__ float lds[]` in the kernels)
static emu_idx emu_block_idx, emu_block_dim;

// Barriers: a lane that arrives yields until everybody of its scope (its 64-lane wave, or the workgroup) has arrived
// -- lanes that already returned from the kernel do not count.  Generation counters make the barriers reusable.
static EMU_TLS unsigned emu_nwaves = 1;
static EMU_TLS unsigned emu_wave_live[16], emu_wave_wait[16], emu_wave_gen[16];
static EMU_TLS unsigned emu_block_live, emu_block_wait, emu_block_gen;
static inline void emu_yield() {
    emu_lanes[emu_cur].state = 1;
    swapcontext(&emu_lanes[emu_cur].ctx, &emu_sched_ctx);
}
static inline void emu_wave_sync() {
    const unsigned w = emu_cur >> 6;
    const unsigned my = emu_wave_gen[w];
    if (++emu_wave_wait[w] >= emu_wave_live[w]) { emu_wave_wait[w] = 0; ++emu_wave_gen[w]; return; }
    while (emu_wave_gen[w] == my) emu_yield();
}
static inline void __syncthreads() {
    const unsigned my = emu_block_gen;
    if (++emu_block_wait >= emu_block_live) { emu_block_wait = 0; ++emu_block_gen; return; }
    while (emu_block_gen == my) emu_yield();
}
static inline void emu_lane_exit() {   // a finished lane releases barriers the others are already waiting at
    const unsigned w = emu_cur >> 6;
    --emu_wave_live[w];
    --emu_block_live;
    if (emu_wave_live[w] > 0 && emu_wave_wait[w] >= emu_wave_live[w]) { emu_wave_wait[w] = 0; ++emu_wave_gen[w]; }
    if (emu_block_live > 0 && emu_block_wait >= emu_block_live) { emu_block_wait = 0; ++emu_block_gen; }
}
// wave-scope collectives (64 lanes): exchange through a per-lane slot, two wave barriers around the read
static inline unsigned long long __ballot(int pred) {
    emu_xchg[threadIdx.x] = pred ? 1u : 0u;
    emu_wave_sync();
    const unsigned w0 = threadIdx.x & ~63u;
    unsigned long long m = 0;
    for (unsigned i = 0; i < 64 && w0 + i < blockDim.x; ++i) m |= (unsigned long long)emu_xchg[w0 + i] << i;
    emu_wave_sync();
    return m;
}
static inline float __shfl_xor(float v, int mask, int) {
    memcpy(&emu_xchg[threadIdx.x], &v, 4);
    emu_wave_sync();
    float r;
    memcpy(&r, &emu_xchg[threadIdx.x ^ (unsigned)mask], 4);
    emu_wave_sync();
    return r;
}
static std::mutex emu_atomic_mutex;
static inline float atomicAdd(float* p, float v) { std::lock_guard<std::mutex> g(emu_atomic_mutex); float o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { std::lock_guard<std::mutex> g(emu_atomic_mutex); unsigned long long o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { std::lock_guard<std::mutex> g(emu_atomic_mutex); unsigned o = *p; *p = o + v; return o; }
static inline unsigned long long atomicExch(unsigned long long* p, unsigned long long v) { std::lock_guard<std::mutex> g(emu_atomic_mutex); unsigned long long o = *p; *p = v; return o; }
static inline void __threadfence() {}
static inline int atomicOr(int* p, int v) { std::lock_guard<std::mutex> g(emu_atomic_mutex); int o = *p; *p = o | v; return o; }

// ---- intrinsics ----------------------------------------------------------------------------------------------------
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
#define __expf(x) expf(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
#define __builtin_amdgcn_logf(x) log2f(x)
#define __builtin_amdgcn_sinf(x) sinf(6.28318530717958647692f * (x))
#define __builtin_amdgcn_cosf(x) cosf(6.28318530717958647692f * (x))
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) \
    memcpy((char*)(l) + (threadIdx.x & 63u) * (size), (const void*)(g), (size))
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
using std::max;
using std::min;

// v_mfma_f32_32x32x16_bf16 for the 64-lane wave the calling lane belongs to: lane l supplies A[l & 31][8 (l >> 5) + t]
// and B[8 (l >> 5) + t][l & 31], t = 0..7 (bf16), and receives D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31], r = 0..15
static EMU_TLS unsigned short emu_mfma_a[1024][8], emu_mfma_b[1024][8];
static inline float emu_bf16(unsigned short h) { const unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
template <class FRAG, class ACC>
static inline ACC emu_mfma_f32_32x32x16_bf16(const FRAG& a, const FRAG& b, ACC c) {
    static_assert(sizeof(FRAG) == 16, "operand = 8 bf16");
    const unsigned tid = threadIdx.x;
    memcpy(emu_mfma_a[tid], &a, 16);
    memcpy(emu_mfma_b[tid], &b, 16);
    emu_wave_sync();
    const unsigned w0 = tid & ~63u, l = tid & 63u, j = l & 31u, half = l >> 5;
    for (unsigned r = 0; r < 16; ++r) {
        const unsigned i = (r & 3u) + 8u * (r >> 2) + 4u * half;
        float s = 0.f;
        for (unsigned k = 0; k < 16; ++k)
            s += emu_bf16(emu_mfma_a[w0 + i + 32u * (k >> 3)][k & 7u]) * emu_bf16(emu_mfma_b[w0 + j + 32u * (k >> 3)][k & 7u]);
        c[r] += s;
    }
    emu_wave_sync();
    return c;
}

static EMU_TLS const std::function<void()>* emu_body;
static void emu_trampoline() {
    (*emu_body)();
    emu_lane_exit();
    emu_lanes[emu_cur].state = 2;
    swapcontext(&emu_lanes[emu_cur].ctx, &emu_sched_ctx);
}
static void emu_run_block(const std::function<void()>& body, unsigned b, dim3 block, dim3 grid) {
    static const size_t STACK = 2u << 20;
    emu_body = &body;
    for (unsigned t = 0; t < block.x; ++t) {
        emu_lane& L = emu_lanes[t];
        if (!L.stack) L.stack = (char*)malloc(STACK);
        getcontext(&L.ctx);
        L.ctx.uc_stack.ss_sp = L.stack;
        L.ctx.uc_stack.ss_size = STACK;
        L.ctx.uc_link = &emu_sched_ctx;
        L.state = 0;
        makecontext(&L.ctx, emu_trampoline, 0);
    }
    emu_nwaves = (block.x + 63) / 64;
    for (unsigned w = 0; w < emu_nwaves; ++w) {
        const unsigned first = w * 64, last = first + 64 < block.x ? first + 64 : block.x;
        emu_wave_live[w] = last - first; emu_wave_wait[w] = 0;
    }
    emu_block_live = block.x; emu_block_wait = 0;
    // Wave order of a scheduling pass (FWG_EMU_ORDER, read once): 0 = waves in index order (default); 1 = reverse order (the
    // gym wave of a two-wave step runs until it needs its partner BEFORE the physics wave starts); r<seed> = a pseudo-random
    // order per pass.  The hardware promises no order between the waves of a workgroup: what one wave may only read before /
    // after its partner writes must be ordered by a message, a mark or a barrier, and a protocol that merely happens to hold in
    // index order fails under another one (tests/test_emu_coverage.py runs the steady-state test under all three).
    const char* order_env = getenv("FWG_EMU_ORDER");   // (per block: a test may change it between launches)
    const int mode = order_env == nullptr ? 0 : (order_env[0] == 'r' ? 2 : atoi(order_env));
    unsigned long long lcg = 0x9E3779B97F4A7C15ull * (b + 1) + (mode == 2 ? strtoull(order_env + 1, nullptr, 10) : 0ull);
    for (;;) {
        unsigned alive = 0;
        unsigned worder[16];
        for (unsigned w = 0; w < emu_nwaves; ++w) worder[w] = mode == 1 ? emu_nwaves - 1 - w : w;
        if (mode == 2)
            for (unsigned w = emu_nwaves; w > 1; --w) {
                lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                const unsigned k = (unsigned)((lcg >> 33) % w), tmp = worder[w - 1];
                worder[w - 1] = worder[k]; worder[k] = tmp;
            }
        for (unsigned wi = 0; wi < emu_nwaves; ++wi)
            for (unsigned t = worder[wi] * 64; t < worder[wi] * 64 + 64 && t < block.x; ++t) {
                if (emu_lanes[t].state == 2) continue;
                ++alive;
                emu_cur = t;
                threadIdx = {t, 0, 0}; blockIdx = {b, 0, 0}; blockDim = {block.x, 1, 1}; gridDim = {grid.x, 1, 1};
                emu_lanes[t].state = 0;
                swapcontext(&emu_sched_ctx, &emu_lanes[t].ctx);
            }
        if (!alive) break;
    }
}
template <typename K, typename... Args>
static void emu_launch(K kernel, dim3 grid, dim3 block, Args... args) {
    const std::function<void()> body = [=]() { kernel(args...); };
#ifdef FWG_EMU_OMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int b = 0; b < (int)grid.x; ++b) emu_run_block(body, (unsigned)b, block, grid);
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(kernel, grid, block, __VA_ARGS__)
