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
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__
#define FWG_DMA_DRAIN() ((void)0)

struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct emu_idx { unsigned x, y, z; };
static thread_local emu_idx threadIdx, blockIdx, blockDim;

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

// ---- block-wide lock-step primitives -------------------------------------------------------------------------------
struct emu_barrier {
    std::mutex m; std::condition_variable cv; unsigned n = 0, count = 0, gen = 0;
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        const unsigned g = gen;
        if (++count == n) { count = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};
static emu_barrier emu_bar;
static uint32_t emu_xchg[1024];
alignas(16) float lds[64 * 1024 / 4];  // the block's dynamic LDS (`extern __shared__ float lds[]` in the kernels)

static inline void __syncthreads() { emu_bar.wait(); }
static inline unsigned long long __ballot(int pred) {
    emu_xchg[threadIdx.x] = pred ? 1u : 0u;
    emu_bar.wait();
    unsigned long long m = 0;
    for (unsigned i = 0; i < blockDim.x && i < 64; ++i) m |= (unsigned long long)emu_xchg[i] << i;
    emu_bar.wait();
    return m;
}
static inline float __shfl_xor(float v, int mask, int) {
    memcpy(&emu_xchg[threadIdx.x], &v, 4);
    emu_bar.wait();
    float r;
    memcpy(&r, &emu_xchg[threadIdx.x ^ (unsigned)mask], 4);
    emu_bar.wait();
    return r;
}
static std::mutex emu_atomic_mutex;
static inline float atomicAdd(float* p, float v) { std::lock_guard<std::mutex> g(emu_atomic_mutex); float o = *p; *p = o + v; return o; }
static inline int atomicOr(int* p, int v) { std::lock_guard<std::mutex> g(emu_atomic_mutex); int o = *p; *p = o | v; return o; }

// ---- intrinsics ----------------------------------------------------------------------------------------------------
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
#define __expf(x) expf(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) \
    (((float*)(l))[threadIdx.x] = *(const float*)(g))
using std::max;
using std::min;

template <typename K, typename... Args>
static void emu_launch(K kernel, dim3 grid, dim3 block, Args... args) {
    for (unsigned b = 0; b < grid.x; ++b) {
        emu_bar.n = block.x; emu_bar.count = 0;
        std::vector<std::thread> ts;
        for (unsigned t = 0; t < block.x; ++t)
            ts.emplace_back([=]() {
                threadIdx = {t, 0, 0}; blockIdx = {b, 0, 0}; blockDim = {block.x, 1, 1};
                kernel(args...);
            });
        for (auto& t : ts) t.join();
    }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(kernel, grid, block, __VA_ARGS__)
