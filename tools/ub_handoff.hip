// micro-benchmark: cost of handing 16 bytes per lane from one wave of a workgroup to another through LDS on gfx950, and the
// VALU issue rate of 1..4 co-resident waves per SIMD.  Behind DESIGN section 5 "team of waves": a function split of the
// right-hand side needs two such hand-offs per RK4 stage, so their latency decides whether it pays.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ub_handoff tools/ub_handoff.hip && /tmp/ub_handoff
#include <hip/hip_runtime.h>
#include <stdio.h>

#define ROUNDS 200

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f mk4(float a, float b, float c, float d) { v4f r = {a, b, c, d}; return r; }
__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(size_t)p; }

// MODE 0: data b128 + separate flag word (ds_write_b32 by lane 0), consumer polls the flag then reads the data
// MODE 1: tag inside the 16-byte group (4th word), consumer polls the data itself
// MODE 2: s_barrier (lgkmcnt(0) + s_barrier), both waves
// MODE 3: like 1, poll loop without s_sleep
template <int MODE>
__global__ __launch_bounds__(128, 2) void k_ping(long long* cyc, float* out, int work) {
    __shared__ __attribute__((aligned(16))) float buf[2][64 * 4];
    __shared__ unsigned flag[2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 2) flag[threadIdx.x] = 0u;
    for (int i = threadIdx.x; i < 2 * 64 * 4; i += 128) (&buf[0][0])[i] = 0.f;
    __syncthreads();
    float x = lane * 1e-3f + 1.f, y = 0.5f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 1; r <= ROUNDS; ++r) {
        // wave 0 sends round r on channel 0, wave 1 answers on channel 1
        for (int side = 0; side < 2; ++side) {
            if (wave == side) {
                // some dependent work before sending (so the compiler cannot collapse the loop)
                for (int w = 0; w < work; ++w) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
                if (MODE == 2) {
                    asm volatile("ds_write_b128 %0, %1" ::"v"(lds_off(&buf[side][lane * 4])), "v"(mk4(x, x, x, __uint_as_float((unsigned)r))) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                } else {
                    asm volatile("ds_write_b128 %0, %1" ::"v"(lds_off(&buf[side][lane * 4])), "v"(mk4(x, x, x, __uint_as_float((unsigned)r))) : "memory");
                    if (MODE == 0 && lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(lds_off(&flag[side])), "v"((unsigned)r) : "memory");
                }
            } else {
                v4f q;
                if (MODE == 2) {
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(lds_off(&buf[side][lane * 4])) : "memory");
                } else if (MODE == 0) {
                    unsigned f;
                    do {
                        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(lds_off(&flag[side])) : "memory");
                        f = (unsigned)__builtin_amdgcn_readfirstlane((int)f);
                        if (f < (unsigned)r) __builtin_amdgcn_s_sleep(1);
                    } while (f < (unsigned)r);
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(lds_off(&buf[side][lane * 4])) : "memory");
                } else {
                    bool ok;
                    do {
                        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(lds_off(&buf[side][lane * 4])) : "memory");
                        ok = __ballot(__float_as_uint(q.w) != (unsigned)r) == 0ull;
                        if (MODE == 1 && !ok) __builtin_amdgcn_s_sleep(1);
                    } while (!ok);
                }
                x = q.x + 1e-6f;
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && wave == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 128 + threadIdx.x] = x;
}

// VALU issue: WAVES waves per workgroup of 64*WAVES threads, blocks chosen so that each SIMD hosts `per_simd` waves
template <int CHAINS>
__global__ void k_valu(long long* cyc, float* out) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    const float a = 0.999f, b = 1e-3f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 200; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (CHAINS == 1) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
            } else {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}

__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }

static double mean(const long long* h, int n) { double s = 0; for (int i = 0; i < n; ++i) s += (double)h[i]; return s / n; }

int main() {
    long long* cyc; float* out;
    hipMalloc(&cyc, 16384 * 8); hipMalloc(&out, 16384 * 256 * 4);
    static long long h[16384];
    const char* names[4] = {"flag word + data", "tag in the group, sleep", "s_barrier", "tag in the group, spin"};
    for (int blocks : {1, 1024}) {
        for (int work : {0, 32}) {
            for (int mode = 0; mode < 4; ++mode) {
                if (mode == 0) hipLaunchKernelGGL(k_ping<0>, dim3(blocks), dim3(128), 0, 0, cyc, out, work);
                if (mode == 1) hipLaunchKernelGGL(k_ping<1>, dim3(blocks), dim3(128), 0, 0, cyc, out, work);
                if (mode == 2) hipLaunchKernelGGL(k_ping<2>, dim3(blocks), dim3(128), 0, 0, cyc, out, work);
                if (mode == 3) hipLaunchKernelGGL(k_ping<3>, dim3(blocks), dim3(128), 0, 0, cyc, out, work);
                hipDeviceSynchronize();
                hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost);
                printf("hand-off %-26s blocks %4d, %2d fma before each send: %7.1f ticks per one-way hand-off (incl. the work)\n", names[mode], blocks, work,
                       mean(h, blocks) / (2.0 * ROUNDS));
            }
        }
    }
    // VALU issue rate against waves per SIMD: 256 CUs x 4 SIMDs = 1024 SIMDs; workgroups of 64 threads
    for (int per_simd = 1; per_simd <= 4; ++per_simd) {
        const int blocks = 1024 * per_simd;
        for (int chains : {1, 4}) {
            if (chains == 1) hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(64), 0, 0, cyc, out);
            else hipLaunchKernelGGL(k_valu<4>, dim3(blocks), dim3(64), 0, 0, cyc, out);
            hipDeviceSynchronize();
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a, 0);
            if (chains == 1) hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(64), 0, 0, cyc, out);
            else hipLaunchKernelGGL(k_valu<4>, dim3(blocks), dim3(64), 0, 0, cyc, out);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost);
            printf("valu %d waves per SIMD, %d chains: %5.2f ticks per instruction per wave, kernel %.1f us\n", per_simd, chains, mean(h, blocks) / (200.0 * 64), ms * 1e3);
        }
    }
    // the same with workgroups of 128 / 192 / 256 threads (waves of a workgroup are dealt to the SIMDs in turn)
    for (int waves = 2; waves <= 4; ++waves) {
        hipLaunchKernelGGL(k_valu<4>, dim3(1024), dim3(64 * waves), 0, 0, cyc, out);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 8 * 1024 * waves, hipMemcpyDeviceToHost);
        printf("valu 1024 workgroups of %d waves, 4 chains: %5.2f ticks per instruction per wave\n", waves, mean(h, 1024 * waves) / (200.0 * 64));
    }
    // empty-kernel launch cost inside a graph by workgroup shape
    hipStream_t s; hipStreamCreate(&s);
    const int shapes[5][2] = {{1024, 128}, {1024, 192}, {1024, 256}, {512, 256}, {512, 384}};
    for (auto& sh : shapes) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(sh[0]), dim3(sh[1]), 16384, s, nullptr);
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e9;
        for (int r = 0; r < 5; ++r) { hipEventRecord(a, s); hipGraphLaunch(ge, s); hipEventRecord(b, s); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
        printf("empty kernel %4d x %3d threads, 16 KiB LDS: %.2f us per launch in a graph\n", sh[0], sh[1], best / 200 * 1e3);
    }
    return 0;
}
