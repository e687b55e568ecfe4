// micro-benchmark: VALU issue model of one wave per SIMD on gfx950 -- cycles per instruction for dependent chains,
// independent chains (ILP 2/4/8), packed f32, half-masked waves, transcendentals and the integer multiplies Philox uses.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ub_valu tools/ub_valu.hip && /tmp/ub_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP 64
#define ITER 200

template <int MODE>
__global__ __launch_bounds__(64, 1) void k(float* out, long long* cyc, int half) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    const float a = 0.999f, b = 1e-3f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    const f2 pa = {a, a}, pb = {b, b};
    unsigned u0 = threadIdx.x + 1, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
    if (half && threadIdx.x >= 32) return;   // MODE with half != 0: only the lower 32 lanes stay active
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (MODE == 0) {        // 1 dependent chain
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
            } else if (MODE == 1) { // 2 chains
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
                }
            } else if (MODE == 2) { // 4 chains
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
                }
            } else if (MODE == 3) { // 8 chains
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x4) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x5) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x6) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x7) : "v"(a), "v"(b));
            } else if (MODE == 4) { // packed, 1 chain (8 pk instructions = 16 fma)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pa), "v"(pb));
            } else if (MODE == 5) { // packed, 4 chains
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pa), "v"(pb));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pa), "v"(pb));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(pa), "v"(pb));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(pa), "v"(pb));
                }
            } else if (MODE == 6) { // transcendental, 4 chains: v_exp_f32
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x0));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x1));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x2));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x3));
                }
            } else if (MODE == 7) { // v_rcp dependent
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_rcp_f32 %0, %0" : "+v"(x0));
            } else if (MODE == 8) { // v_mul_hi_u32, 4 chains
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u0) : "v"(u3));
                    asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u1) : "v"(u3));
                    asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u2) : "v"(u3));
                    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u0) : "v"(u3));
                }
            } else if (MODE == 9) { // v_cndmask / v_max mix, 4 chains (clamps, selects)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    asm volatile("v_max_f32 %0, %0, %1" : "+v"(x0) : "v"(b));
                    asm volatile("v_min_f32 %0, %0, %1" : "+v"(x1) : "v"(a));
                    asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(b), "v"(a));
                    asm volatile("v_mov_b32 %0, %1" : "+v"(x3) : "v"(x2));
                }
            } else if (MODE == 10) { // fma dependent chain alternating with independent transcendental
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x1));
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(u0 + u1 + u2);
}

template <int MODE>
static void run(const char* name, int blocks, int half, float* out, long long* cyc, float instr_per_rep) {
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, half);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, half);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    static long long h[8192];
    hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
    const double n_instr = (double)ITER * (REP / 8) * instr_per_rep;
    printf("%-44s blocks %5d half %d: %6.2f memtime-ticks/instr, kernel %.1f us -> %.2f ns/instr\n", name, blocks, half, s / blocks / n_instr, ms * 1e3,
           ms * 1e6 / n_instr);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 8192 * 64 * 4); hipMalloc(&cyc, 8192 * 8);
    for (int blocks : {1024, 2048}) {
        run<0>("v_fma_f32 1 dependent chain", blocks, 0, out, cyc, 8);
        run<1>("v_fma_f32 2 chains", blocks, 0, out, cyc, 8);
        run<2>("v_fma_f32 4 chains", blocks, 0, out, cyc, 8);
        run<3>("v_fma_f32 8 chains", blocks, 0, out, cyc, 8);
        run<3>("v_fma_f32 8 chains, lanes 32..63 off", blocks, 1, out, cyc, 8);
        run<4>("v_pk_fma_f32 1 chain (per pk instr)", blocks, 0, out, cyc, 8);
        run<5>("v_pk_fma_f32 4 chains (per pk instr)", blocks, 0, out, cyc, 8);
        run<6>("v_exp_f32 4 chains", blocks, 0, out, cyc, 8);
        run<7>("v_rcp_f32 dependent", blocks, 0, out, cyc, 8);
        run<8>("v_mul_hi/lo_u32 mix", blocks, 0, out, cyc, 8);
        run<9>("min/max/med3/mov 4 chains", blocks, 0, out, cyc, 8);
        run<10>("fma chain + independent v_exp interleaved", blocks, 0, out, cyc, 8);
    }
    return 0;
}
