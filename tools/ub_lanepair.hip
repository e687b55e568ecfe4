// micro-benchmark for the LANE-PAIR split of the physics chain (review of round 5, item 3): what the inputs of its cost model are
// on gfx950 -- (a) a cross-lane exchange inside a dependent VALU chain (v_mov_b32_dpp quad_perm:[1,0,3,2], the partner lane of a
// pair; the same exchange folded into a VOP2 instruction's DPP operand), (b) the issue interval a wave sees with 1, 2, 3 and 4
// waves resident per SIMD (the split puts THREE waves per 64 envs on a SIMD where k_step2 has two).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ub_lanepair tools/ub_lanepair.hip && /tmp/ub_lanepair
// Output: ticks of s_memtime per instruction of ONE wave (median over the waves of the launch).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define ITER 100
#define REP 64   // instructions of the measured kind per iteration

// MODE 0: dependent v_fma chain.  1: every 8th instruction of the chain is a DPP exchange with the partner lane (v_mov_b32_dpp)
// whose result the next fma consumes.  2: the exchange folded into a dependent v_add_f32_dpp (VOP2 with a DPP source): 8 per 64.
// 3: four independent chains (what a scheduler makes of straight-line physics code).  4: four chains, 8 exchanges per 64.
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, y = 0.f;
    const float a = 0.999f, b = 1e-3f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
            } else if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 7; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x0));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x0) : "v"(y), "v"(b));
            } else if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 7; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x0));
            } else if (MODE == 3) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
                }
            } else {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
                asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x3));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x3) : "v"(y), "v"(b));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static double run(int waves_per_simd, float* out, long long* cyc, int n_simd) {
    const int blocks = n_simd * waves_per_simd;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    // instructions of the wave per iteration: REP of the measured kind (+ 8 exchanges + 8 consumers in the modes that have them)
    const double instr = (MODE == 1 || MODE == 4) ? REP + 8 : REP;
    return (double)h[blocks / 2] / (ITER * instr);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_simd = p.multiProcessorCount * 4;
    float* out; long long* cyc;
    hipMalloc((void**)&out, (size_t)n_simd * 8 * 64 * sizeof(float));
    hipMalloc((void**)&cyc, (size_t)n_simd * 8 * sizeof(long long));
    printf("%s, %d CUs; ticks of s_memtime (100 MHz x ... see the note in profiles) per VALU instruction of ONE wave, median over the launch\n", p.name, p.multiProcessorCount);
    printf("%-74s %8s %8s %8s %8s\n", "waves resident per SIMD (blocks of one wave, grid = SIMDs x waves)", "1", "2", "3", "4");
    const char* names[5] = {"0: one dependent v_fma chain", "1: ... every 8th link a v_mov_b32_dpp quad_perm:[1,0,3,2] exchange + consumer",
                            "2: ... every 8th link a v_add_f32_dpp (exchange folded into the VOP2 operand)", "3: four independent v_fma chains",
                            "4: four chains, one exchange + consumer per 8 instructions"};
    for (int m = 0; m < 5; ++m) {
        double r[4];
        for (int w = 1; w <= 4; ++w)
            r[w - 1] = m == 0 ? run<0>(w, out, cyc, n_simd) : m == 1 ? run<1>(w, out, cyc, n_simd) : m == 2 ? run<2>(w, out, cyc, n_simd)
                     : m == 3 ? run<3>(w, out, cyc, n_simd) : run<4>(w, out, cyc, n_simd);
        printf("%-74s %8.2f %8.2f %8.2f %8.2f\n", names[m], r[0], r[1], r[2], r[3]);
    }
    return 0;
}
