// micro-benchmark: cost of launching 65 536 threads as 1024x64, 512x128, 256x256 workgroups (empty kernels, hipGraph)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
int main() {
    hipStream_t s; hipStreamCreate(&s);
    const int shapes[3][2] = {{1024, 64}, {512, 128}, {256, 256}};
    for (int lds = 0; lds <= 32768; lds += 32768)
    for (auto& sh : shapes) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(sh[0]), dim3(sh[1]), lds * (sh[1] / 64), s, nullptr);
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e9;
        for (int r = 0; r < 5; ++r) { hipEventRecord(a, s); hipGraphLaunch(ge, s); hipEventRecord(b, s); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
        printf("%4d x %3d threads, %5d B LDS per wave: %.2f us per launch\n", sh[0], sh[1], lds, best / 200 * 1e3);
    }
    return 0;
}
