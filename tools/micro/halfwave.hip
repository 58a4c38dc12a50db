// halfwave.hip — does a wave64 VALU instruction cost less when only lanes 0..31 are active (EXEC[63:32] = 0)?
// 8 waves per SIMD, a long loop of independent v_fma_f32; compare the time of full waves with half-active ones.
// hipcc --offload-arch=gfx950 -O3 tools/micro/halfwave.hip -o /tmp/halfwave && /tmp/halfwave
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k(float* out, int iters, int activeLanes) {
    const int lane = threadIdx.x;
    if (lane >= activeLanes) return;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    const float m = 0.9999f, c = 0.0001f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
        a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
    }
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    float* d;
    hipMalloc(&d, 8192 * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves : {1024, 2048, 4096, 8192})
        for (int lanes : {64, 32, 16}) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, d, 200000, lanes);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            // 8 fma per iteration
            printf("waves %5d (%.0f per SIMD) active lanes %2d: %8.3f ms  -> %.3f ns per wave-instruction per SIMD\n", waves, waves / 1024.0, lanes, best,
                   best * 1e6 / (200000.0 * 8 * (waves / 1024.0)));
        }
    return 0;
}
