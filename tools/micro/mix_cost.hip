// mix_cost.hip — marginal cost of instruction classes inside a realistic mix (config5: 2 MACS + 1 INTERP per group).
// Variants drop one class at a time; time per group and SIMD at 4 waves per SIMD tells what each class costs when the
// SIMD interleaves four waves of mixed code (a homogeneous loop, valu_rate.hip, overstates the cost of the 4-clock class).
// hipcc --offload-arch=gfx950 -O3 tools/micro/mix_cost.hip -o tools/micro/mix_cost && tools/micro/mix_cost
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define MACS(r, sat) "v_mul_f32 %" #r ", %11, %" #r "\n v_add_f32 %" #r ", %12, %" #r "\n" sat
#define MED(r) "v_med3_f32 %" #r ", %" #r ", -1.0, 1.0\n"
#define INTERP_F64 "v_mul_f32 %2, %11, %2\n v_cvt_f64_f32 %8, %0\n v_cvt_f64_f32 %9, %2\n v_fma_f64 %8, %8, %10, %9\n v_cvt_f32_f64 %3, %8\n"
#define INTERP_MULONLY "v_mul_f32 %2, %11, %2\n"
#define INTERP_FAST5 "v_mul_f32 %2, %11, %2\n v_mul_f32 %4, %11, %4\n v_add_f32 %4, %12, %4\n v_mul_f32 %5, %11, %5\n v_add_f32 %3, %12, %5\n"
#define OPS : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), "+v"(d0), "+v"(d1) : "v"(oned), "v"(one), "v"(zero)

template <int KIND>
__global__ void __launch_bounds__(64) k(float* out, int iters) {
    const int lane = threadIdx.x;
    float f0 = 0.5f + 0.001f * lane, f1 = f0 + 0.01f, f2 = f0 + 0.02f, f3 = f0 + 0.03f, f4 = f0 + 0.04f, f5 = f0 + 0.05f, f6 = f0 + 0.06f, f7 = f0 + 0.07f;
    double d0 = f0, d1 = f1;
    const float one = 1.0f, zero = 0.0f;
    const double oned = 1.0;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(".rept 10\n" MACS(0, MED(0)) MACS(1, MED(1)) INTERP_F64 MED(3) ".endr\n" OPS);          // full: 12
        else if (KIND == 1) asm volatile(".rept 10\n" MACS(0, "") MACS(1, "") INTERP_F64 ".endr\n" OPS);                      // no med3: 9
        else if (KIND == 2) asm volatile(".rept 10\n" MACS(0, MED(0)) MACS(1, MED(1)) INTERP_MULONLY MED(3) ".endr\n" OPS); // no fp64-class: 8
        else if (KIND == 3) asm volatile(".rept 10\n" MACS(0, "") MACS(1, "") INTERP_MULONLY ".endr\n" OPS);                  // fast only: 5
        else if (KIND == 4) asm volatile(".rept 10\n" MACS(0, MED(0)) MACS(1, MED(1)) INTERP_FAST5 MED(3) ".endr\n" OPS);   // fp64-class replaced by fast ops: 12
        else if (KIND == 5) asm volatile(".rept 10\n" MACS(0, MED(0)) MACS(1, MED(1)) "v_mul_f32 %2, %11, %2\n v_cvt_f64_f32 %8, %0\n v_fma_f64 %8, %8, %10, %8\n v_cvt_f32_f64 %3, %8\n" MED(3) ".endr\n" OPS);  // one cvt fewer: 11
        else if (KIND == 6)  // the full mix, fp64-class instructions spread between the others instead of in a row
            asm volatile(".rept 10\n"
                         "v_mul_f32 %0, %11, %0\n v_cvt_f64_f32 %8, %4\n v_add_f32 %0, %12, %0\n v_med3_f32 %0, %0, -1.0, 1.0\n"
                         "v_cvt_f64_f32 %9, %5\n v_mul_f32 %1, %11, %1\n v_add_f32 %1, %12, %1\n v_fma_f64 %8, %8, %10, %9\n"
                         "v_med3_f32 %1, %1, -1.0, 1.0\n v_mul_f32 %2, %11, %2\n v_cvt_f32_f64 %3, %8\n v_med3_f32 %6, %6, -1.0, 1.0\n"
                         ".endr\n" OPS);
    }
    out[blockIdx.x * 64 + lane] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (float)(d0 + d1);
}

template <int KIND>
double run(const char* name, int n, float* d) {
    const int waves = 4096, iters = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(waves), dim3(64), 0, 0, d, iters / 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(waves), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)iters * 10 * 4;  // groups per SIMD
    printf("%-44s %2d instructions per group: %7.2f ms  %6.2f ns per group and SIMD  (%.3f ns per instruction)\n", name, n, ms, ms * 1e6 / groups, ms * 1e6 / groups / n);
    return ms * 1e6 / groups;
}

int main() {
    float* d;
    hipMalloc(&d, 4096 * 64 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("full mix (2 MACS + INTERP, saturating)", 12, d);
        run<1>("without the 3 v_med3_f32", 9, d);
        run<2>("without the 4 fp64-class instructions", 8, d);
        run<3>("plain fp32 only (5 mul / add)", 5, d);
        run<4>("fp64-class replaced by 4 more mul / add", 12, d);
        run<5>("one v_cvt_f64_f32 fewer", 11, d);
        run<6>("full mix, fp64-class spread out", 12, d);
    }
    return 0;
}
