// placement.hip — where do the waves of a grid of one-wave workgroups go, and when do they run?  Every wave records its
// XCC / SE / CU / SIMD (HW_ID registers) and its start and end on the 100 MHz clock; the host prints how many waves each
// SIMD received and how many ran concurrently.  workgroup = 64 x wavesPerGroup threads.
// hipcc --offload-arch=gfx950 -O3 tools/micro/placement.hip -o tools/micro/placement && tools/micro/placement
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

struct Rec { uint64_t t0, t1; uint32_t hwid, xcc; };

// the same with the instruction mix of a translated INTERP + MACS chain (fp64 fma and conversions between fp32 multiplies, adds and
// v_med3): a kernel at the board's power limit, where every XCC settles at a clock of its own
__global__ void __attribute__((amdgpu_waves_per_eu(4, 4))) kHeavy(float* out, Rec* rec, int iters, int rotate) {   // (4 per SIMD: the 128-register build's residency)
    float f0 = threadIdx.x * 0.001f, f1 = f0 + 0.1f, f2 = f0 + 0.2f, f3 = f0 + 0.3f;
    const float m = 0.9999f;
    const double c = 0.7;
    uint64_t r0 = wall_clock64();
    uint32_t slot;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(slot));
    for (int it = 0; it < iters; ++it) {
        if (rotate > 0 && (it & ((1 << rotate) - 1)) == 0) {   // every 2^rotate iterations: priority ((it >> rotate) + wave-buffer slot) & 3
            switch (((it >> rotate) + slot) & 3) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
        }
#define SECTION(f)                                                                                              \
        {                                                                                                       \
            float p;                                                                                            \
            double a, b;                                                                                        \
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(m), "v"(f));                                    \
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a) : "v"(f));                                            \
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(b) : "v"(p));                                            \
            asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(b) : "v"(c), "v"(a));                                \
            asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(b));                                            \
            asm volatile("v_med3_f32 %0, %0, -1.0, 1.0" : "+v"(f));                                             \
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(m), "v"(f));                                    \
            asm volatile("v_add_f32 %0, %1, %0" : "+v"(f) : "v"(p));                                            \
            asm volatile("v_med3_f32 %0, %0, -1.0, 1.0" : "+v"(f));                                             \
        }
        SECTION(f0) SECTION(f1) SECTION(f2) SECTION(f3)
#undef SECTION
    }
    uint64_t r1 = wall_clock64();
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
    out[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1 + f2 + f3;
    if ((threadIdx.x & 63) == 0) rec[wave] = Rec{r0, r1, hwid, xcc};
}

__global__ void k(float* out, Rec* rec, int iters) {
    float f0 = threadIdx.x * 0.001f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    const float m = 0.9999f;
    uint64_t r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f0) : "v"(m));
        asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f1) : "v"(m));
        asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f2) : "v"(m));
        asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f3) : "v"(m));
    }
    uint64_t r1 = wall_clock64();
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
    out[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1 + f2 + f3;
    if ((threadIdx.x & 63) == 0) rec[wave] = Rec{r0, r1, hwid, xcc};
}

int main(int argc, char** argv) {
    const int rotate = argc > 1 ? atoi(argv[1]) : 0;   // heavy kernel: rotate the wave priorities every 2^rotate iterations (0: leave them)
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    float* d; Rec* dr;
    hipMalloc(&d, 16384 * 64 * 4);
    hipMalloc(&dr, 16384 * sizeof(Rec));
    for (int heavy : {0, 1})
    for (int wavesPerGroup : {1, 4})
        for (int waves : {1024, 4096, 8192}) {
            if (heavy && wavesPerGroup != 1) continue;
            const int iters = heavy ? 60000 : 400000;
            if (heavy && waves == 1024) printf("-- the INTERP + MACS mix (fp64 fma, conversions, v_med3): the board at its power limit; priorities rotated every 2^%d iterations (0: not)\n", rotate);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            if (heavy) hipLaunchKernelGGL(kHeavy, dim3(waves / wavesPerGroup), dim3(64 * wavesPerGroup), 0, 0, d, dr, 1000, rotate);
            else hipLaunchKernelGGL(k, dim3(waves / wavesPerGroup), dim3(64 * wavesPerGroup), 0, 0, d, dr, 1000);
            hipEventRecord(e0);
            if (heavy) hipLaunchKernelGGL(kHeavy, dim3(waves / wavesPerGroup), dim3(64 * wavesPerGroup), 0, 0, d, dr, iters, rotate);
            else hipLaunchKernelGGL(k, dim3(waves / wavesPerGroup), dim3(64 * wavesPerGroup), 0, 0, d, dr, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<Rec> h(waves);
            hipMemcpy(h.data(), dr, waves * sizeof(Rec), hipMemcpyDeviceToHost);
            std::map<uint32_t, int> perSimd, perCu;
            uint64_t first = ~0ull, last = 0;
            double busy = 0;
            for (const Rec& r : h) {
                const uint32_t simd = (r.hwid >> 4) & 3, cu = (r.hwid >> 8) & 15, sh = (r.hwid >> 12) & 1, se = (r.hwid >> 13) & 7, xcc = r.xcc & 15;
                perSimd[(xcc << 12) | (se << 8) | (sh << 7) | (cu << 2) | simd]++;
                perCu[(xcc << 12) | (se << 8) | (sh << 7) | (cu << 2)]++;
                first = std::min(first, r.t0); last = std::max(last, r.t1);
                busy += (double)(r.t1 - r.t0);
            }
            std::map<int, int> histogram;
            for (auto& kv : perSimd) histogram[kv.second]++;
            printf("waves %5d in groups of %d: %7.2f ms, span %.2f ms, mean wave %.2f ms (%.0f %% of span); SIMDs used %zu, CUs used %zu; waves per SIMD:",
                   waves, wavesPerGroup, ms, (last - first) * 1e-5, busy / waves * 1e-5, 100.0 * busy / waves / (double)(last - first), perSimd.size(), perCu.size());
            for (auto& kv : histogram) printf("  %d x%d", kv.first, kv.second);
            printf("\n");
            // per XCC: how long its waves ran (a launch that fills every slot once ends with its slowest wave: the spread between
            // XCCs - each has a clock of its own under the board's power limit - is what a single-round launch loses against
            // the mean; a launch of several rounds hands the faster XCCs more waves instead)
            if (heavy && waves == 4096) {
                // which wave-buffer slots do the four waves of a SIMD sit in, and which workgroup ids share a SIMD?
                std::map<uint32_t, int> slots;
                std::map<uint32_t, std::vector<int>> idsOf;
                for (int w = 0; w < waves; ++w) {
                    const Rec& r = h[w];
                    const uint32_t simd = (r.hwid >> 4) & 3, cu = (r.hwid >> 8) & 15, sh = (r.hwid >> 12) & 1, se = (r.hwid >> 13) & 7, xcc = r.xcc & 15;
                    slots[r.hwid & 15]++;
                    idsOf[(xcc << 12) | (se << 8) | (sh << 7) | (cu << 2) | simd].push_back(w);
                }
                printf("        wave-buffer slots in use:");
                for (auto& kv : slots) printf("  slot %u x%d", kv.first, kv.second);
                printf("\n        workgroup ids that share a SIMD (three SIMDs):");
                int shown = 0;
                for (auto& kv : idsOf) {
                    if (shown++ >= 3) break;
                    printf("  {");
                    for (int w : kv.second) printf(" %d(slot %u, %.2f ms)", w, h[w].hwid & 15, (double)(h[w].t1 - h[w].t0) * 1e-5);
                    printf(" }");
                }
                printf("\n");
            }
            std::map<uint32_t, std::vector<double>> perXcc;
            for (const Rec& r : h) perXcc[r.xcc & 15].push_back((double)(r.t1 - r.t0) * 1e-5);
            printf("        per XCC, wave duration mean (min .. max) ms, waves:");
            for (auto& kv : perXcc) {
                double sum = 0, lo = 1e30, hi = 0;
                for (double t : kv.second) { sum += t; lo = std::min(lo, t); hi = std::max(hi, t); }
                printf("  [%u] %.2f (%.2f .. %.2f) x%zu", kv.first, sum / kv.second.size(), lo, hi, kv.second.size());
            }
            printf("\n");
        }
    return 0;
}
