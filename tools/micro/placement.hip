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

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    float* d; Rec* dr;
    hipMalloc(&d, 16384 * 64 * 4);
    hipMalloc(&dr, 16384 * sizeof(Rec));
    for (int wavesPerGroup : {1, 4})
        for (int waves : {1024, 4096, 8192}) {
            const int iters = 400000;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(waves / wavesPerGroup), dim3(64 * wavesPerGroup), 0, 0, d, dr, 1000);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(waves / wavesPerGroup), dim3(64 * wavesPerGroup), 0, 0, d, dr, iters);
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
