// clock_trace.hip — the shader clock and the progress of a wavefront over TIME, at ~15 us resolution, while the chip runs
// config5's instruction mix at four wavefronts per SIMD (the load of the benchmark's shard).
// Why: tools/rt_tail_probe.py finds one real-time block in every ~2.6 ms ~80 us slower than its neighbours, whatever the block
// size and whether blocks are paced - a cycle in time that the library does not produce.  Here a few wavefronts of a grid that
// fills the chip record (s_memrealtime = 100 MHz wall clock, s_memtime = shader clock) every CHUNK loop iterations.  Printed:
// the shader clock per interval (its levels and how often it changes), intervals in which the wavefront made no or little
// progress although wall time passed (stalls: the clock stopped or the wavefront was not scheduled), and the period of both.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/micro/clock_trace.hip -o tools/micro/clock_trace && tools/micro/clock_trace [ms] [duty%]
// duty < 100: the grid is launched in bursts of ~500 us with idle gaps in between (a real-time stream of blocks) instead of as one
// long kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

#define OPS : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), "+v"(d0), "+v"(d1) : "v"(oned), "v"(one), "v"(zero)
// %0..%7 floats, %8, %9 doubles, %10 = 1.0 (double), %11 = 1.0f, %12 = 0.0f
#define MACS(r) "v_mul_f32 %" #r ", %11, %" #r "\n v_add_f32 %" #r ", %12, %" #r "\n v_med3_f32 %" #r ", %" #r ", -1.0, 1.0\n"
#define QUARTET "v_cvt_f64_f32 %8, %0\n v_cvt_f64_f32 %9, %2\n v_fma_f64 %8, %8, %10, %9\n v_cvt_f32_f64 %3, %8\n"

constexpr int kChunk = 32;        // loop iterations (of 48 instructions) between two records
constexpr int kRecorders = 16;    // wavefronts that record (spread over the grid)

__global__ void __launch_bounds__(64) load(float* out, uint64_t* trace, int chunks, int recordEvery) {
    const int lane = threadIdx.x;
    float f0 = 0.5f + 0.001f * lane, f1 = f0 + 0.01f, f2 = f0 + 0.02f, f3 = f0 + 0.03f, f4 = f0 + 0.04f, f5 = f0 + 0.05f, f6 = f0 + 0.06f, f7 = f0 + 0.07f;
    double d0 = f0, d1 = f1;
    const float one = 1.0f, zero = 0.0f;
    const double oned = 1.0;
    const bool rec = blockIdx.x % recordEvery == 0 && (int)(blockIdx.x / recordEvery) < kRecorders;
    uint64_t* mine = trace + (size_t)(blockIdx.x / recordEvery) * (size_t)(chunks + 1) * 2;
    for (int c = 0; c <= chunks; ++c) {
        if (rec && lane == 0) {
            mine[2 * c] = wall_clock64();
            mine[2 * c + 1] = __builtin_readcyclecounter();
        }
        if (c == chunks) break;
        for (int it = 0; it < kChunk; ++it)
            asm volatile(".rept 4\n" MACS(0) MACS(1) "v_mul_f32 %2, %11, %2\n" QUARTET "v_med3_f32 %3, %3, -1.0, 1.0\n" ".endr\n" OPS);
    }
    if (f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (float)d0 + (float)d1 == 12345.678f) out[blockIdx.x * 64 + lane] = f0;
}

static double pct(std::vector<double> v, double q) {
    std::sort(v.begin(), v.end());
    return v[std::min(v.size() - 1, (size_t)std::ceil(q * v.size()) - (q > 0 ? 1 : 0))];
}

int main(int argc, char** argv) {
    const double ms = argc > 1 ? std::atof(argv[1]) : 60.0;
    const int duty = argc > 2 ? std::atoi(argv[2]) : 100;
    const int grid = 4096;   // 4 single-wavefront workgroups per SIMD
    float* out = nullptr;
    uint64_t* trace = nullptr;
    CHECK(hipMalloc(&out, (size_t)grid * 64 * 4));
    // calibrate: how long does one chunk take?
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipMalloc(&trace, (size_t)kRecorders * 2 * 8 * (1 << 16)));
    for (int warm = 0; warm < 3; ++warm) {   // bring the chip to its loaded state
        hipLaunchKernelGGL(load, dim3(grid), dim3(64), 0, 0, out, trace, 2000, grid / kRecorders);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(load, dim3(grid), dim3(64), 0, 0, out, trace, 1000, grid / kRecorders);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float cal = 0;
    CHECK(hipEventElapsedTime(&cal, e0, e1));
    const double chunkUs = cal * 1e3 / 1000;
    std::printf("one chunk (%d iterations x 48 instructions, 4 wavefronts per SIMD) = %.2f us\n", kChunk, chunkUs);

    if (duty >= 100) {
        const int chunks = std::min<int>((1 << 16) - 1, (int)(ms * 1e3 / chunkUs));
        hipLaunchKernelGGL(load, dim3(grid), dim3(64), 0, 0, out, trace, chunks, grid / kRecorders);
        CHECK(hipDeviceSynchronize());
        std::vector<uint64_t> h((size_t)kRecorders * (chunks + 1) * 2);
        CHECK(hipMemcpy(h.data(), trace, h.size() * 8, hipMemcpyDeviceToHost));
        std::printf("one kernel of %d chunks (%.1f ms), %d recording wavefronts\n", chunks, chunks * chunkUs * 1e-3, kRecorders);
        for (int r = 0; r < kRecorders; r += 5) {
            const uint64_t* t = h.data() + (size_t)r * (chunks + 1) * 2;
            std::vector<double> us, mhz;
            for (int c = 0; c < chunks; ++c) {
                const double dt = (double)(t[2 * (c + 1)] - t[2 * c]) / 100.0;   // us
                const double dc = (double)(t[2 * (c + 1) + 1] - t[2 * c + 1]);
                us.push_back(dt);
                mhz.push_back(dt > 0 ? dc / dt : 0);
            }
            const double med = pct(us, 0.5);
            std::printf("wavefront %2d: chunk time us: median %.2f  p1 %.2f  p99 %.2f  p99.9 %.2f  max %.2f;  shader clock MHz: median %.0f  p1 %.0f  p99 %.0f  min %.0f  max %.0f\n", r, med,
                        pct(us, 0.01), pct(us, 0.99), pct(us, 0.999), *std::max_element(us.begin(), us.end()), pct(mhz, 0.5), pct(mhz, 0.01), pct(mhz, 0.99),
                        *std::min_element(mhz.begin(), mhz.end()), *std::max_element(mhz.begin(), mhz.end()));
            // slow chunks: when, how much longer, at which clock; the gaps between them
            std::vector<int> slow;
            for (int c = 0; c < chunks; ++c)
                if (us[c] > 1.5 * med) slow.push_back(c);
            std::printf("   chunks over 1.5 x the median: %zu", slow.size());
            double lost = 0;
            for (int c : slow) lost += us[c] - med;
            std::printf(" (%.1f us lost in all = %.2f %% of the run)\n", lost, 100.0 * lost / (chunks * med));
            std::printf("   first of them: ");
            for (size_t k = 0; k < std::min<size_t>(slow.size(), 14); ++k)
                std::printf("[t=%.2f ms: %.1f us at %.0f MHz] ", (double)(t[2 * slow[k]] - t[0]) / 1e5, us[slow[k]], mhz[slow[k]]);
            std::printf("\n   gaps between consecutive ones (ms): ");
            for (size_t k = 1; k < std::min<size_t>(slow.size(), 20); ++k) std::printf("%.2f ", (double)(t[2 * slow[k]] - t[2 * slow[k - 1]]) / 1e5);
            std::printf("\n");
            // the clock over time: 1 ms averages of the first 24 ms
            std::printf("   shader clock, 0.5 ms averages: ");
            double accC = 0, accT = 0;
            int printed = 0;
            for (int c = 0; c < chunks && printed < 48; ++c) {
                accC += (double)(t[2 * (c + 1) + 1] - t[2 * c + 1]);
                accT += us[c];
                if (accT >= 500.0) {
                    std::printf("%.0f ", accC / accT);
                    accC = accT = 0;
                    ++printed;
                }
            }
            std::printf("\n");
        }
    } else {
        // bursts: ~500 us of load, then idle until the next multiple of the period (500 us / duty)
        const int chunks = std::max(1, (int)(500.0 / chunkUs));
        const double period = 500.0 * 100.0 / duty;
        const int bursts = (int)(ms * 1e3 / period);
        std::vector<float> kernelUs;
        const auto start = std::chrono::steady_clock::now();
        for (int b = 0; b < bursts; ++b) {
            while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - start).count() < b * period) {}
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(load, dim3(grid), dim3(64), 0, 0, out, trace, chunks, grid / kRecorders);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float t = 0;
            CHECK(hipEventElapsedTime(&t, e0, e1));
            kernelUs.push_back(t * 1e3f);
        }
        std::vector<double> v(kernelUs.begin(), kernelUs.end());
        std::printf("%d bursts of %d chunks, one per %.1f us (duty %d %%): kernel us median %.1f  p1 %.1f  p99 %.1f  p99.9 %.1f  max %.1f\n", bursts, chunks, period, duty, pct(v, 0.5),
                    pct(v, 0.01), pct(v, 0.99), pct(v, 0.999), *std::max_element(v.begin(), v.end()));
        std::printf("   the first 40: ");
        for (int b = 0; b < std::min(bursts, 40); ++b) std::printf("%.0f ", kernelUs[b]);
        std::printf("\n");
    }
    return 0;
}
