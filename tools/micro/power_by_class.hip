// power_by_class.hip — what do the watts buy?  
// The translated loop issues in 95-98 % of its quad-cycles and the chip runs it at 2.1-2.25 of 2.4 GHz, against its power limit
// (~1.3 kW): the remaining lever is the CLOCK, i.e. the energy of what is issued.  Per instruction class - a homogeneous loop at
// four wavefronts per SIMD (4 096 single-wavefront workgroups, the shape of the 1/8 shard of configs[4]) kept running for ~1.5 s
// so that the power management settles - this prints: time per wave-instruction and SIMD, the shader clock and socket power the
// chip sustained meanwhile (amdgpu hwmon, sampled every 10 ms from the host, first 30 % of the samples dropped), clocks per
// instruction at THAT clock, and energy per wave-instruction above idle.  Then config5's own mix (2 MACS + 1 INTERP per group, as
// in mix_cost.hip) in three forms: as generated (the INTERP quartet cvt, cvt, fma_f64, cvt in a row), with the quartet spread
// between the plain instructions, and with the quartet replaced by the 14 plain fp32 instructions an error-free fp32 INTERP would
// need (TwoProduct + TwoSum + residual + certificate: DESIGN.md) - priced in emulated INTERPs per second at the clock each
// sustains, which is what counts.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/micro/power_by_class.hip -o tools/micro/power_by_class && tools/micro/power_by_class
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>

#define OPS : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(oned), "v"(one), "v"(zero)
// operands: %0..%7 floats, %8..%11 doubles, %12 = 1.0 (double), %13 = 1.0f, %14 = 0.0f
#define MACS(r, sat) "v_mul_f32 %" #r ", %13, %" #r "\n v_add_f32 %" #r ", %14, %" #r "\n" sat
#define MED(r) "v_med3_f32 %" #r ", %" #r ", -1.0, 1.0\n"
#define QUARTET "v_cvt_f64_f32 %8, %0\n v_cvt_f64_f32 %9, %2\n v_fma_f64 %8, %8, %12, %9\n v_cvt_f32_f64 %3, %8\n"
#define PLAIN14 "v_mul_f32 %4, %13, %4\n v_add_f32 %4, %14, %4\n v_mul_f32 %5, %13, %5\n v_add_f32 %5, %14, %5\n v_mul_f32 %6, %13, %6\n v_add_f32 %6, %14, %6\n v_mul_f32 %7, %13, %7\n" \
                "v_add_f32 %7, %14, %7\n v_mul_f32 %4, %13, %4\n v_add_f32 %5, %14, %5\n v_mul_f32 %6, %13, %6\n v_add_f32 %7, %14, %7\n v_mul_f32 %4, %13, %4\n v_add_f32 %3, %14, %5\n"

template <int KIND>
__global__ void __launch_bounds__(64) k(float* out, int iters) {
    const int lane = threadIdx.x;
    float f0 = 0.5f + 0.001f * lane, f1 = f0 + 0.01f, f2 = f0 + 0.02f, f3 = f0 + 0.03f, f4 = f0 + 0.04f, f5 = f0 + 0.05f, f6 = f0 + 0.06f, f7 = f0 + 0.07f;
    double d0 = f0, d1 = f1, d2 = f2, d3 = f3;
    const float one = 1.0f, zero = 0.0f;
    const double oned = 1.0;
    for (int it = 0; it < iters; ++it) {
        // homogeneous loops: 48 instructions per iteration, four independent chains each
        if (KIND == 0) asm volatile(".rept 12\n v_mul_f32 %0, %13, %0\n v_mul_f32 %1, %13, %1\n v_mul_f32 %2, %13, %2\n v_mul_f32 %3, %13, %3\n .endr\n" OPS);
        else if (KIND == 1) asm volatile(".rept 12\n v_add_f32 %0, %14, %0\n v_add_f32 %1, %14, %1\n v_add_f32 %2, %14, %2\n v_add_f32 %3, %14, %3\n .endr\n" OPS);
        else if (KIND == 2) asm volatile(".rept 12\n v_fma_f32 %0, %0, %13, %14\n v_fma_f32 %1, %1, %13, %14\n v_fma_f32 %2, %2, %13, %14\n v_fma_f32 %3, %3, %13, %14\n .endr\n" OPS);
        else if (KIND == 3) asm volatile(".rept 12\n" MED(0) MED(1) MED(2) MED(3) ".endr\n" OPS);
        else if (KIND == 4) asm volatile(".rept 12\n v_cvt_f64_f32 %8, %0\n v_cvt_f64_f32 %9, %1\n v_cvt_f64_f32 %10, %2\n v_cvt_f64_f32 %11, %3\n .endr\n" OPS);
        else if (KIND == 5) asm volatile(".rept 12\n v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n .endr\n" OPS);
        else if (KIND == 6) asm volatile(".rept 12\n v_fma_f64 %8, %8, %12, %9\n v_fma_f64 %9, %9, %12, %10\n v_fma_f64 %10, %10, %12, %11\n v_fma_f64 %11, %11, %12, %8\n .endr\n" OPS);
        else if (KIND == 7) asm volatile(".rept 12\n v_add_f64 %8, %8, %9\n v_add_f64 %9, %9, %10\n v_add_f64 %10, %10, %11\n v_add_f64 %11, %11, %8\n .endr\n" OPS);
        else if (KIND == 8) asm volatile(".rept 12\n v_mul_f64 %8, %8, %12\n v_mul_f64 %9, %9, %12\n v_mul_f64 %10, %10, %12\n v_mul_f64 %11, %11, %12\n .endr\n" OPS);
        else if (KIND == 9) asm volatile(".rept 12\n v_mov_b32 %4, %0\n v_mov_b32 %5, %1\n v_mov_b32 %6, %2\n v_mov_b32 %7, %3\n .endr\n" OPS);
        else if (KIND == 10) asm volatile(".rept 12\n v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %4, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %5, %2, %3, vcc\n .endr\n" OPS : "vcc");
        // config5's mix: 4 groups of 12 (or 22) per iteration
        else if (KIND == 20) asm volatile(".rept 4\n" MACS(0, MED(0)) MACS(1, MED(1)) "v_mul_f32 %2, %13, %2\n" QUARTET MED(3) ".endr\n" OPS);
        else if (KIND == 21)
            asm volatile(".rept 4\n"
                         "v_mul_f32 %0, %13, %0\n v_cvt_f64_f32 %8, %4\n v_add_f32 %0, %14, %0\n v_med3_f32 %0, %0, -1.0, 1.0\n"
                         "v_cvt_f64_f32 %9, %5\n v_mul_f32 %1, %13, %1\n v_add_f32 %1, %14, %1\n v_fma_f64 %8, %8, %12, %9\n"
                         "v_med3_f32 %1, %1, -1.0, 1.0\n v_mul_f32 %2, %13, %2\n v_cvt_f32_f64 %3, %8\n v_med3_f32 %6, %6, -1.0, 1.0\n"
                         ".endr\n" OPS);
        else if (KIND == 22) asm volatile(".rept 4\n" MACS(0, MED(0)) MACS(1, MED(1)) "v_mul_f32 %2, %13, %2\n" PLAIN14 MED(3) ".endr\n" OPS);
        else if (KIND == 23) asm volatile(".rept 4\n" MACS(0, MED(0)) MACS(1, MED(1)) "v_mul_f32 %2, %13, %2\n" MED(3) ".endr\n" OPS);   // the mix without its INTERP arithmetic
    }
    out[blockIdx.x * 64 + lane] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (float)(d0 + d1 + d2 + d3);
}

struct Hwmon {
    std::string freq, power;
    double read(const std::string& p) const {
        FILE* f = p.empty() ? nullptr : fopen(p.c_str(), "r");
        if (!f) return -1;
        double v = -1;
        if (fscanf(f, "%lf", &v) != 1) v = -1;
        fclose(f);
        return v;
    }
};

Hwmon findHwmon() {
    Hwmon h;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), 0) != hipSuccess) return h;
    for (char* c = bdf; *c; ++c) *c = (char)tolower(*c);
    const std::string base = std::string("/sys/bus/pci/devices/") + bdf + "/hwmon";
    DIR* d = opendir(base.c_str());
    if (!d) return h;
    while (dirent* e = readdir(d)) {
        if (strncmp(e->d_name, "hwmon", 5) != 0) continue;
        const std::string dir = base + "/" + e->d_name;
        for (const char* name : {"/power1_input", "/power1_average"}) {
            FILE* f = fopen((dir + name).c_str(), "r");
            if (f) { fclose(f); if (h.power.empty()) h.power = dir + name; }
        }
        FILE* f = fopen((dir + "/freq1_input").c_str(), "r");
        if (f) { fclose(f); h.freq = dir + "/freq1_input"; }
    }
    closedir(d);
    return h;
}

struct Result { double nsPerInstr, mhz, watts; };

template <int KIND>
Result run(const Hwmon& hw, float* d, int perIter, double seconds) {
    const int waves = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // calibrate the iteration count for the wanted duration
    int iters = 2000;
    hipLaunchKernelGGL(k<KIND>, dim3(waves), dim3(64), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(waves), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * seconds * 1e3 / ms);
    std::atomic<bool> stop{false};
    std::vector<std::pair<double, double>> samples;
    std::thread sampler([&] {
        while (!stop) {
            samples.emplace_back(hw.read(hw.freq), hw.read(hw.power));
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    });
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(waves), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    stop = true;
    sampler.join();
    hipEventElapsedTime(&ms, e0, e1);
    double f = 0, w = 0;
    int nf = 0, nw = 0;
    for (size_t i = samples.size() * 3 / 10; i < samples.size(); ++i) {
        if (samples[i].first > 0) { f += samples[i].first; ++nf; }
        if (samples[i].second > 0) { w += samples[i].second; ++nw; }
    }
    Result r;
    r.nsPerInstr = (double)ms * 1e6 / ((double)iters * perIter * 4.0);   // per wave-instruction and SIMD (4 wavefronts per SIMD)
    r.mhz = nf ? f / nf / 1e6 : -1;
    r.watts = nw ? w / nw / 1e6 : -1;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return r;
}

int main() {
    float* d;
    hipMalloc(&d, 4096 * 64 * 4);
    const Hwmon hw = findHwmon();
    printf("hwmon: clock %s, power %s\n", hw.freq.empty() ? "(not found)" : hw.freq.c_str(), hw.power.empty() ? "(not found)" : hw.power.c_str());
    std::this_thread::sleep_for(std::chrono::milliseconds(1500));
    double idleW = 0;
    for (int i = 0; i < 20; ++i) { idleW += hw.read(hw.power) / 1e6; std::this_thread::sleep_for(std::chrono::milliseconds(20)); }
    idleW /= 20;
    printf("idle socket power %.0f W\n\n", idleW);
    const double secs = 1.5;
    printf("%-52s %9s %8s %7s %9s %12s\n", "instruction class (4 wavefronts per SIMD, 1 024 SIMDs)", "ns/instr", "MHz", "W", "clk/instr", "nJ/wave-instr");
    auto line = [&](const char* name, const Result& r) {
        const double perSimdInstrPerS = 1e9 / r.nsPerInstr;                 // wave-instructions per second and SIMD
        const double nj = (r.watts - idleW) / (perSimdInstrPerS * 1024.0) * 1e9;
        printf("%-52s %9.3f %8.0f %7.0f %9.2f %12.2f\n", name, r.nsPerInstr, r.mhz, r.watts, r.nsPerInstr * r.mhz * 1e-3, nj);
        fflush(stdout);
    };
    for (int rep = 0; rep < 2; ++rep) {
        line("v_mul_f32 (plain fp32)", run<0>(hw, d, 48, secs));
        line("v_add_f32", run<1>(hw, d, 48, secs));
        line("v_mov_b32", run<9>(hw, d, 48, secs));
        line("v_fma_f32", run<2>(hw, d, 48, secs));
        line("v_med3_f32", run<3>(hw, d, 48, secs));
        line("v_cmp_lt_f32 + v_cndmask_b32", run<10>(hw, d, 48, secs));
        line("v_cvt_f64_f32", run<4>(hw, d, 48, secs));
        line("v_cvt_f32_f64", run<5>(hw, d, 48, secs));
        line("v_fma_f64", run<6>(hw, d, 48, secs));
        line("v_add_f64", run<7>(hw, d, 48, secs));
        line("v_mul_f64", run<8>(hw, d, 48, secs));
        printf("\n");
    }
    printf("config5's mix: a group = 2 MACS (mul, add, med3 each) + 1 INTERP (mul + its fp64 arithmetic + med3)\n");
    printf("%-52s %9s %8s %7s %12s %14s %16s\n", "form of the INTERP", "ns/group", "MHz", "W", "clk/group", "G groups/s", "uJ per M groups");
    auto group = [&](const char* name, const Result& r, int perGroup) {
        const double nsGroup = r.nsPerInstr * perGroup;
        const double groupsPerS = 1e9 / nsGroup * 1024.0;                   // all SIMDs
        printf("%-52s %9.2f %8.0f %7.0f %12.1f %14.2f %16.1f\n", name, nsGroup, r.mhz, r.watts, nsGroup * r.mhz * 1e-3, groupsPerS / 1e9, (r.watts - idleW) / groupsPerS * 1e12);
        fflush(stdout);
    };
    for (int rep = 0; rep < 2; ++rep) {
        group("as generated: cvt, cvt, fma_f64, cvt in a row (12)", run<20>(hw, d, 48, secs), 12);
        group("the quartet spread between the plain ones (12)", run<21>(hw, d, 48, secs), 12);
        group("quartet replaced by 14 plain fp32 (22)", run<22>(hw, d, 88, secs), 22);
        group("no INTERP arithmetic at all (8): the floor", run<23>(hw, d, 32, secs), 8);
        printf("\n");
    }
    return 0;
}
