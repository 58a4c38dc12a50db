// valu_rate.hip — what does one wave64 VALU instruction of each kind cost per SIMD, by the number of waves resident on the
// SIMD, and what clock does the chip hold meanwhile?  Straight-line bodies of 128 instructions (8 independent chains per
// wave) in a loop; HIP events for the time, s_memtime / s_memrealtime inside each wave for the shader clock.
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate && tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define F8 "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
#define D8 "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
#define REP16(body) ".rept 16\n" body ".endr\n"

template <int KIND>
__global__ void __launch_bounds__(64) k(float* out, uint64_t* clocks, int iters) {
    const int lane = threadIdx.x;
    float f0 = 0.5f + 0.001f * lane, f1 = f0 + 0.01f, f2 = f0 + 0.02f, f3 = f0 + 0.03f, f4 = f0 + 0.04f, f5 = f0 + 0.05f, f6 = f0 + 0.06f, f7 = f0 + 0.07f;
    double d0 = f0, d1 = f1, d2 = f2, d3 = f3, d4 = f4, d5 = f5, d6 = f6, d7 = f7;
    const float one = 1.0f, zero = 0.0f;
    const double oned = 1.0, zerod = 0.0;
    uint64_t t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0)
            asm volatile(REP16("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n") : F8 : "v"(one));
        else if (KIND == 1)
            asm volatile(REP16("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n") : F8 : "v"(zero));
        else if (KIND == 2)
            asm volatile(REP16("v_med3_f32 %0, %0, -1.0, 1.0\n v_med3_f32 %1, %1, -1.0, 1.0\n v_med3_f32 %2, %2, -1.0, 1.0\n v_med3_f32 %3, %3, -1.0, 1.0\n v_med3_f32 %4, %4, -1.0, 1.0\n v_med3_f32 %5, %5, -1.0, 1.0\n v_med3_f32 %6, %6, -1.0, 1.0\n v_med3_f32 %7, %7, -1.0, 1.0\n") : F8);
        else if (KIND == 3)
            asm volatile(REP16("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n") : F8 : "v"(one), "v"(zero));
        else if (KIND == 4)
            asm volatile(REP16("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %9\n v_cvt_f64_f32 %6, %10\n v_cvt_f64_f32 %7, %11\n") : D8 : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
        else if (KIND == 5)
            asm volatile(REP16("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n v_cvt_f32_f64 %4, %8\n v_cvt_f32_f64 %5, %9\n v_cvt_f32_f64 %6, %10\n v_cvt_f32_f64 %7, %11\n") : F8 : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
        else if (KIND == 6)
            asm volatile(REP16("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n") : D8 : "v"(oned), "v"(zerod));
        else if (KIND == 7)
            asm volatile(REP16("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n") : D8 : "v"(zerod));
        else if (KIND == 8)
            asm volatile(REP16("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n") : D8 : "v"(oned));
        else if (KIND == 9)
            // the config5 mix, 12 per 3 emulated instructions: 2 MACS (mul, add, med3) + 1 INTERP (mul, cvt, cvt, fma_f64, cvt back, med3)
            asm volatile(".rept 10\n"
                         "v_mul_f32 %0, %11, %0\n v_add_f32 %0, %12, %0\n v_med3_f32 %0, %0, -1.0, 1.0\n"
                         "v_mul_f32 %1, %11, %1\n v_add_f32 %1, %12, %1\n v_med3_f32 %1, %1, -1.0, 1.0\n"
                         "v_mul_f32 %2, %11, %2\n v_cvt_f64_f32 %8, %0\n v_cvt_f64_f32 %9, %2\n v_fma_f64 %8, %8, %10, %9\n v_cvt_f32_f64 %3, %8\n v_med3_f32 %3, %3, -1.0, 1.0\n"
                         ".endr\n" : F8, "+v"(d0), "+v"(d1) : "v"(oned), "v"(one), "v"(zero));
        else if (KIND == 10)
            asm volatile(REP16("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n") : D8 : "v"(oned));
#define BIN8(op, src) op " %0, " src ", %0\n" op " %1, " src ", %1\n" op " %2, " src ", %2\n" op " %3, " src ", %3\n" op " %4, " src ", %4\n" op " %5, " src ", %5\n" op " %6, " src ", %6\n" op " %7, " src ", %7\n"
#define UN8(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define CMP8(op, dst, src) op " " dst ", " src ", %0\n" op " " dst ", " src ", %1\n" op " " dst ", " src ", %2\n" op " " dst ", " src ", %3\n" op " " dst ", " src ", %4\n" op " " dst ", " src ", %5\n" op " " dst ", " src ", %6\n" op " " dst ", " src ", %7\n"
#define TRI8(op, a, b) op " %0, %0, " a ", " b "\n" op " %1, %1, " a ", " b "\n" op " %2, %2, " a ", " b "\n" op " %3, %3, " a ", " b "\n" op " %4, %4, " a ", " b "\n" op " %5, %5, " a ", " b "\n" op " %6, %6, " a ", " b "\n" op " %7, %7, " a ", " b "\n"
        else if (KIND == 11) asm volatile(REP16(BIN8("v_sub_f32", "%8")) : F8 : "v"(zero));
        else if (KIND == 12) asm volatile(REP16(BIN8("v_max_f32", "%8")) : F8 : "v"(zero));
        else if (KIND == 13) asm volatile(REP16(BIN8("v_mul_f32", "0x3f7fff00")) : F8);                   // 32-bit literal: 8-byte encoding
        else if (KIND == 14) asm volatile(REP16(BIN8("v_mul_f32", "%8")) : F8 : "s"(one));                // SGPR source
        else if (KIND == 15) asm volatile(REP16(UN8("v_cvt_i32_f32")) : F8);
        else if (KIND == 16) asm volatile(REP16(BIN8("v_lshlrev_b32", "3")) : F8);
        else if (KIND == 17) asm volatile(REP16(CMP8("v_cmp_ge_f32", "vcc", "%8")) : F8 : "v"(zero) : "vcc");
        else if (KIND == 18) asm volatile(REP16(CMP8("v_cmp_lt_f32_e64", "s[20:21]", "%8")) : F8 : "v"(zero) : "s20", "s21");
        else if (KIND == 19) asm volatile(REP16(UN8("v_mov_b32")) : F8);
        else if (KIND == 20) asm volatile(REP16(TRI8("v_med3_i32", "0", "63")) : F8);
        else if (KIND == 21) asm volatile(REP16(TRI8("v_cndmask_b32_e64", "%8", "vcc")) : F8 : "v"(zero) : "vcc");
        else if (KIND == 22) asm volatile(REP16(TRI8("v_bfi_b32", "%8", "%9")) : F8 : "v"(zero), "v"(one));
        else if (KIND == 23) asm volatile(REP16(BIN8("v_add_u32", "1")) : F8);
        else if (KIND == 24) asm volatile(REP16(BIN8("v_min_f32", "1.0")) : F8);
        else if (KIND == 25) asm volatile(REP16(TRI8("v_max3_f32", "%8", "%9")) : F8 : "v"(zero), "v"(one));
        else if (KIND == 26) asm volatile(REP16(TRI8("v_med3_f32", "%8", "%9")) : F8 : "v"(zero), "v"(one));  // no constants
        else if (KIND == 27) asm volatile(REP16(TRI8("v_add_f32_e64", "%8", "clamp")) : F8 : "v"(zero));       // VOP3 form + clamp
        else if (KIND == 28) asm volatile(REP16(BIN8("v_mul_f32", "0.5")) : F8);                                  // inline constant
        else if (KIND == 29) asm volatile(REP16(TRI8("v_fma_f32", "%8", "%9")) : F8 : "s"(one), "v"(zero));         // one SGPR source
        else if (KIND == 30) asm volatile(REP16(TRI8("v_fma_f32", "%8", "%8")) : F8 : "s"(one));                    // the same SGPR twice
        else if (KIND == 31) asm volatile(REP16(TRI8("v_fma_f32", "1.0", "0")) : F8);                              // inline constants
        else if (KIND == 32) asm volatile(REP16(BIN8("v_add_f32", "0x38d1b717")) : F8);                             // literal
        else if (KIND == 33) asm volatile(REP16(TRI8("v_lshl_add_u32", "3", "%8")) : F8 : "v"(zero));
        else if (KIND == 34) asm volatile(REP16(BIN8("v_and_b32", "%8")) : F8 : "v"(one));
        else if (KIND == 35) asm volatile(REP16(TRI8("v_mul_f32_e64", "%8", "mul:2")) : F8 : "v"(one));             // VOP3 + output modifier
        else if (KIND == 36) asm volatile(REP16(BIN8("v_sub_u32", "%8")) : F8 : "v"(zero));
        else if (KIND == 37) asm volatile(REP16(BIN8("v_subrev_f32", "%8")) : F8 : "v"(zero));
        else if (KIND == 38) asm volatile(REP16(BIN8("v_fmac_f32", "%8")) : F8 : "v"(zero));
        else if (KIND == 39) asm volatile(REP16(BIN8("v_mul_f32", "-%8")) : F8 : "v"(one));                         // source modifier (VOP3 encoding)
        else if (KIND == 40) asm volatile(REP16("v_add_co_u32 %0, vcc, %8, %0\n v_add_co_u32 %1, vcc, %8, %1\n v_add_co_u32 %2, vcc, %8, %2\n v_add_co_u32 %3, vcc, %8, %3\n v_add_co_u32 %4, vcc, %8, %4\n v_add_co_u32 %5, vcc, %8, %5\n v_add_co_u32 %6, vcc, %8, %6\n v_add_co_u32 %7, vcc, %8, %7\n") : F8 : "v"(zero) : "vcc");
        else if (KIND == 41) asm volatile(REP16(CMP8("v_cmp_le_u32", "vcc", "%8")) : F8 : "v"(zero) : "vcc");
        else if (KIND == 42) asm volatile(REP16(UN8("v_trunc_f32")) : F8);
        else if (KIND == 43) asm volatile(REP16(BIN8("v_or_b32", "%8")) : F8 : "v"(zero));
        else if (KIND == 44) asm volatile(REP16(BIN8("v_xor_b32", "%8")) : F8 : "v"(zero));
        else if (KIND == 45) asm volatile(REP16(BIN8("v_mul_u32_u24", "%8")) : F8 : "v"(one));
        else if (KIND == 46) asm volatile(REP16(BIN8("v_lshrrev_b32", "1")) : F8);
    }
    uint64_t t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    out[blockIdx.x * 64 + lane] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if (lane == 0) { clocks[blockIdx.x * 2] = t1 - t0; clocks[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int KIND>
void run(const char* name, int perIter, float* d, uint64_t* dc, int waves, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<uint64_t> h(waves * 2);
    hipLaunchKernelGGL(k<KIND>, dim3(waves), dim3(64), 0, 0, d, dc, iters / 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(waves), dim3(64), 0, 0, d, dc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), dc, waves * 16, hipMemcpyDeviceToHost);
    double sc = 0, rt = 0;
    for (int w = 0; w < waves; ++w) { sc += (double)h[w * 2]; rt += (double)h[w * 2 + 1]; }
    const double perSimd = (double)iters * perIter * (waves / 1024.0);   // wave-instructions per SIMD
    const double mhz = sc / rt * 100.0;                                  // s_memtime ticks per s_memrealtime tick (100 MHz)
    printf("%-16s %4.1f waves/SIMD: %7.2f ms  %.3f ns per wave-instruction per SIMD at %4.0f MHz = %.2f clocks (waves resident %3.0f %% of the launch)\n",
           name, waves / 1024.0, ms, ms * 1e6 / perSimd, mhz, ms * 1e-3 * mhz * 1e6 / perSimd, 100.0 * (rt / waves) * 1e-5 / ms);
}

int main() {
    float* d; uint64_t* dc;
    hipMalloc(&d, 8192 * 64 * 4);
    hipMalloc(&dc, 8192 * 16);
    for (int waves : {1024, 2048, 4096, 8192}) {
        const int iters = 40000 * 1024 / waves * 4;
        run<0>("v_mul_f32", 128, d, dc, waves, iters);
        run<1>("v_add_f32", 128, d, dc, waves, iters);
        run<2>("v_med3_f32", 128, d, dc, waves, iters);
        run<3>("v_fma_f32", 128, d, dc, waves, iters);
        run<10>("v_pk_mul_f32", 128, d, dc, waves, iters);
        run<4>("v_cvt_f64_f32", 128, d, dc, waves, iters);
        run<5>("v_cvt_f32_f64", 128, d, dc, waves, iters);
        run<6>("v_fma_f64", 128, d, dc, waves, iters);
        run<7>("v_add_f64", 128, d, dc, waves, iters);
        run<8>("v_mul_f64", 128, d, dc, waves, iters);
        run<9>("config5 mix", 120, d, dc, waves, iters);
        if (waves != 4096) continue;
        run<11>("v_sub_f32", 128, d, dc, waves, iters);
        run<12>("v_max_f32", 128, d, dc, waves, iters);
        run<24>("v_min_f32 (1.0)", 128, d, dc, waves, iters);
        run<13>("v_mul_f32 literal", 128, d, dc, waves, iters);
        run<14>("v_mul_f32 sgpr", 128, d, dc, waves, iters);
        run<15>("v_cvt_i32_f32", 128, d, dc, waves, iters);
        run<16>("v_lshlrev_b32", 128, d, dc, waves, iters);
        run<23>("v_add_u32", 128, d, dc, waves, iters);
        run<17>("v_cmp_ge_f32 vcc", 128, d, dc, waves, iters);
        run<18>("v_cmp_lt_f32 sgpr", 128, d, dc, waves, iters);
        run<19>("v_mov_b32", 128, d, dc, waves, iters);
        run<20>("v_med3_i32", 128, d, dc, waves, iters);
        run<21>("v_cndmask_b32", 128, d, dc, waves, iters);
        run<22>("v_bfi_b32", 128, d, dc, waves, iters);
        run<25>("v_max3_f32", 128, d, dc, waves, iters);
        run<26>("v_med3_f32 vgprs", 128, d, dc, waves, iters);
        run<27>("v_add_f32 clamp", 128, d, dc, waves, iters);
        run<28>("v_mul_f32 inline", 128, d, dc, waves, iters);
        run<32>("v_add_f32 literal", 128, d, dc, waves, iters);
        run<29>("v_fma_f32 sgpr", 128, d, dc, waves, iters);
        run<30>("v_fma_f32 sgpr x2", 128, d, dc, waves, iters);
        run<31>("v_fma_f32 inline", 128, d, dc, waves, iters);
        run<33>("v_lshl_add_u32", 128, d, dc, waves, iters);
        run<34>("v_and_b32", 128, d, dc, waves, iters);
        run<35>("v_mul_f32 mul:2", 128, d, dc, waves, iters);
        run<36>("v_sub_u32", 128, d, dc, waves, iters);
        run<37>("v_subrev_f32", 128, d, dc, waves, iters);
        run<38>("v_fmac_f32", 128, d, dc, waves, iters);
        run<39>("v_mul_f32 neg", 128, d, dc, waves, iters);
        run<40>("v_add_co_u32", 128, d, dc, waves, iters);
        run<41>("v_cmp_le_u32", 128, d, dc, waves, iters);
        run<42>("v_trunc_f32", 128, d, dc, waves, iters);
        run<43>("v_or_b32", 128, d, dc, waves, iters);
        run<44>("v_xor_b32", 128, d, dc, waves, iters);
        run<45>("v_mul_u32_u24", 128, d, dc, waves, iters);
        run<46>("v_lshrrev_b32", 128, d, dc, waves, iters);
    }
    return 0;
}
