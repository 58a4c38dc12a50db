// fx_kernel.hip — the FX8010 interpreter kernel for gfx950 (MI355X, CDNA4).
//
// One lane = one emulated DSP.  A wavefront (64 lanes = 64 consecutive instances) walks the
// host-decoded opcode stream; every record is fetched with one scalar load and dispatched
// with a wave-uniform branch, so the only per-lane control flow is the SKIP predicate.
//
//   per-lane register file : LDS, row r of lane l at byte r*256 + l*4 (bank = lane: conflict-free)
//   CCR                    : LDS row 0, written only where the decoder proved it observable
//   uniform registers      : immediates inside the record (SGPRs)
//   TRAM                   : HBM, [wave][slot][64]; a wave's read/write of one slot is 256 contiguous bytes
//   PCM in/out             : HBM, [sample][channel][instance]; next sample's input is prefetched
//   cursors, LFSR, skip    : VGPRs
//
// Arithmetic is the reference's (source/FX8010.cpp:1023-1249): IEEE fp32 with fp64 in
// INTERP/LOG/EXP, multiply and add never fused, fp32 denormals kept, x86 float->int
// conversion semantics reproduced by cvtt_*.  Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "fx_decode.hpp"
#include "fx_kernel.hpp"

#pragma clang fp contract(off)

namespace fx {
namespace {

struct LaneState {
    int numSkip;
    int iw, ir, xw, xr;    // TRAM cursors (reference: include/FX8010.h:214-217)
    int g1, g2;            // white-noise generator (include/FX8010.h:290-291)
    uint32_t ood;
    uint32_t dynCount;     // executed instructions among the shadowed ones
    bool ran;              // did the last counted instruction execute on this lane
    bool alive, isEnd;     // multipass bookkeeping
};

__device__ __forceinline__ float asF(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t asU(float f) { return __float_as_uint(f); }

// x86 cvttss2si / cvttsd2si: NaN and out-of-range give 0x80000000
__device__ __forceinline__ int cvtt_f32(float v) {
    const bool bad = !(v < 2147483648.0f) || (v < -2147483648.0f);
    const int r = (int)(bad ? 0.0f : v);
    return bad ? (int)0x80000000 : r;
}
__device__ __forceinline__ int cvtt_f64(double v) {
    const bool bad = !(v < 2147483648.0) || (v <= -2147483649.0);
    const int r = (int)(bad ? 0.0 : v);
    return bad ? (int)0x80000000 : r;
}

// reference saturate(x, 1.0f), source/FX8010.cpp:275-279 (NaN passes through)
__device__ __forceinline__ float saturate1(float v) { return (v >= 1.0f) ? 1.0f : ((v <= -1.0f) ? -1.0f : v); }

// reference setCCR, source/FX8010.cpp:211-232 — the value CCR takes for result r
__device__ __forceinline__ float ccrOf(float r) {
    float c = 0.0f;
    c = (r == -1.0f) ? 20.0f : c;
    c = (r == 1.0f) ? 16.0f : c;
    c = (r > 0.0f && r < 1.0f) ? 2.0f : c;
    c = (r < 0.0f && r > -1.0f) ? 6.0f : c;
    c = (r == 0.0f) ? 8.0f : c;
    return c;
}

// reference wrapAround, source/FX8010.cpp:299-328 (its CCR side effect is overwritten by setCCR)
__device__ __forceinline__ float wrapAround(float a) { return (a >= 1.0f) ? (a - 2.0f) : ((a < -1.0f) ? (a + 2.0f) : a); }

// reference logicOps, source/FX8010.cpp:330-360
__device__ __forceinline__ int logicOps(float a_, float x_, float y_) {
    const int A = cvtt_f32(a_), X = cvtt_f32(x_), Y = cvtt_f32(y_);
    int r = (A & X) ^ Y;
    r = (Y == 0xFFFFFF) ? (~A & X) : r;
    r = (Y == ~X) ? (A | Y) : r;
    r = (X == 0xFFFFFFF && Y == 0xFFFFFF) ? ~A : r;
    r = (X == 0xFFFFFF) ? (A ^ Y) : r;
    r = (Y == 0) ? (A & X) : r;
    return r;
}

struct Ctx {
    char* lds;            // this lane's LDS base: wave region + lane*4
    const KernelArgs* a;
    float* itramLane;     // itram + wave*iSlots*64 + lane
    float* xtramLane;
};

__device__ __forceinline__ float ldsRead(const Ctx& c, uint32_t off) { return *reinterpret_cast<const float*>(c.lds + off); }
__device__ __forceinline__ void ldsWrite(const Ctx& c, uint32_t off, float v) { *reinterpret_cast<float*>(c.lds + off) = v; }

// reference linearInterpolate, source/FX8010.cpp:283-296 with x_min=-1, x_max=1 and a 64-entry table
__device__ __forceinline__ float lutInterpolate(const double* __restrict__ tbl, float xin, uint32_t& ood) {
    const double x = (double)xin;
    const double step = (1.0 - -1.0) / 63.0;
    int idx = cvtt_f64((x - -1.0) / step);
    if (idx < 0 || idx > 63) { ood |= OOD_LUT_INDEX; idx = idx < 0 ? 0 : 63; }
    const double x1 = -1.0 + idx * step;
    const double x2 = -1.0 + (idx + 1) * step;
    const double y1 = tbl[idx];
    const double y2 = tbl[idx + 1];
    const double y = (y2 - y1) / (x2 - x1) * (x - x1) + y1;
    return (float)y;
}

// reference read/write{Small,Large}Delay, source/FX8010.cpp:909-967
__device__ __forceinline__ float tramRead(float* __restrict__ base, int slots, int size, int& rpos, int position, uint32_t& ood) {
    if (size <= 0) { ood |= OOD_TRAM_SIZE0; return 0.0f; }
    position = position > size - 1 ? size - 1 : position;
    position = position < 0 ? 0 : position;
    int idx = rpos - position;               // (rpos - p) % size with 0 <= rpos < size
    if (idx < 0) { ood |= OOD_TRAM_READ_NEG; idx += size; }
    const float v = (idx < slots) ? base[(size_t)idx * 64] : 0.0f;
    rpos = (rpos + 1 >= size) ? 0 : rpos + 1;
    return v;
}
__device__ __forceinline__ void tramWrite(float* __restrict__ base, int slots, int size, int refCap, int& wpos, int position, float v, uint32_t& ood) {
    if (size <= 0) { ood |= OOD_TRAM_SIZE0; return; }
    position = position > size - 1 ? size - 1 : position;
    position = position < 0 ? 0 : position;
    const int idx = wpos + position;         // the reference applies no modulo here
    if (idx >= refCap || idx >= slots) ood |= OOD_TRAM_WRITE_OOB;
    else base[(size_t)idx * 64] = v;
    wpos = (wpos + 1 >= size) ? 0 : wpos + 1;
}

// Execute one record for the lanes currently enabled.
template <bool MULTIPASS>
__device__ __forceinline__ void execOp(const Ctx& c, LaneState& st, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3,
                                       uint32_t w4, uint32_t w5, uint32_t w6) {
    const uint32_t rOff = w1 & 0xffffu, aOff = w1 >> 16, xOff = w2 & 0xffffu, yOff = w2 >> 16;
    auto opA = [&]() { return (w0 & F_UA) ? asF(w4) : ldsRead(c, aOff); };
    auto opX = [&]() { return (w0 & F_UX) ? asF(w5) : ldsRead(c, xOff); };
    auto opY = [&]() { return (w0 & F_UY) ? asF(w6) : ldsRead(c, yOff); };
    auto finish = [&](float r) {  // store R, then setCCR(R) where observable
        ldsWrite(c, rOff, r);
        if (w0 & F_CCR) ldsWrite(c, 0, ccrOf(r));
    };
    switch (w0 & 0xffu) {
        case H_MACS: {  // R = sat(A + X*Y)            FX8010.cpp:1077-1085 (MACINTS :1095-1103 is identical)
            const float a = opA(), x = opX(), y = opY();
            const float p = x * y;
            finish(saturate1(a + p));
            break;
        }
        case H_MACSN: {  // R = sat(A - X*Y)           :1086-1094
            const float a = opA(), x = opX(), y = opY();
            const float p = x * y;
            finish(saturate1(a - p));
            break;
        }
        case H_ACC3: {  // R = sat((A + X) + Y)        :1104-1112
            const float a = opA(), x = opX(), y = opY();
            const float t = a + x;
            finish(saturate1(t + y));
            break;
        }
        case H_INTERP: {  // R = sat((float)((1.0 - X)*A + (double)(X*Y)))   :1180-1187
            const float a = opA(), x = opX(), y = opY();
            const float p = x * y;
            const double d = (1.0 - (double)x) * (double)a + (double)p;
            finish(saturate1((float)d));
            break;
        }
        case H_MACW: {  // R = A + wrap(X*Y)           :1126-1131
            const float a = opA(), x = opX(), y = opY();
            finish(a + wrapAround(x * y));
            break;
        }
        case H_MACWN: {  // R = A - wrap(X*Y)          :1132-1137
            const float a = opA(), x = opX(), y = opY();
            finish(a - wrapAround(x * y));
            break;
        }
        case H_MACINTW: {  // R = wrap(A + X*Y)        :1138-1143
            const float a = opA(), x = opX(), y = opY();
            const float p = x * y;
            finish(wrapAround(a + p));
            break;
        }
        case H_MACMV: {  // R = A (the accumulator it feeds is never observable)   :1144-1149
            finish(opA());
            break;
        }
        case H_ANDXOR: {  // :1150-1154
            const float a = opA(), x = opX(), y = opY();
            finish((float)logicOps(a, x, y));
            break;
        }
        case H_TSTNEG: {  // R = A >= Y ? X : intToFloat(~floatToInt(X))   :1155-1162, :1009-1020
            const float a = opA(), x = opX(), y = opY();
            const int xi = cvtt_f32(x * 2147483648.0f);
            const float neg = (float)(~xi) * 4.656612873077392578125e-10f;  // exact /2^31
            finish(a >= y ? x : neg);
            break;
        }
        case H_LIMIT: {  // R = A >= Y ? X : Y         :1163-1168
            const float a = opA(), x = opX(), y = opY();
            finish(a >= y ? x : y);
            break;
        }
        case H_LIMITN: {  // R = A < Y ? X : Y         :1169-1174
            const float a = opA(), x = opX(), y = opY();
            finish(a < y ? x : y);
            break;
        }
        case H_LOG:
        case H_EXP: {  // R = (float)linearInterpolate(A, table[(int)X])   :1113-1125; no clamp
            const float a = opA();
            int t;
            if (w0 & F_UX) {
                t = (int)w3;
                if (w0 & F_STATIC_OOD) st.ood |= OOD_LUT_TABLE;
            } else {
                t = cvtt_f32(ldsRead(c, xOff));
                if (t < 0 || t > 31) { st.ood |= OOD_LUT_TABLE; t = t < 0 ? 0 : 31; }
                t += ((w0 & 0xffu) == H_EXP) ? 32 : 0;
            }
            finish(lutInterpolate(c.a->lut + (size_t)t * 65, a, st.ood));
            break;
        }
        case H_SKIP: {  // if ((float)(int)X == CCR) numSkip = (int)Y   :1175-1179
            const float x = opX(), y = opY();
            const float ccr = ldsRead(c, 0);
            if ((float)cvtt_f32(x) == ccr) st.numSkip = cvtt_f32(y);
            break;
        }
        case H_TRAM_IR: {  // A = readSmallDelay((int)Y)   :1188-1193
            const int p = cvtt_f32(opY());
            ldsWrite(c, aOff, tramRead(c.itramLane, c.a->iSlots, c.a->iSize, st.ir, p, st.ood));
            break;
        }
        case H_TRAM_IW: {  // writeSmallDelay(A, (int)Y)   :1194-1198
            const float v = opA();
            const int p = cvtt_f32(opY());
            tramWrite(c.itramLane, c.a->iSlots, c.a->iSize, kMaxITram, st.iw, p, v, st.ood);
            break;
        }
        case H_TRAM_XR: {  // :1200-1205
            const int p = cvtt_f32(opY());
            ldsWrite(c, aOff, tramRead(c.xtramLane, c.a->xSlots, c.a->xSize, st.xr, p, st.ood));
            break;
        }
        case H_TRAM_XW: {  // :1206-1210
            const float v = opA();
            const int p = cvtt_f32(opY());
            tramWrite(c.xtramLane, c.a->xSlots, c.a->xSize, kMaxXTram, st.xw, p, v, st.ood);
            break;
        }
        case H_REFRESH:  // INPUT operand <- this sample's input of A's channel   :1053-1061
        case H_LATCH:    // outputBuffer[R.IOIndex] = R                           :1229-1233
            ldsWrite(c, rOff, ldsRead(c, aOff));
            break;
        case H_NOISE: {  // whitenoise()   :993-1000
            st.g1 ^= st.g2;
            const float nz = (float)st.g2 * 4.656612873077392578125e-10f;  // g_fScale = 2.0f/0xffffffff = 2^-31
            st.g2 = (int)((uint32_t)st.g2 + (uint32_t)st.g1);
            ldsWrite(c, rOff, nz);
            break;
        }
        case H_END:  // :1212-1215
            if (MULTIPASS) st.isEnd = true;
            break;
        default:  // H_NOP: counted, does nothing (e.g. idelay whose R is neither read nor write)
            break;
    }
}

template <bool MULTIPASS>
__global__ __launch_bounds__(64) void fx_step_block(const KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const long long wave = blockIdx.x;
    const long long inst = wave * 64 + lane;
    const bool live = inst < a.n;

    Ctx c;
    c.lds = smem + lane * 4;
    c.a = &a;
    c.itramLane = a.itram ? a.itram + (size_t)wave * a.iSlots * 64 + lane : nullptr;
    c.xtramLane = a.xtram ? a.xtram + (size_t)wave * a.xSlots * 64 + lane : nullptr;

    uint32_t* __restrict__ stLane = a.state + inst;  // rows are nPad apart; inst < nPad always
    const size_t rowStride = (size_t)a.nPad;

    // ---- prologue: state -> LDS rows / VGPRs
    for (int i = 0; i < a.nLoad; ++i) {
        const uint32_t e = a.rowTable[i];
        ldsWrite(c, (e & 0xffffu) * 256u, asF(stLane[(size_t)(e >> 16) * rowStride]));
    }
    LaneState st;
    st.numSkip = 0;
    st.iw = (int)stLane[(size_t)(a.cursorBase + 0) * rowStride];
    st.ir = (int)stLane[(size_t)(a.cursorBase + 1) * rowStride];
    st.xw = (int)stLane[(size_t)(a.cursorBase + 2) * rowStride];
    st.xr = (int)stLane[(size_t)(a.cursorBase + 3) * rowStride];
    st.g1 = (int)stLane[(size_t)(a.noiseBase + 0) * rowStride];
    st.g2 = (int)stLane[(size_t)(a.noiseBase + 1) * rowStride];
    st.ood = 0;
    st.dynCount = 0;
    st.ran = true;
    st.alive = true;
    st.isEnd = false;

    const size_t n = (size_t)a.n;
    const int CH = a.channels;
    float nextIn[kMaxChannels];
#pragma unroll
    for (int ch = 0; ch < kMaxChannels; ++ch)
        nextIn[ch] = (ch < CH && a.inRow[ch] >= 0 && live && a.nSamples > 0) ? a.in[(size_t)ch * n + inst] : 0.0f;

    for (int s = 0; s < a.nSamples; ++s) {
#pragma unroll
        for (int ch = 0; ch < kMaxChannels; ++ch)
            if (ch < CH && a.inRow[ch] >= 0) ldsWrite(c, (uint32_t)a.inRow[ch] * 256u, nextIn[ch]);
        if (s + 1 < a.nSamples) {
#pragma unroll
            for (int ch = 0; ch < kMaxChannels; ++ch)
                if (ch < CH && a.inRow[ch] >= 0 && live) nextIn[ch] = a.in[((size_t)(s + 1) * CH + ch) * n + inst];
        }

        const uint32_t* __restrict__ prog = (s == a.nSamples - 1) ? a.last : a.steady;
        st.numSkip = 0;  // reference: local to process(), FX8010.cpp:1030
        if (MULTIPASS) { st.alive = true; st.isEnd = false; }
        int passes = 0;
        bool again;
        do {
            for (int pc = 0; pc < a.nOps; ++pc) {
                const uint32_t* __restrict__ rec = prog + (size_t)pc * 8;
                const uint32_t w0 = rec[0], w1 = rec[1], w2 = rec[2], w3 = rec[3], w4 = rec[4], w5 = rec[5], w6 = rec[6];
                if (w0 & F_SHADOW) {
                    bool run;
                    if (w0 & F_POSTFIX) run = st.ran;
                    else {
                        run = (st.numSkip == 0);
                        if (MULTIPASS) run = run && st.alive;
                        if (!(w0 & F_PREFIX)) {
                            // a skipped instruction only counts the skip down (FX8010.cpp:1235-1241)
                            const bool countDown = MULTIPASS ? (st.alive && !run) : !run;
                            if (countDown) st.numSkip = st.numSkip > 0 ? st.numSkip - 1 : 0;
                            st.ran = run;
                            st.dynCount += run ? 1u : 0u;
                        }
                    }
                    if (run) execOp<MULTIPASS>(c, st, w0, w1, w2, w3, w4, w5, w6);
                } else {
                    execOp<MULTIPASS>(c, st, w0, w1, w2, w3, w4, w5, w6);
                }
            }
            again = false;
            if (MULTIPASS) {
                st.alive = st.alive && !st.isEnd;
                ++passes;
                again = __any(st.alive) && passes < kPassCap;
                if (!again && st.alive) st.ood |= OOD_PASS_CAP;
            }
        } while (again);

        if (live) {
#pragma unroll
            for (int ch = 0; ch < kMaxChannels; ++ch)
                if (ch < CH) a.out[((size_t)s * CH + ch) * n + inst] = ldsRead(c, (uint32_t)a.latchRow[ch] * 256u);
        }
    }

    // ---- epilogue: LDS rows / VGPRs -> state
    for (int i = 0; i < a.nStore; ++i) {
        const uint32_t e = a.rowTable[a.nLoad + i];
        stLane[(size_t)(e >> 16) * rowStride] = asU(ldsRead(c, (e & 0xffffu) * 256u));
    }
    stLane[(size_t)(a.cursorBase + 0) * rowStride] = (uint32_t)st.iw;
    stLane[(size_t)(a.cursorBase + 1) * rowStride] = (uint32_t)st.ir;
    stLane[(size_t)(a.cursorBase + 2) * rowStride] = (uint32_t)st.xw;
    stLane[(size_t)(a.cursorBase + 3) * rowStride] = (uint32_t)st.xr;
    stLane[(size_t)(a.noiseBase + 0) * rowStride] = (uint32_t)st.g1;
    stLane[(size_t)(a.noiseBase + 1) * rowStride] = (uint32_t)st.g2;
    stLane[(size_t)a.oodRow * rowStride] |= st.ood;
    const unsigned long long add = (unsigned long long)a.staticCount * (unsigned long long)a.nSamples + st.dynCount;
    unsigned long long cnt = ((unsigned long long)stLane[(size_t)a.countHi * rowStride] << 32) | stLane[(size_t)a.countLo * rowStride];
    cnt += add;
    stLane[(size_t)a.countLo * rowStride] = (uint32_t)cnt;
    stLane[(size_t)a.countHi * rowStride] = (uint32_t)(cnt >> 32);
}

__global__ void fx_fill_rows(uint32_t* state, long long nPad, const uint32_t* rows, const uint32_t* values, int nRows) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPad) return;
    for (int r = blockIdx.y; r < nRows; r += gridDim.y) state[(size_t)rows[r] * (size_t)nPad + i] = values[r];
}

__global__ void fx_reduce_row(const uint32_t* state, long long nPad, long long n, int rowLo, int rowHi, int rowOr,
                              unsigned long long* sum, uint32_t* orOut) {
    unsigned long long s = 0;
    uint32_t o = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        s += ((unsigned long long)state[(size_t)rowHi * nPad + i] << 32) | state[(size_t)rowLo * nPad + i];
        o |= state[(size_t)rowOr * nPad + i];
    }
    for (int d = 32; d > 0; d >>= 1) {
        s += __shfl_down(s, d);
        o |= __shfl_down(o, d);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(sum, s);
        atomicOr(orOut, o);
    }
}

}  // namespace

hipError_t launchStepBlock(const KernelArgs& a, bool multipass, hipStream_t stream) {
    const unsigned grid = (unsigned)(a.nPad / 64);
    const size_t ldsBytes = (size_t)a.nRows * 256;
    auto kern = multipass ? fx_step_block<true> : fx_step_block<false>;
    if (ldsBytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), ldsBytes, stream, a);
    return hipGetLastError();
}

hipError_t launchFillRows(uint32_t* state, long long nPad, const uint32_t* d_rows, const uint32_t* d_values, int nRows,
                          hipStream_t stream) {
    if (nRows <= 0) return hipSuccess;
    dim3 grid((unsigned)((nPad + 255) / 256), (unsigned)(nRows < 64 ? nRows : 64));
    hipLaunchKernelGGL(fx_fill_rows, grid, dim3(256), 0, stream, state, nPad, d_rows, d_values, nRows);
    return hipGetLastError();
}

hipError_t launchReduceRow(const uint32_t* state, long long nPad, long long n, int rowLo, int rowHi, int rowOr,
                           unsigned long long* d_sum, uint32_t* d_or, hipStream_t stream) {
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fx_reduce_row, dim3((unsigned)blocks), dim3(256), 0, stream, state, nPad, n, rowLo, rowHi, rowOr, d_sum, d_or);
    return hipGetLastError();
}

}  // namespace fx
