// fx_kernel.hip — the FX8010 interpreter kernel for gfx950 (MI355X, CDNA4).
//
// One lane steps K emulated DSPs (K = 1, 2 or 4 consecutive instances); a wavefront therefore
// owns 64*K instances and walks the host-decoded opcode stream for all of them.  Every record is
// fetched with ONE scalar load (constant address space, prefetched one record ahead) and
// dispatched with wave-uniform branches, so the fixed cost of fetch/decode/dispatch is shared by
// 64*K instances and the only per-instance control flow is the SKIP predicate (a select).
//
//   per-instance register file : LDS, row r = [64 lanes][K] floats at byte r*256*K; a lane reads its
//                                K values of a row with one ds_read_b32/b64/b128 (conflict-free)
//   CCR                        : LDS row 0, written only where the decoder proved it observable
//   uniform registers          : immediates inside the record (SGPRs)
//   skip / cursors / LFSR      : LDS rows too, allocated only for programs that need them, so the
//                                dispatch loop carries no per-lane VGPR state
//   TRAM                       : HBM, [wave][slot][64][K]: a wave's access to one slot is 256*K
//                                contiguous bytes
//   PCM in/out                 : HBM, [sample][channel][instance]; next sample's input prefetched
//
// Arithmetic is the reference's (source/FX8010.cpp:1023-1249): IEEE fp32, fp64 inside
// INTERP/LOG/EXP, multiply and add never fused, fp32 denormals kept, x86 float->int conversion
// semantics reproduced by cvtt_*.  Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "fx_decode.hpp"
#include "fx_kernel.hpp"

#pragma clang fp contract(off)

namespace fx {
namespace {

typedef const uint32_t __attribute__((address_space(4))) * ConstU32;  // scalar-loadable stream

__device__ __forceinline__ float asF(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t asU(float f) { return __float_as_uint(f); }

// K floats of one lane
template <int K>
struct Vec {
    float v[K];
};

template <int K>
__device__ __forceinline__ Vec<K> splat(float f) {
    Vec<K> r;
#pragma unroll
    for (int k = 0; k < K; ++k) r.v[k] = f;
    return r;
}

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

// Arithmetic with a fixed SOURCE order (the compiler may commute a plain + or *): which NaN an instruction hands on when
// several operands are NaN depends on it - gfx950: the first source; the x86 build of the reference: the first operand
// of the SSE instruction g++ chose.  negated() is an exact sign change that leaves a NaN's sign alone (v_sub_f32 and neg
// modifiers flip it; x86 subss does not).  See tools/micro/nanrules.hip and tests/golden/nan_collisions.json.
__device__ __forceinline__ float addFirst(float first, float second) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(first), "v"(second));
    return r;
}
__device__ __forceinline__ float mulFirst(float first, float second) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(first), "v"(second));
    return r;
}
__device__ __forceinline__ float negated(float v) {
    float r;
    asm("v_mul_f32 %0, -1.0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ double addFirst64(double first, double second) {
    double r;
    asm("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(first), "v"(second));
    return r;
}
__device__ __forceinline__ double mulFirst64(double first, double second) {
    double r;
    asm("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(first), "v"(second));
    return r;
}
__device__ __forceinline__ double oneMinus(double x) {  // 1.0 - x in one rounding, a NaN keeps its sign
    double r;
    asm("v_fma_f64 %0, %1, -1.0, 1.0" : "=v"(r) : "v"(x));
    return r;
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

// ---- LDS access: a lane's K values of one row are contiguous (4*K bytes) ----
template <int K>
__device__ __forceinline__ Vec<K> ldsLoad(const char* lane, uint32_t off) {
    Vec<K> r;
    if constexpr (K == 1) {
        r.v[0] = *reinterpret_cast<const float*>(lane + off);
    } else if constexpr (K == 2) {
        const float2 t = *reinterpret_cast<const float2*>(lane + off);
        r.v[0] = t.x; r.v[1] = t.y;
    } else {
        const float4 t = *reinterpret_cast<const float4*>(lane + off);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    }
    return r;
}
template <int K>
__device__ __forceinline__ void ldsStore(char* lane, uint32_t off, const Vec<K>& x) {
    if constexpr (K == 1) {
        *reinterpret_cast<float*>(lane + off) = x.v[0];
    } else if constexpr (K == 2) {
        *reinterpret_cast<float2*>(lane + off) = make_float2(x.v[0], x.v[1]);
    } else {
        *reinterpret_cast<float4*>(lane + off) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
    }
}
// ---- global access of K consecutive floats (vector when the run is aligned and complete) ----
template <int K>
__device__ __forceinline__ Vec<K> gLoad(const float* __restrict__ p, bool vecOk, int nValid) {
    Vec<K> r;
    if constexpr (K == 1) {
        r.v[0] = nValid > 0 ? p[0] : 0.0f;
    } else {
        if (vecOk) {
            if constexpr (K == 2) { const float2 t = *reinterpret_cast<const float2*>(p); r.v[0] = t.x; r.v[1] = t.y; }
            else { const float4 t = *reinterpret_cast<const float4*>(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) r.v[k] = k < nValid ? p[k] : 0.0f;
        }
    }
    return r;
}
template <int K>
__device__ __forceinline__ void gStore(float* __restrict__ p, const Vec<K>& x, bool vecOk, int nValid) {
    if constexpr (K == 1) {
        if (nValid > 0) p[0] = x.v[0];
    } else {
        if (vecOk) {
            if constexpr (K == 2) *reinterpret_cast<float2*>(p) = make_float2(x.v[0], x.v[1]);
            else *reinterpret_cast<float4*>(p) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) if (k < nValid) p[k] = x.v[k];
        }
    }
}

// reference linearInterpolate, source/FX8010.cpp:283-296 (x_min=-1, x_max=1, 64-entry table), evaluated
// from the host-precomputed thresholds / segment table (fx_model.hpp, LutDevice): same IEEE results, no
// fp64 division in the kernel.  `seg` points at the table's 64 {slope, y1} pairs.
__device__ __forceinline__ float lutInterpolate(const double* __restrict__ lut, int table, float xin, uint32_t& ood) {
    const double x = (double)xin;
    const double t = x - -1.0;
    int idx = 0;
    if (!(t >= 0.0) || t > 2.0) {
        // outside [-1,1] (or NaN): outside the parity domain; follow the reference's arithmetic, then clamp
        // (the quotient is clamped as a number: NaN and anything below 0 -> 0, anything from 64 up, +Inf included, -> 63)
        const double q = t / ((1.0 - -1.0) / 63.0);
        idx = !(q >= 0.0) ? (q == q && q > -1.0 ? 0 : -1) : (q >= 64.0 ? 64 : cvtt_f64(q));
        if (idx < 0 || idx > 63) { ood |= OOD_LUT_INDEX; idx = idx < 0 ? 0 : 63; }
    } else {
        idx = (int)(t * 31.5);
        idx = idx > 63 ? 63 : idx;
        const double* thr = lut + kLutThrOff;
        idx += (t >= thr[idx + 1]) ? 1 : 0;
        idx -= (t < thr[idx]) ? 1 : 0;
    }
    const double x1 = lut[kLutX1Off + idx];
    const double* seg = lut + kLutSegOff + ((size_t)table * 64 + idx) * 2;
    const double y = seg[0] * (x - x1) + seg[1];
    return (float)y;
}

template <int K>
struct Ctx {
    char* lane;           // this lane's LDS base: smem + lane*4*K
    const KernelArgs* a;
    float* itramLane;     // itram + (wave*iSlots*64 + lane)*K
    float* xtramLane;
};

template <int K>
__device__ __forceinline__ void orOod(const Ctx<K>& c, const uint32_t (&bits)[K]) {
    bool any = false;
#pragma unroll
    for (int k = 0; k < K; ++k) any = any || bits[k] != 0;
    if (any) {
        Vec<K> o = ldsLoad<K>(c.lane, c.a->oodOff);
#pragma unroll
        for (int k = 0; k < K; ++k) o.v[k] = asF(asU(o.v[k]) | bits[k]);
        ldsStore<K>(c.lane, c.a->oodOff, o);
    }
}

// reference read/write{Small,Large}Delay, source/FX8010.cpp:909-967; the cursors live in LDS rows
// Opt-in DANE delay-line model (NOT reference behaviour: docs/TRAM Registermapping.pdf, kX/DANE convention; the oracle has
// the same switch): one address counter per TRAM - the write-cursor row - that steps DOWN once per sample period
// (daneStep); a tap, read or write, addresses (counter + position) mod size, position in samples or, with F_TRAM_SHIFT,
// in DANE addresses of 0x800 per sample.  A value written at position pw is read pr - pw samples later at position pr.
template <int K>
__device__ __forceinline__ void daneTap(const Ctx<K>& c, float* __restrict__ base, int slots, int size, uint32_t counterOff, uint32_t flags,
                                        bool isRead, Vec<K>& av, const Vec<K>& yv, const bool (&run)[K]) {
    uint32_t ood[K];
    const Vec<K> cur = ldsLoad<K>(c.lane, counterOff);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        ood[k] = 0;
        if (!run[k]) continue;
        if (size <= 0) { ood[k] = OOD_TRAM_SIZE0; if (isRead) av.v[k] = 0.0f; continue; }
        // whole samples, or (F_TRAM_SHIFT) a DANE address held as the register's fixed-point fraction: value * 2^31, 0x800 per sample
        const int position = (flags & F_TRAM_SHIFT) ? (cvtt_f32(yv.v[k] * 2147483648.0f) >> 11) : cvtt_f32(yv.v[k]);
        long long idx = ((long long)(int)asU(cur.v[k]) + position) % size;
        if (idx < 0) idx += size;
        float* cell = base + (size_t)idx * (64 * K) + k;
        if (idx < slots) { if (isRead) av.v[k] = *cell; else *cell = av.v[k]; }
        else if (isRead) av.v[k] = 0.0f;
        if (isRead && (flags & F_TRAM_INTERP)) {
            // opt-in: x0 + f * (x1 - x0) with the address's low 11 bits (oracle dane_read: four fp32 operations, x1 - x0 as
            // x1 + (-1.0 * x0) so that a NaN keeps its sign, sources ordered like the oracle's operands)
            const int frac = cvtt_f32(yv.v[k] * 2147483648.0f) & 0x7ff;
            if (frac != 0) {
                long long i1 = ((long long)(int)asU(cur.v[k]) + position + 1) % size;
                if (i1 < 0) i1 += size;
                const float x0 = av.v[k], x1 = i1 < slots ? base[(size_t)i1 * (64 * K) + k] : 0.0f;
                const float d = addFirst(x1, mulFirst(-1.0f, x0));
                av.v[k] = addFirst(x0, mulFirst((float)frac * 0.00048828125f, d));
            }
        }
    }
    orOod<K>(c, ood);
}
template <int K>
__device__ __forceinline__ void daneStep(const Ctx<K>& c, int size, uint32_t counterOff) {
    if (size <= 0) return;
    Vec<K> cur = ldsLoad<K>(c.lane, counterOff);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int v = (int)asU(cur.v[k]);
        cur.v[k] = asF((uint32_t)(v <= 0 ? size - 1 : v - 1));
    }
    ldsStore<K>(c.lane, counterOff, cur);
}

template <int K>
__device__ __forceinline__ Vec<K> tramRead(const Ctx<K>& c, float* __restrict__ base, int slots, int size, uint32_t cursorOff,
                                           const Vec<K>& yv, const bool (&run)[K]) {
    Vec<K> out = splat<K>(0.0f);
    uint32_t ood[K];
    Vec<K> cur = ldsLoad<K>(c.lane, cursorOff);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        ood[k] = 0;
        if (!run[k]) continue;
        if (size <= 0) { ood[k] = OOD_TRAM_SIZE0; continue; }
        int rpos = (int)asU(cur.v[k]);
        int position = cvtt_f32(yv.v[k]);
        position = position > size - 1 ? size - 1 : position;
        position = position < 0 ? 0 : position;
        int idx = rpos - position;               // (rpos - p) % size with 0 <= rpos < size
        if (idx < 0) { ood[k] = OOD_TRAM_READ_NEG; idx += size; }
        out.v[k] = (idx < slots) ? base[(size_t)idx * (64 * K) + k] : 0.0f;
        rpos = (rpos + 1 >= size) ? 0 : rpos + 1;
        cur.v[k] = asF((uint32_t)rpos);
    }
    ldsStore<K>(c.lane, cursorOff, cur);
    orOod<K>(c, ood);
    return out;
}
template <int K>
__device__ __forceinline__ void tramWrite(const Ctx<K>& c, float* __restrict__ base, int slots, int size, int refCap, uint32_t cursorOff,
                                          const Vec<K>& av, const Vec<K>& yv, const bool (&run)[K]) {
    uint32_t ood[K];
    Vec<K> cur = ldsLoad<K>(c.lane, cursorOff);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        ood[k] = 0;
        if (!run[k]) continue;
        if (size <= 0) { ood[k] = OOD_TRAM_SIZE0; continue; }
        int wpos = (int)asU(cur.v[k]);
        int position = cvtt_f32(yv.v[k]);
        position = position > size - 1 ? size - 1 : position;
        position = position < 0 ? 0 : position;
        const int idx = wpos + position;         // the reference applies no modulo here
        if (idx >= refCap || idx >= slots) ood[k] = OOD_TRAM_WRITE_OOB;
        else base[(size_t)idx * (64 * K) + k] = av.v[k];
        wpos = (wpos + 1 >= size) ? 0 : wpos + 1;
        cur.v[k] = asF((uint32_t)wpos);
    }
    ldsStore<K>(c.lane, cursorOff, cur);
    orOod<K>(c, ood);
}

// One record.  `run` = which of this lane's K instances execute it (all true outside SKIP shadows).
template <int K, bool MULTIPASS>
__device__ __forceinline__ void execOp(const Ctx<K>& c, uint32_t w0, uint32_t rOff, uint32_t wA, uint32_t wX, uint32_t wY, uint32_t w5,
                                       const bool (&run)[K], bool shadow) {
    constexpr uint32_t ROWB = 256u * K;
    // operand fetch is opcode-independent: LDS row or immediate
    const Vec<K> a = (w0 & F_UA) ? splat<K>(asF(wA)) : ldsLoad<K>(c.lane, wA);
    const Vec<K> x = (w0 & F_UX) ? splat<K>(asF(wX)) : ldsLoad<K>(c.lane, wX);
    const Vec<K> y = (w0 & F_UY) ? splat<K>(asF(wY)) : ldsLoad<K>(c.lane, wY);
    Vec<K> r = a;
    switch (w0 & 0xffu) {
        // NaN operands: the x86 build hands on the NaN of the first operand of each SSE instruction (quieted, sign untouched);
        // gfx950 that of the first source, but flips the sign of a negated source.  first(a, b) keeps a as src0; sums with a
        // subtrahend are A + (-1.0 * p); operand order per opcode as pinned by tests/golden/nan_collisions.json.
        case H_MACS:  // R = sat(A + X*Y)   FX8010.cpp:1077-1085 (MACINTS :1095-1103 is the same expression)
#pragma unroll
            for (int k = 0; k < K; ++k) { const float p = mulFirst(x.v[k], y.v[k]); r.v[k] = saturate1(addFirst(p, a.v[k])); }
            break;
        case H_MACSN:  // R = sat(A - X*Y)   :1086-1094
#pragma unroll
            for (int k = 0; k < K; ++k) { const float p = mulFirst(x.v[k], y.v[k]); r.v[k] = saturate1(addFirst(a.v[k], negated(p))); }
            break;
        case H_ACC3:  // R = sat((A + X) + Y)   :1104-1112
#pragma unroll
            for (int k = 0; k < K; ++k) { const float t = addFirst(a.v[k], x.v[k]); r.v[k] = saturate1(addFirst(t, y.v[k])); }
            break;
        case H_INTERP:  // R = sat((float)((1.0 - X)*A + (double)(X*Y)))   :1180-1187
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float p = mulFirst(x.v[k], y.v[k]);
                const double d = addFirst64(mulFirst64(oneMinus((double)x.v[k]), (double)a.v[k]), (double)p);
                r.v[k] = saturate1((float)d);
            }
            break;
        case H_MACW:  // R = A + wrap(X*Y)   :1126-1131
#pragma unroll
            for (int k = 0; k < K; ++k) r.v[k] = addFirst(a.v[k], wrapAround(mulFirst(x.v[k], y.v[k])));
            break;
        case H_MACWN:  // R = A - wrap(X*Y)   :1132-1137
#pragma unroll
            for (int k = 0; k < K; ++k) r.v[k] = addFirst(a.v[k], negated(wrapAround(mulFirst(x.v[k], y.v[k]))));
            break;
        case H_MACINTW:  // R = wrap(A + X*Y)   :1138-1143
#pragma unroll
            for (int k = 0; k < K; ++k) { const float p = mulFirst(x.v[k], y.v[k]); r.v[k] = wrapAround(addFirst(p, a.v[k])); }
            break;
        case H_MOV:  // MACMV (:1144-1149, accumulator unobservable), input refresh (:1053-1061), output latch (:1229-1233)
            break;
        case H_ANDXOR:  // :1150-1154
#pragma unroll
            for (int k = 0; k < K; ++k) r.v[k] = (float)logicOps(a.v[k], x.v[k], y.v[k]);
            break;
        case H_TSTNEG:  // R = A >= Y ? X : intToFloat(~floatToInt(X))   :1155-1162, :1009-1020
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int xi = cvtt_f32(x.v[k] * 2147483648.0f);
                const float neg = (float)(~xi) * 4.656612873077392578125e-10f;  // exact /2^31
                r.v[k] = a.v[k] >= y.v[k] ? x.v[k] : neg;
            }
            break;
        case H_LIMIT:  // R = A >= Y ? X : Y   :1163-1168
#pragma unroll
            for (int k = 0; k < K; ++k) r.v[k] = a.v[k] >= y.v[k] ? x.v[k] : y.v[k];
            break;
        case H_LIMITN:  // R = A < Y ? X : Y   :1169-1174
#pragma unroll
            for (int k = 0; k < K; ++k) r.v[k] = a.v[k] < y.v[k] ? x.v[k] : y.v[k];
            break;
        case H_LOG:
        case H_EXP: {  // R = (float)linearInterpolate(A, table[(int)X])   :1113-1125; no clamp
            uint32_t ood[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                ood[k] = 0;
                int t;
                if (w0 & F_UX) {
                    t = (int)w5;
                    if (w0 & F_STATIC_OOD) ood[k] = OOD_LUT_TABLE;
                } else {
                    t = cvtt_f32(x.v[k]);
                    if (t < 0 || t > 31) { ood[k] = OOD_LUT_TABLE; t = t < 0 ? 0 : 31; }
                    t += ((w0 & 0xffu) == H_EXP) ? 32 : 0;
                }
                r.v[k] = lutInterpolate(c.a->lut, t, a.v[k], ood[k]);
                if (!run[k]) ood[k] = 0;
            }
            orOod<K>(c, ood);
            break;
        }
        case H_SKIP: {  // if ((float)(int)X == CCR) numSkip = (int)Y   :1175-1179
            const Vec<K> ccr = ldsLoad<K>(c.lane, 0);
            Vec<K> ns = ldsLoad<K>(c.lane, c.a->skipOff);
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (run[k] && (float)cvtt_f32(x.v[k]) == ccr.v[k]) ns.v[k] = asF((uint32_t)cvtt_f32(y.v[k]));
            ldsStore<K>(c.lane, c.a->skipOff, ns);
            break;
        }
        case H_TRAM_IR:  // A = readSmallDelay((int)Y)   :1188-1193
            if (w0 & F_TRAM_DANE) { r = splat<K>(0.0f); daneTap<K>(c, c.itramLane, c.a->iSlots, c.a->iSize, c.a->cursorOff, w0, true, r, y, run); break; }
            r = tramRead<K>(c, c.itramLane, c.a->iSlots, c.a->iSize, c.a->cursorOff + 1u * ROWB, y, run);
            break;
        case H_TRAM_IW:  // writeSmallDelay(A, (int)Y)   :1194-1198
            if (w0 & F_TRAM_DANE) { Vec<K> av = a; daneTap<K>(c, c.itramLane, c.a->iSlots, c.a->iSize, c.a->cursorOff, w0, false, av, y, run); break; }
            tramWrite<K>(c, c.itramLane, c.a->iSlots, c.a->iSize, kMaxITram, c.a->cursorOff, a, y, run);
            break;
        case H_TRAM_XR:  // :1200-1205
            if (w0 & F_TRAM_DANE) { r = splat<K>(0.0f); daneTap<K>(c, c.xtramLane, c.a->xSlots, c.a->xSize, c.a->cursorOff + 2u * ROWB, w0, true, r, y, run); break; }
            r = tramRead<K>(c, c.xtramLane, c.a->xSlots, c.a->xSize, c.a->cursorOff + 3u * ROWB, y, run);
            break;
        case H_TRAM_XW:  // :1206-1210
            if (w0 & F_TRAM_DANE) { Vec<K> av = a; daneTap<K>(c, c.xtramLane, c.a->xSlots, c.a->xSize, c.a->cursorOff + 2u * ROWB, w0, false, av, y, run); break; }
            tramWrite<K>(c, c.xtramLane, c.a->xSlots, c.a->xSize, kMaxXTram, c.a->cursorOff + 2u * ROWB, a, y, run);
            break;
        case H_NOISE: {  // whitenoise()   :993-1000
            Vec<K> g1 = ldsLoad<K>(c.lane, c.a->noiseOff);
            Vec<K> g2 = ldsLoad<K>(c.lane, c.a->noiseOff + ROWB);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t n1 = asU(g1.v[k]) ^ asU(g2.v[k]);
                const float nz = (float)(int)asU(g2.v[k]) * 4.656612873077392578125e-10f;  // g_fScale = 2.0f/0xffffffff = 2^-31
                const uint32_t n2 = asU(g2.v[k]) + n1;
                r.v[k] = nz;
                if (run[k]) { g1.v[k] = asF(n1); g2.v[k] = asF(n2); }
            }
            ldsStore<K>(c.lane, c.a->noiseOff, g1);
            ldsStore<K>(c.lane, c.a->noiseOff + ROWB, g2);
            break;
        }
        case H_END:  // :1212-1215
            if (MULTIPASS) {
                Vec<K> e = ldsLoad<K>(c.lane, c.a->aliveOff + ROWB);
#pragma unroll
                for (int k = 0; k < K; ++k) if (run[k]) e.v[k] = asF(1u);
                ldsStore<K>(c.lane, c.a->aliveOff + ROWB, e);
            }
            break;
        default:  // H_NOP: counted, does nothing (e.g. idelay whose R is neither read nor write)
            break;
    }
    if (w0 & F_WRITE_R) {
        if (shadow) {
            const Vec<K> old = ldsLoad<K>(c.lane, rOff);
#pragma unroll
            for (int k = 0; k < K; ++k) r.v[k] = run[k] ? r.v[k] : old.v[k];
        }
        ldsStore<K>(c.lane, rOff, r);
        if (w0 & F_CCR) {  // setCCR(R), source/FX8010.cpp:211-232
            Vec<K> cc;
#pragma unroll
            for (int k = 0; k < K; ++k) cc.v[k] = ccrOf(r.v[k]);
            if (shadow) {
                const Vec<K> oldc = ldsLoad<K>(c.lane, 0);
#pragma unroll
                for (int k = 0; k < K; ++k) cc.v[k] = run[k] ? cc.v[k] : oldc.v[k];
            }
            ldsStore<K>(c.lane, 0, cc);
        }
    }
}

struct Rec {
    uint32_t w0, w1, w2, w3, w4, w5;
};
__device__ __forceinline__ Rec fetch(ConstU32 prog, int pc) {
    ConstU32 p = prog + (size_t)pc * 8;
    Rec r;
    r.w0 = p[0]; r.w1 = p[1]; r.w2 = p[2]; r.w3 = p[3]; r.w4 = p[4]; r.w5 = p[5];
    return r;
}

template <int K, bool MULTIPASS>
__global__ __launch_bounds__(64) void fx_step_block(const KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr uint32_t ROWB = 256u * K;
    const int lane = threadIdx.x;
    const long long wave = blockIdx.x;
    const long long inst0 = (wave * 64 + lane) * K;  // first of this lane's K instances
    const long long left = a.n - inst0;
    const int nValid = left >= K ? K : (left > 0 ? (int)left : 0);
    const bool vecIo = (nValid == K) && ((a.n % K) == 0);  // aligned, complete run of K instances

    Ctx<K> c;
    c.lane = smem + lane * 4 * K;
    c.a = &a;
    c.itramLane = a.itram ? a.itram + ((size_t)wave * a.iSlots * 64 + lane) * K : nullptr;
    c.xtramLane = a.xtram ? a.xtram + ((size_t)wave * a.xSlots * 64 + lane) * K : nullptr;

    ConstU32 rowTable = (ConstU32)a.rowTable;
    uint32_t* __restrict__ stLane = a.state + inst0;  // rows are nPad apart; inst0+K <= nPad
    const size_t rowStride = (size_t)a.nPad;

    // ---- prologue: state rows -> LDS rows (nPad is a multiple of 256: always full vectors)
    for (int i = 0; i < a.nLoad; ++i) {
        const uint32_t e = rowTable[i];
        ldsStore<K>(c.lane, (e & 0xffffu) * ROWB, gLoad<K>(reinterpret_cast<const float*>(stLane + (size_t)(e >> 16) * rowStride), true, K));
    }
    for (int i = 0; i < a.nZero; ++i) ldsStore<K>(c.lane, rowTable[a.nLoad + a.nStore + i] * ROWB, splat<K>(0.0f));

    const size_t n = (size_t)a.n;
    const int CH = a.channels;
    Vec<K> nextIn[kMaxChannels];
#pragma unroll
    for (int ch = 0; ch < kMaxChannels; ++ch)
        nextIn[ch] = (ch < CH && a.inRow[ch] >= 0 && a.nSamples > 0) ? gLoad<K>(a.in + (size_t)ch * n + inst0, vecIo, nValid) : splat<K>(0.0f);

    bool allRun[K];
#pragma unroll
    for (int k = 0; k < K; ++k) allRun[k] = true;

    for (int s = 0; s < a.nSamples; ++s) {
#pragma unroll
        for (int ch = 0; ch < kMaxChannels; ++ch)
            if (ch < CH && a.inRow[ch] >= 0) ldsStore<K>(c.lane, (uint32_t)a.inRow[ch] * ROWB, nextIn[ch]);
        if (s + 1 < a.nSamples) {
#pragma unroll
            for (int ch = 0; ch < kMaxChannels; ++ch)
                if (ch < CH && a.inRow[ch] >= 0) nextIn[ch] = gLoad<K>(a.in + ((size_t)(s + 1) * CH + ch) * n + inst0, vecIo, nValid);
        }

        ConstU32 prog = (ConstU32)((s == a.nSamples - 1) ? a.last : a.steady);
        if (a.hasShadow) ldsStore<K>(c.lane, a.skipOff, splat<K>(0.0f));  // numSkip is local to process(), FX8010.cpp:1030
        if (MULTIPASS) {
            ldsStore<K>(c.lane, a.aliveOff, splat<K>(asF(1u)));
            ldsStore<K>(c.lane, a.aliveOff + ROWB, splat<K>(0.0f));
        }
        int passes = 0;
        bool again;
        do {
            Rec cur = fetch(prog, 0);
            for (int pc = 0; pc < a.nOps; ++pc) {
                const Rec nxt = fetch(prog, pc + 1 < a.nOps ? pc + 1 : pc);  // prefetch one record ahead
                const uint32_t w0 = cur.w0;
                if (w0 & F_SHADOW) {
                    bool run[K];
                    bool any = false;
                    if (w0 & F_POSTFIX) {
                        const Vec<K> ran = ldsLoad<K>(c.lane, a.skipOff + ROWB);
#pragma unroll
                        for (int k = 0; k < K; ++k) { run[k] = asU(ran.v[k]) != 0; any = any || run[k]; }
                    } else {
                        Vec<K> ns = ldsLoad<K>(c.lane, a.skipOff);
                        Vec<K> alive = splat<K>(0.0f);
                        if (MULTIPASS) alive = ldsLoad<K>(c.lane, a.aliveOff);
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            const int v = (int)asU(ns.v[k]);
                            const bool live = MULTIPASS ? (asU(alive.v[k]) != 0) : true;
                            run[k] = live && v == 0;
                            any = any || run[k];
                            // a skipped instruction only counts the skip down (FX8010.cpp:1235-1241)
                            if (live && !run[k]) ns.v[k] = asF((uint32_t)(v > 0 ? v - 1 : 0));
                        }
                        if (!(w0 & F_PREFIX)) {
                            ldsStore<K>(c.lane, a.skipOff, ns);
                            Vec<K> ran, dyn = ldsLoad<K>(c.lane, a.skipOff + 2 * ROWB);
#pragma unroll
                            for (int k = 0; k < K; ++k) { ran.v[k] = asF(run[k] ? 1u : 0u); dyn.v[k] = asF(asU(dyn.v[k]) + (run[k] ? 1u : 0u)); }
                            ldsStore<K>(c.lane, a.skipOff + ROWB, ran);
                            ldsStore<K>(c.lane, a.skipOff + 2 * ROWB, dyn);
                        }
                    }
                    if (__any(any)) execOp<K, MULTIPASS>(c, w0, cur.w1, cur.w2, cur.w3, cur.w4, cur.w5, run, true);
                } else {
                    execOp<K, MULTIPASS>(c, w0, cur.w1, cur.w2, cur.w3, cur.w4, cur.w5, allRun, false);
                }
                cur = nxt;
            }
            again = false;
            if (MULTIPASS) {
                Vec<K> alive = ldsLoad<K>(c.lane, a.aliveOff);
                const Vec<K> ended = ldsLoad<K>(c.lane, a.aliveOff + ROWB);
                bool anyAlive = false;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool al = asU(alive.v[k]) != 0 && asU(ended.v[k]) == 0;
                    alive.v[k] = asF(al ? 1u : 0u);
                    anyAlive = anyAlive || al;
                }
                ldsStore<K>(c.lane, a.aliveOff, alive);
                ++passes;
                again = __any(anyAlive) && passes < kPassCap;
                if (!again && __any(anyAlive)) {
                    uint32_t ood[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) ood[k] = asU(alive.v[k]) ? OOD_PASS_CAP : 0u;
                    orOod<K>(c, ood);
                }
            }
        } while (again);

#pragma unroll
        for (int ch = 0; ch < kMaxChannels; ++ch)
            if (ch < CH) gStore<K>(a.out + ((size_t)s * CH + ch) * n + inst0, ldsLoad<K>(c.lane, (uint32_t)a.latchRow[ch] * ROWB), vecIo, nValid);
        if (a.tramDane) {  // opt-in delay-line model: the address counters step once per sample period
            daneStep<K>(c, a.iSize, a.cursorOff);
            daneStep<K>(c, a.xSize, a.cursorOff + 2u * ROWB);
        }
    }

    // ---- epilogue: LDS rows -> state rows
    for (int i = 0; i < a.nStore; ++i) {
        const uint32_t e = rowTable[a.nLoad + i];
        gStore<K>(reinterpret_cast<float*>(stLane + (size_t)(e >> 16) * rowStride), ldsLoad<K>(c.lane, (e & 0xffffu) * ROWB), true, K);
    }
    const Vec<K> oodv = ldsLoad<K>(c.lane, a.oodOff);
    Vec<K> dyn = splat<K>(0.0f);
    if (a.hasShadow) dyn = ldsLoad<K>(c.lane, a.skipOff + 2 * ROWB);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        stLane[(size_t)a.oodRow * rowStride + k] |= asU(oodv.v[k]);
        const unsigned long long add = (unsigned long long)a.staticCount * (unsigned long long)a.nSamples + asU(dyn.v[k]);
        unsigned long long cnt = ((unsigned long long)stLane[(size_t)a.countHi * rowStride + k] << 32) | stLane[(size_t)a.countLo * rowStride + k];
        cnt += add;
        stLane[(size_t)a.countLo * rowStride + k] = (uint32_t)cnt;
        stLane[(size_t)a.countHi * rowStride + k] = (uint32_t)(cnt >> 32);
    }
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

template <int K>
hipError_t launchK(const KernelArgs& a, bool multipass, hipStream_t stream) {
    const unsigned grid = (unsigned)((a.n + 64 * K - 1) / (64 * K));
    const size_t ldsBytes = (size_t)a.nRows * 256 * K;
    auto kern = multipass ? fx_step_block<K, true> : fx_step_block<K, false>;
    if (ldsBytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        if (e != hipSuccess) return e;
    }
    // (hipLaunchKernelGGL reports through the thread's "last error": an older one - a failed allocation of this library, or of the
    // host's own code on this thread - must not be taken for this launch's)
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), ldsBytes, stream, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launchStepBlock(const KernelArgs& a, bool multipass, hipStream_t stream) {
    switch (a.instPerLane) {
        case 1: return launchK<1>(a, multipass, stream);
        case 2: return launchK<2>(a, multipass, stream);
        case 4: return launchK<4>(a, multipass, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launchFillRows(uint32_t* state, long long nPad, const uint32_t* d_rows, const uint32_t* d_values, int nRows,
                          hipStream_t stream) {
    if (nRows <= 0) return hipSuccess;
    dim3 grid((unsigned)((nPad + 255) / 256), (unsigned)(nRows < 64 ? nRows : 64));
    (void)hipGetLastError();
    hipLaunchKernelGGL(fx_fill_rows, grid, dim3(256), 0, stream, state, nPad, d_rows, d_values, nRows);
    return hipGetLastError();
}

hipError_t launchReduceRow(const uint32_t* state, long long nPad, long long n, int rowLo, int rowHi, int rowOr,
                           unsigned long long* d_sum, uint32_t* d_or, hipStream_t stream) {
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(fx_reduce_row, dim3((unsigned)blocks), dim3(256), 0, stream, state, nPad, n, rowLo, rowHi, rowOr, d_sum, d_or);
    return hipGetLastError();
}

}  // namespace fx
