// fx_asm.cpp — record encoding for, and launching of, the hand-written gfx950 interpreter.
#include "fx_asm.hpp"

#include <cstring>
#include <mutex>
#include <string>

#include "fx_kernel.hpp"
#include "fx_model.hpp"

namespace fx {
namespace {

// the linked code object of fx_interp_gfx950.s, embedded by the Makefile
const unsigned char kInterpBlob[] = {
#include "build/fx_interp_blob.inc"
};

inline uint32_t handlerOf(const MicroOp& m) { return m.w[0] & 0xffu; }
inline bool has(const MicroOp& m, uint32_t f) { return (m.w[0] & f) != 0; }

}  // namespace

bool asmEligible(const Lowered& low, std::string* why) {
    auto no = [&](const char* w) { if (why) *why = w; return false; };
    if (low.instPerLane != 1) return no("more than one instance per lane");
    if (low.rowPitch != 1 && (size_t)low.nRows * 256 > 160 * 1024) return no("register file above the 160 KiB of LDS of a CU (640 rows)");
    if (low.rowPitch == 1 && low.nRows > kAsmVgprRows[ASM_V256]) return no("register file above 224 VGPR rows");
    for (const MicroOp& m : low.steady) {
        const uint32_t h = handlerOf(m);
        if ((h == H_LOG || h == H_EXP) && has(m, F_STATIC_OOD)) return no("LOG/EXP with an out-of-range table");
    }
    return true;
}

namespace {
inline float asFloat(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t asBits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float sat1(float v) { return (v >= 1.0f) ? 1.0f : ((v <= -1.0f) ? -1.0f : v); }  // reference saturate()
// Host-side folding of uniform operands must hand on the same NaN the reference's x86 build would: the FIRST operand's
// (quieted, sign untouched), then the second's - whatever order this file's compiler picks for a plain + or *.
// Operand order per opcode: tests/golden/nan_collisions.json (see fx_xlate.cpp product()).
inline bool isNanF(float f) { return f != f; }
inline float quietF(float f) { uint32_t u = asBits(f) | 0x00400000u; return asFloat(u); }
inline double quietD(double d) { uint64_t u; std::memcpy(&u, &d, 8); u |= 0x0008000000000000ull; std::memcpy(&d, &u, 8); return d; }
inline float addFirst(float a, float b) { return isNanF(a) ? quietF(a) : (isNanF(b) ? quietF(b) : a + b); }
inline float subFirst(float a, float b) { return isNanF(a) ? quietF(a) : (isNanF(b) ? quietF(b) : a - b); }
inline float mulFirst(float a, float b) { return isNanF(a) ? quietF(a) : (isNanF(b) ? quietF(b) : a * b); }
inline double addFirstD(double a, double b) { return a != a ? quietD(a) : (b != b ? quietD(b) : a + b); }
inline double mulFirstD(double a, double b) { return a != a ? quietD(a) : (b != b ? quietD(b) : a * b); }
}  // namespace

std::vector<MicroOp> encodeAsmStream(const std::vector<MicroOp>& ops, const uint64_t* handlers, bool foldUniform) {
    std::vector<MicroOp> out;
    out.reserve(ops.size() + ops.size() / 2 + 8);
    // records cycle through the kernel's four register sets; each set has its own copy of every handler
    auto setAddress = [&](MicroOp& r, uint32_t slot) {
        const uint64_t a = handlers ? handlers[(out.size() % kAsmSets) * kAsmSlots + slot] : (uint64_t)slot;
        r.w[0] = (uint32_t)a;
        r.w[1] = (uint32_t)(a >> 32);
    };
    auto bare = [&](uint32_t slot) {
        MicroOp r{};
        setAddress(r, slot);
        out.push_back(r);
    };
    bool predOpen = false;
    for (size_t i = 0; i < ops.size(); ++i) {
        const MicroOp& m = ops[i];
        // a group = [prefix ops] main [postfix op]; it starts at the first prefix, or at a main op not preceded by one
        const bool groupStart = !has(m, F_POSTFIX) && (i == 0 || !has(ops[i - 1], F_PREFIX));
        if (groupStart) {
            if (has(m, F_SHADOW)) { bare(AS_PRED); predOpen = true; }   // PRED re-opens EXEC itself
            else if (predOpen) { bare(AS_UNPRED); predOpen = false; }
        }
        const uint32_t h = handlerOf(m);
        const uint32_t kind = (has(m, F_UA) ? 1u : 0u) | (has(m, F_UX) ? 2u : 0u) | (has(m, F_UY) ? 4u : 0u);
        const uint32_t ccr = has(m, F_CCR) ? 1u : 0u;
        MicroOp r{};
        r.w[2] = m.w[2];
        r.w[3] = m.w[3];
        r.w[4] = m.w[4];
        r.w[5] = m.w[1];               // destination row
        r.w[6] = kind | (ccr << 3);    // flags of the generic handlers
        if (has(m, F_TRAM_DANE)) r.w[6] |= 16u | (has(m, F_TRAM_SHIFT) ? 32u : 0u) | (has(m, F_TRAM_INTERP) ? 64u : 0u);  // (opt-in DANE taps)
        uint32_t slot = AS_NOP;
        switch (h) {
            case H_MACS:
            case H_MACSN: {
                slot = (h == H_MACS ? AS_MACS : AS_MACSN) + kind * 2 + ccr;
                if (foldUniform && (kind & 6u) == 6u) {  // uniform X and Y: p = X*Y once, here
                    const float prod = mulFirst(asFloat(m.w[3]), asFloat(m.w[4]));
                    r.w[3] = asBits(prod);
                    if (kind == 7u) {
                        const float a = asFloat(m.w[2]);
                        r.w[2] = asBits(sat1(h == H_MACS ? addFirst(prod, a) : subFirst(a, prod)));
                    }
                }
                break;
            }
            case H_ACC3:
                slot = AS_ACC3 + kind * 2 + ccr;
                if (foldUniform && (kind & 3u) == 3u) {  // uniform A and X: t = A + X once, here
                    const float t = addFirst(asFloat(m.w[2]), asFloat(m.w[3]));
                    r.w[2] = asBits(kind == 7u ? sat1(addFirst(t, asFloat(m.w[4]))) : t);
                }
                break;
            case H_INTERP: {
                slot = AS_INTERP + kind * 2 + ccr;
                const float x = asFloat(m.w[3]);
                const double omx = isNanF(x) ? quietD((double)x) : 1.0 - (double)x;
                if (kind & 2u) std::memcpy(&r.w[6], &omx, 8);  // uniform X: (1.0 - (double)X) is the same for every instance
                if (foldUniform && (kind & 6u) == 6u) {
                    const float prod = mulFirst(x, asFloat(m.w[4]));
                    r.w[3] = asBits(prod);
                    if (kind == 7u) {
                        const double d = addFirstD(mulFirstD(omx, (double)asFloat(m.w[2])), (double)prod);
                        r.w[2] = asBits(sat1((float)d));
                    }
                }
                break;
            }
            case H_MACW: slot = AS_MACW; break;
            case H_MACWN: slot = AS_MACWN; break;
            case H_MACINTW: slot = AS_MACINTW; break;
            case H_MOV: slot = AS_MOV; break;
            case H_ANDXOR: slot = AS_ANDXOR; break;
            case H_TSTNEG: slot = AS_TSTNEG; break;
            case H_LIMIT: slot = AS_LIMIT; break;
            case H_LIMITN: slot = AS_LIMITN; break;
            case H_LOG:
            case H_EXP: {
                slot = AS_LUT;
                if (!has(m, F_UX)) {
                    // the table number is a per-instance value: X stays a row, the Y word (unused by the reference, FX8010.cpp:1114)
                    // carries the byte offset of table 0 of the family
                    r.w[4] = (uint32_t)((kLutSegOff + (size_t)(h == H_EXP ? 32 : 0) * 128) * 8);
                    r.w[6] |= 4u;
                    break;
                }
                r.w[3] = (uint32_t)((kLutSegOff + (size_t)m.w[5] * 128) * 8);  // this table's {slope, y1}[64]
                // a uniform operand inside the table: the result is the same constant for every instance and sample - the
                // reference's own arithmetic (linearInterpolate, source/FX8010.cpp:283-296), once, here; R = that constant
                // (outside the table the kernel's code runs: it raises the out-of-domain flag)
                const float x = asFloat(m.w[2]);
                if (foldUniform && has(m, F_UA) && x >= -1.0f && x <= 1.0f && m.w[5] < 64u) {
                    static const Luts luts;
                    const double* tbl = m.w[5] < 32u ? luts.log_[m.w[5]] : luts.exp_[m.w[5] - 32u];
                    const double xd = (double)x, step = (1.0 - -1.0) / (double)(64 - 1);
                    const int idx = (int)((xd - -1.0) / step);
                    const double x1 = -1.0 + idx * step, x2 = -1.0 + (idx + 1) * step, y1 = tbl[idx], y2 = tbl[idx + 1];
                    const double y = (y2 - y1) / (x2 - x1) * (xd - x1) + y1;
                    slot = AS_MOV;
                    r.w[2] = asBits((float)y);
                    r.w[3] = 0;
                    r.w[4] = 0;
                    r.w[6] = 1u | (ccr << 3);
                }
                break;
            }
            case H_SKIP: slot = AS_SKIP; break;
            case H_TRAM_IR: slot = AS_TRAM_IR; break;
            case H_TRAM_IW: slot = AS_TRAM_IW; break;
            case H_TRAM_XR: slot = AS_TRAM_XR; break;
            case H_TRAM_XW: slot = AS_TRAM_XW; break;
            case H_NOISE: slot = AS_NOISE; break;
            case H_END: slot = has(m, F_SHADOW) ? AS_END : AS_NOP; break;  // (END in a shadow: a multi-pass program - the handler notes who has finished)
            default: slot = AS_NOP; break;  // NOP only counts
        }
        setAddress(r, slot);
        out.push_back(r);
    }
    bare(AS_ENDSAMPLE);
    for (int k = 0; k < 4; ++k) bare(AS_NOP);  // pad: the fetch runs up to two windows (four records) ahead
    return out;
}

namespace {
std::mutex g_mu;
hipModule_t g_modules[64] = {};
hipFunction_t g_funcs[64][ASM_VARIANTS] = {};
uint64_t* g_tables[64][ASM_VARIANTS] = {};
const char* const kVariantNames[ASM_VARIANTS] = {"fx_interp_lds", "fx_interp_v64", "fx_interp_v72", "fx_interp_v80", "fx_interp_v96",
                                                 "fx_interp_v128", "fx_interp_v168", "fx_interp_v256"};

hipError_t functionFor(AsmVariant variant, int device, hipFunction_t* fn) {
    if (device < 0 || device >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_modules[device]) {
        hipError_t e = hipModuleLoadData(&g_modules[device], kInterpBlob);
        if (e != hipSuccess) return e;
    }
    if (!g_funcs[device][variant]) {
        hipError_t e = hipModuleGetFunction(&g_funcs[device][variant], g_modules[device], kVariantNames[variant]);
        if (e != hipSuccess) return e;
    }
    *fn = g_funcs[device][variant];
    return hipSuccess;
}

hipError_t launchRaw(hipFunction_t fn, const AsmArgs& args, unsigned grid, size_t ldsBytes, hipStream_t stream, unsigned wavesPerGroup = 1) {
    AsmArgs a = args;
    size_t size = sizeof(a);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(fn, grid, 1, 1, 64 * wavesPerGroup, 1, 1, (unsigned)ldsBytes, stream, nullptr, config);
}
}  // namespace

// One wavefront of the build's probe kernel (<name>_probe, same code object) writes the absolute address of
// each of the build's kAsmSets * kAsmSlots handlers; they stay valid for as long as the module is loaded.
// g_mu guards the tables only: the probe launch, its wait and the copy back run outside the lock (several shards on several
// devices initialise at once; two threads racing for the same table both probe, the first to publish wins).
const uint64_t* asmHandlerTable(AsmVariant variant, int device, hipError_t* err) {
    hipFunction_t fn;
    hipError_t e = functionFor(variant, device, &fn);  // loads the module (under the lock, once per device)
    uint64_t* table = nullptr;
    if (e == hipSuccess) {
        hipModule_t module;
        {
            std::lock_guard<std::mutex> lock(g_mu);
            table = g_tables[device][variant];
            module = g_modules[device];
        }
        if (!table) {
            const std::string probeName = std::string(kVariantNames[variant]) + "_probe";
            e = hipModuleGetFunction(&fn, module, probeName.c_str());
            const size_t bytes = sizeof(uint64_t) * kAsmSets * kAsmSlots;
            uint64_t* dbuf = nullptr;
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dbuf), bytes);
            if (e == hipSuccess) {
                AsmArgs a{};
                a.out = reinterpret_cast<float*>(dbuf);
                a.n = 64;
                a.nPad = 256;
                hipStream_t stream = nullptr;
                e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);  // (a stream of its own: no device-wide wait)
                if (e == hipSuccess) e = launchRaw(fn, a, 1, 0, stream);
                uint64_t* host = new uint64_t[kAsmSets * kAsmSlots];
                if (e == hipSuccess) e = hipMemcpyAsync(host, dbuf, bytes, hipMemcpyDeviceToHost, stream);
                if (e == hipSuccess) e = hipStreamSynchronize(stream);
                if (stream) (void)hipStreamDestroy(stream);
                (void)hipFree(dbuf);
                if (e == hipSuccess) {
                    std::lock_guard<std::mutex> lock(g_mu);
                    if (!g_tables[device][variant]) g_tables[device][variant] = host;
                    else delete[] host;
                    table = g_tables[device][variant];
                } else {
                    delete[] host;
                }
            }
        }
    }
    if (err) *err = e;
    return e == hipSuccess ? table : nullptr;
}

hipError_t launchAsmFunction(hipFunction_t fn, const AsmArgs& args, unsigned grid, size_t ldsBytes, hipStream_t stream, unsigned wavesPerGroup) {
    return launchRaw(fn, args, grid, ldsBytes, stream, wavesPerGroup);
}

hipError_t launchAsmInterp(const AsmArgs& args, AsmVariant variant, size_t ldsBytes, int device, hipStream_t stream) {
    hipFunction_t fn;
    hipError_t e = functionFor(variant, device, &fn);
    if (e != hipSuccess) return e;
    return launchRaw(fn, args, (unsigned)((args.n + 63) / 64), ldsBytes, stream);
}

}  // namespace fx
