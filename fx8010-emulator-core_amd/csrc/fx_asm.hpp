// fx_asm.hpp — host side of the hand-written gfx950 interpreter (fx_interp_gfx950.s):
// record encoding, code-object loading and launch.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "fx_decode.hpp"

namespace fx {

// kernarg block of fx_interp_k1 — offsets are the KA_* constants in fx_interp_gfx950.s
struct AsmArgs {
    const uint32_t* steady;
    const uint32_t* last;
    const uint32_t* rowTable;
    uint32_t* state;
    const float* in;
    float* out;
    float* itram;
    float* xtram;
    const double* lut;
    long long n;
    long long nPad;
    int nLoad, nStore;
    int nSamples, channels;
    int inOff[4];
    int latchOff[4];
    int iSlots, xSlots, iSize, xSize;
    int cursorRow, noiseRow;
    int oodRow, countLo, countHi, staticCount;
    int lutX1Off;
    int initOff;  // translated programs: byte offset (from the kernel entry) of code to run once before the first sample, 0 = none
    const uint32_t* tracks;  // translated programs with control tracks: the block's TrackEvent list + values (fx_xlate.hpp), else nullptr
    const uint32_t* stages;  // translated programs cut into stages: StageDescriptor[nStages] (fx_xlate.hpp), else nullptr
    int nStages;             // wavefronts per workgroup (0 / 1: the whole program in one)
    int tramDane;            // interpreter builds, bit 0: the opt-in DANE delay-line model is in force (the address counters step once per sample
                             // period); bit 1: multi-pass program (END lies in a SKIP shadow: the kernel runs passes until every lane has executed it);
                             // bit 2: the wavefronts of a SIMD take turns at the top priority, 2^(bits 12:8) ticks of 10 ns each
};
static_assert(offsetof(AsmArgs, lut) == 0x40, "AsmArgs layout");
static_assert(offsetof(AsmArgs, nLoad) == 0x58, "AsmArgs layout");
static_assert(offsetof(AsmArgs, nSamples) == 0x60, "AsmArgs layout");
static_assert(offsetof(AsmArgs, inOff) == 0x68, "AsmArgs layout");
static_assert(offsetof(AsmArgs, latchOff) == 0x78, "AsmArgs layout");
static_assert(offsetof(AsmArgs, iSlots) == 0x88, "AsmArgs layout");
static_assert(offsetof(AsmArgs, cursorRow) == 0x98, "AsmArgs layout");
static_assert(offsetof(AsmArgs, oodRow) == 0xa0, "AsmArgs layout");
static_assert(offsetof(AsmArgs, lutX1Off) == 0xb0, "AsmArgs layout");
static_assert(offsetof(AsmArgs, tracks) == 0xb8, "AsmArgs layout");
static_assert(offsetof(AsmArgs, stages) == 0xc0, "AsmArgs layout");
static_assert(offsetof(AsmArgs, nStages) == 0xc8, "AsmArgs layout");
static_assert(offsetof(AsmArgs, tramDane) == 0xcc, "AsmArgs layout");
static_assert(sizeof(AsmArgs) == 0xd0, "AsmArgs layout");

// handler slots of fx_interp_gfx950.S (fx_interp_table.inc)
enum AsmSlot : uint32_t {
    AS_ENDSAMPLE = 0, AS_NOP = 1, AS_PRED = 2, AS_UNPRED = 3, AS_MOV = 4, AS_MACW = 5, AS_MACWN = 6, AS_MACINTW = 7,
    AS_ANDXOR = 8, AS_TSTNEG = 9, AS_LIMIT = 10, AS_LIMITN = 11, AS_LUT = 12, AS_SKIP = 13, AS_TRAM_IR = 14,
    AS_TRAM_IW = 15, AS_TRAM_XR = 16, AS_TRAM_XW = 17, AS_NOISE = 18, AS_END = 19 /* END inside a SKIP shadow (multi-pass programs) */,
    AS_MACS = 20, AS_MACSN = 36, AS_ACC3 = 52, AS_INTERP = 68  // + kind*2 + ccr, kind = UA | UX<<1 | UY<<2
};

constexpr int kAsmSlots = 84;  // slots per branch table; the second register set's table follows the first

// the builds of fx_interp_gfx950.S
enum AsmVariant { ASM_LDS = 0, ASM_V64, ASM_V72, ASM_V80, ASM_V96, ASM_V128, ASM_V168, ASM_V256, ASM_VARIANTS };
// register-file rows of the VGPR builds (VGPRs - 32) and the wavefronts per SIMD their VGPR count allows
constexpr int kAsmVgprRows[ASM_VARIANTS] = {0, 32, 40, 48, 64, 96, 136, 224};
constexpr int kAsmWavesPerSimd[ASM_VARIANTS] = {0, 8, 7, 6, 5, 4, 3, 2};

// Can this lowering (K = 1, bookkeeping in VGPRs) run on the assembly kernel?  `why` says why not.
bool asmEligible(const Lowered& low, std::string* why);

constexpr int kAsmSets = 4;            // record register sets of the kernel (records cycle through them)

// Absolute addresses of a build's handlers on a device: [set][slot], obtained once by a probe launch.
const uint64_t* asmHandlerTable(AsmVariant variant, int device, hipError_t* err);

// Translate one lowered stream into the assembly kernel's records: w0:w1 = handler address, w2..w4 = A/X/Y,
// w5 = R, w6 = flags or w6:w7 = (1-X) for INTERP (SKIP shadows become PRED/UNPRED brackets, the stream ends
// with ENDSAMPLE plus pad records for the fetch-ahead).
// foldUniform (VGPR builds): products / sums / whole expressions of uniform operands are evaluated here, with
// the same IEEE operations, and the handlers of those operand kinds expect the folded value (see the .S).
std::vector<MicroOp> encodeAsmStream(const std::vector<MicroOp>& ops, const uint64_t* handlers, bool foldUniform);

// Loads the embedded code object on the device (once per device) and launches the chosen build
// with ceil(n/64) single-wavefront workgroups and ldsBytes of dynamic LDS (0 for the VGPR builds).
hipError_t launchAsmInterp(const AsmArgs& args, AsmVariant variant, size_t ldsBytes, int device, hipStream_t stream);

// Launches a kernel with the interpreter's argument block from another module (a translated program, fx_xlate.hpp):
// `grid` single-wavefront workgroups, ldsBytes of dynamic LDS each.
hipError_t launchAsmFunction(hipFunction_t fn, const AsmArgs& args, unsigned grid, size_t ldsBytes, hipStream_t stream, unsigned wavesPerGroup = 1);

}  // namespace fx
