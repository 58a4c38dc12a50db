// fx_decode.hpp — lowering of a parsed program to the device opcode stream.
//
// The interpreter kernel steps one host-decoded stream for all instances (lanes).  The
// decoder's job is to take everything that is the same for every instance out of the
// per-lane work:
//   * registers no instruction can write (literals, controls, untouched statics) are
//     UNIFORM: their current value is folded into the stream as an immediate;
//   * registers that can differ per instance get a row in the per-wave LDS register file
//     (row r, lane l at byte r*256 + l*4);
//   * INPUT registers are aliased onto the per-sample input row of their channel whenever
//     that is provably identical to the reference's refresh-on-use (FX8010.cpp:1053-1061);
//   * CCR (register 0) is an ordinary row that is only materialised by instructions whose
//     CCR write can be observed (liveness over SKIP shadows);
//   * instructions that can sit in the shadow of a SKIP are flagged, everything else runs
//     without per-lane predication.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "fx_model.hpp"

namespace fx {

// One record of the device stream: 8 dwords, fetched with a single scalar load.
//   w0  handler id [7:0] | flags [31:8]
//   w1  R: LDS byte offset of the destination row (row * 256 * K)
//   w2  A: LDS byte offset of the operand row, or its IEEE bits when F_UA is set
//   w3  X: likewise (F_UX)      w4  Y: likewise (F_UY)
//   w5  extra (LOG/EXP: table index 0..63 when X is uniform)
//   w6, w7 reserved
struct MicroOp {
    uint32_t w[8];
};

enum Handler : uint32_t {
    H_END = 0, H_NOP, H_MACS, H_MACSN, H_ACC3, H_INTERP, H_MACW, H_MACWN, H_MACINTW,
    H_MOV,  // R = A: MACMV, refresh of an INPUT operand, output latch
    H_ANDXOR, H_TSTNEG, H_LIMIT, H_LIMITN, H_LOG, H_EXP, H_SKIP,
    H_TRAM_IR, H_TRAM_IW, H_TRAM_XR, H_TRAM_XW, H_NOISE,
    H_COUNT_
};

enum : uint32_t {
    F_SHADOW = 1u << 8,    // may be inside a SKIP shadow: run under the per-lane skip predicate
    F_CCR = 1u << 9,       // materialise CCR
    F_UA = 1u << 10,       // operand A is the immediate w4
    F_UX = 1u << 11,
    F_UY = 1u << 12,
    F_PREFIX = 1u << 13,   // helper op ahead of its instruction (same predicate, no skip countdown)
    F_POSTFIX = 1u << 14,  // helper op behind its instruction (runs where the instruction ran)
    F_COUNT = 1u << 15,    // a reference instruction (counts towards getInstructionCounter)
    F_STATIC_OOD = 1u << 16,  // decoder already knows this op leaves the parity domain
    F_WRITE_R = 1u << 17,     // the handler's result is stored to row R (and CCR derived from it if F_CCR)
    F_TRAM_DANE = 1u << 18,   // opt-in delay-line model (fx_model.hpp kOptTramDane): slot = (per-sample counter + position) mod size
    F_TRAM_SHIFT = 1u << 19,  // ... positions are DANE addresses: >> 11
    F_TRAM_INTERP = 1u << 20  // ... and a READ tap interpolates between position and position + 1 with the address's low 11 bits
};

// state rows (32-bit words per instance) in the device state block, after the register rows
struct StateLayout {
    int nRegs = 0;      // rows [0, nRegs): register values
    int outBase = 0;    // rows [outBase, outBase+channels): output latches (reference outputBuffer)
    int cursorBase = 0; // +0 iTRAM write, +1 iTRAM read, +2 xTRAM write, +3 xTRAM read
    int noiseBase = 0;  // +0 g_x1, +1 g_x2
    int oodRow = 0;     // sticky out-of-domain flags
    int countLo = 0, countHi = 0;  // executed-instruction counter (64-bit)
    int totalRows = 0;
};

struct RowCopy {
    uint16_t ldsRow;
    uint16_t stateRow;
};

struct Lowered {
    std::vector<MicroOp> steady;  // samples 0 .. S-2
    std::vector<MicroOp> last;    // final sample of a block: every CCR write materialised
    std::vector<RowCopy> loadRows;   // prologue: LDS row <- state row
    std::vector<RowCopy> storeRows;  // epilogue: state row <- LDS row
    std::vector<int> rowOfReg;       // LDS row of a register, -1 if uniform
    std::vector<int> inRow;          // per channel: LDS row of the input sample, -1 if unused
    std::vector<int> latchRow;       // per channel: LDS row of the output latch
    std::vector<int> zeroRows;       // LDS rows cleared at kernel start (bookkeeping rows)
    int skipRow = -1;    // numSkip, ran, dynCount (3 rows) when any instruction is shadowed
    int cursorRow = -1;  // 4 TRAM cursor rows when the program touches TRAM
    int noiseRow = -1;   // 2 LFSR rows when the program draws noise
    int oodRow = -1;     // sticky out-of-domain flags
    int aliveRow = -1;   // alive, ended (2 rows) in multipass programs
    int instPerLane = 1;
    uint32_t rowPitch = 256;  // bytes between rows as encoded in the records
    StateLayout layout;
    int nRows = 0;
    int nLaneRegs = 0, nUniformRegs = 0;
    int staticCount = 0;   // unshadowed reference instructions per sample
    int nShadowed = 0, nCcrLive = 0;
    int tramOpsPerSample = 0;
    bool multipass = false;
    bool tramDane = false;        // the opt-in DANE delay-line model is in force
    bool usesNoise = false, usesITram = false, usesXTram = false, usesLut = false;
    int iSlots = 0, xSlots = 0;  // TRAM slots to allocate per instance
    std::string error;            // non-empty: cannot be lowered
};

// forcedLane[r] != 0 keeps register r per-instance even if no instruction writes it
// (set after fxb_set_register_i gave instances different values).
// instPerLane (K) fixes the LDS row pitch (256*K bytes) the record offsets are expressed in.
// ldsBookkeeping: give skip counter / TRAM cursors / LFSR / flags their own LDS rows (the HIP C++
// kernel); false keeps them out of the register file (the assembly kernel holds them in VGPRs).
// rowPitch: bytes between rows as written into the records (0 = 256*K, the LDS layout; 1 = plain row
// indices, for the assembly kernel that keeps the register file in VGPRs).
Lowered lowerProgram(const Program& prog, const std::vector<float>& hostValue,
                     const std::vector<uint8_t>& forcedLane, int instPerLane, bool ldsBookkeeping = true,
                     uint32_t rowPitch = 0);

StateLayout makeLayout(int nRegs, int channels);

}  // namespace fx
