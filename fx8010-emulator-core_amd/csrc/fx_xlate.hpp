// fx_xlate.hpp — translation of a lowered FX8010 program into gfx950 machine code.
//
// The interpreter (fx_interp_gfx950.S) spends most of its issue slots on dispatch: record fetch, operand
// index set-up (VGPR index mode) and the jump to the next handler.  A translated program has none of that:
// every emulated instruction becomes the handful of VALU instructions that do its arithmetic, with the
// register-file rows as direct VGPR operands and uniform operands as literals.  Rare opcodes (TRAM, LOG/EXP,
// SKIP, wrap-around and logic ops, anything that writes a live CCR) are not re-implemented: the translated
// code calls the interpreter's own handler for them (operands in the record SGPRs s18..s23, return address
// in s[24:25]), so both paths share one implementation of the reference's semantics.
//
// The code is written over a filler region ("hole") of a template code object — the XLATE flavour of the
// interpreter source: prologue, per-sample frame, one set of handlers, epilogue — in a private copy of the
// ELF image, which is then loaded as a module of its own.  No instruction is ever written to device memory
// by hand and nothing is executed that did not go through the code-object loader.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "fx_asm.hpp"

namespace fx {

// What the translator needs to know about a template build; read from the ELF image (symbols
// <kernel>, <kernel>_table, <kernel>_hole) without a device.
struct XlateTemplate {
    const unsigned char* image = nullptr;
    size_t imageBytes = 0;
    std::string kernelName;
    int vgprs = 0;                        // VGPR budget of the build (register file = v32 .. v[vgprs-1])
    uint32_t handlerOff[kAsmSlots] = {};  // byte offset of each handler (register set _a) from the kernel entry
    uint32_t holeOff = 0;                 // byte offset of the hole from the kernel entry
    uint32_t holeBytes = 0;
    size_t holeFileOff = 0;               // where the hole sits in the ELF file
};

// The template of a VGPR build (ASM_V64 .. ASM_V256); nullptr + err when there is none or the image is malformed.
const XlateTemplate* xlateTemplate(AsmVariant variant, std::string* err);

// What the translator needs to know about the program beyond its records.
struct XlateProgram {
    int iSize = 0, xSize = 0;     // itramsize / xtramsize (the cursors' modulus)
    bool uniformCursors = false;  // all lanes' TRAM cursors move together: kept in SGPRs, TRAM instructions inline
    // LOG/EXP tables in LDS: the fp32 thresholds, x1[] and the {slope, y1} arrays of the tables the program uses
    // (lutTables = their byte offsets in the LUT blob, in LDS order); empty = tables are read from global memory
    std::vector<uint32_t> lutTables;
    bool compactCcr = false;      // set by planXlate for the last-sample streams: live-CCR instructions call the handler
    std::vector<uint8_t> wildRow; // per register-file row: 0 = BOUNDED class (always inside [-1, 1]), 1 = WILD
};
// nRows = rows of the register file, inputRows = the rows the frame writes the PCM input to (-1: unused channel)
XlateProgram xlateProgramOf(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords, int iSize, int xSize,
                            int nRows, const std::vector<int>& inputRows);

struct XlateStats {
    int inlined = 0;     // records translated to straight-line code
    int called = 0;      // records executed by a call to the interpreter's handler
    int instructions = 0;
    int fusedSkips = 0;  // SKIPs translated as a predicate on the value that would have set their CCR
    int regions = 0;     // SKIP shadows run under one EXEC mask (no per-instruction PRED)
    int unitMultipliers = 0;  // multiplications by +-1.0 that were not emitted
    int unsaturated = 0; // saturating instructions whose result provably lies in [-1, 1]: no v_med3 in the fast stream
    bool nonFiniteImmediate = false;  // a NaN / Inf among the uniform operands: no fast stream for this program
};

// Translate one stream of records (encodeAsmStream(ops, nullptr, true): w0 = handler slot) into code that
// starts `codeBase` bytes after the kernel entry.  The code ends with the jump to the end-of-sample frame.
//
// Two flavours.  exactReturns == nullptr: the EXACT stream, whose saturation lets a NaN pass as the reference's
// does (FX8010.cpp:275-279: 3 VALU instructions).  exactReturns != nullptr: the FAST stream, whose saturation is
// one v_med3_f32 - valid while the register file holds only finite values, which the template tracks ("taint":
// state rows, PCM input, TRAM reads and results of non-saturating instructions are checked where they enter).
// After each handler call that can taint, the fast stream tests the taint mask and continues in the exact
// stream at the return address of the same call there (exactReturns[record], produced by the exact translation
// through `returns`).  A wave thus runs fast code until the first non-finite value shows up, exact code after.
// The same hand-over follows the wait for in-flight TRAM reads of programs with uniform cursors (inline TRAM code).
// listing (optional) receives one assembler line per instruction, in llvm-mc syntax.
bool translateStream(const std::vector<MicroOp>& records, const XlateTemplate& tmpl, const XlateProgram& prog, uint32_t codeBase,
                     const std::vector<uint32_t>* exactReturns, std::vector<uint32_t>* code, std::string* listing,
                     XlateStats* stats, std::vector<uint32_t>* returns, std::string* err);

// A loadable code object: the template with the four streams in its hole.
struct XlateImage {
    std::vector<unsigned char> elf;
    // entry offsets from the kernel entry; AsmArgs.steady = steadyFastOff | steadyOff << 32, .last likewise
    uint32_t steadyFastOff = 0, steadyOff = 0, lastFastOff = 0, lastOff = 0;
    uint32_t initOff = 0;   // run-once code (AsmArgs.initOff), 0 = none
    uint32_t ldsBytes = 0;  // dynamic LDS per workgroup
    uint32_t codeBytes = 0;
    XlateStats steady, last;  // of the stream a finite wave runs
    std::vector<uint8_t> wildRow;  // row classes the code relies on: the loader flags BOUNDED rows in the row table
};
// Lays the four streams out ([steady fast][steady exact][last fast][last exact]) and translates them; code[k] /
// listing[k] in that order (listing may be nullptr); code[4] = the run-once code (LDS tables), empty when none.  Without a fast stream (non-finite uniform operand) the
// fast offsets equal the exact ones.
bool planXlate(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords, const XlateTemplate& tmpl,
               const XlateProgram& prog, XlateImage* out, std::vector<uint32_t> code[5], std::string listing[5], std::string* err);
bool buildXlateImage(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords,
                     const XlateTemplate& tmpl, const XlateProgram& prog, XlateImage* out, std::string* err);

}  // namespace fx
