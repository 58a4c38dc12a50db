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

// The template of a VGPR build (ASM_V64 .. ASM_V256); nullptr + err when the image is malformed.
const XlateTemplate* xlateTemplate(AsmVariant variant, std::string* err);

struct XlateStats {
    int inlined = 0;     // records translated to straight-line code
    int called = 0;      // records executed by a call to the interpreter's handler
    int instructions = 0;
};

// Translate one stream of records (encodeAsmStream(ops, nullptr, true): w0 = handler slot) into code that
// starts `codeBase` bytes after the kernel entry.  The code ends with the jump to the end-of-sample frame.
// listing (optional) receives one assembler line per instruction, in llvm-mc syntax.
bool translateStream(const std::vector<MicroOp>& records, const XlateTemplate& tmpl, uint32_t codeBase,
                     std::vector<uint32_t>* code, std::string* listing, XlateStats* stats, std::string* err);

// A loadable code object: the template with both streams in its hole.
struct XlateImage {
    std::vector<unsigned char> elf;
    uint32_t steadyOff = 0, lastOff = 0;  // entry offsets from the kernel entry (passed as AsmArgs.steady/.last)
    uint32_t codeBytes = 0;
    XlateStats steady, last;
};
bool buildXlateImage(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords,
                     const XlateTemplate& tmpl, XlateImage* out, std::string* err);

}  // namespace fx
