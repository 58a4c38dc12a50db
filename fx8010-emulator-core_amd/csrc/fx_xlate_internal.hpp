// fx_xlate_internal.hpp — what the translation units of the translator share beyond fx_xlate.hpp (internal to csrc/):
//   fx_xlate_elf.cpp     the template code objects: ELF symbols, the hole, placing generated code, the fingerprint
//   fx_xlate.cpp         records -> code: the Translator (one stream), run-once code, hoist planning, row classes, layout of the
//                        four streams (planXlate)
//   fx_xlate_stages.cpp  the stage planner: where a program can be cut, the plan's proof, LDS layout, the staged image
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "fx_xlate.hpp"

namespace fx {
namespace xl {

inline uint32_t align64(uint32_t v) { return (v + 63u) & ~63u; }

// register-file rows a record reads and writes (fx_xlate_stages.cpp; also the translator's dead-result analysis)
struct Access { uint32_t reads[3]; int nReads = 0; int write = -1; bool ccr = false, tram = false, noise = false; };
Access accessOf(const MicroOp& r);

// tap position of the opt-in DANE model from a uniform operand, reduced to 0 .. size-1 (fx_xlate.cpp)
int32_t danePosition(uint32_t bits, bool shifted, int32_t size);

// Run-once code of a program (fx_xlate.cpp): LOG / EXP tables -> LDS, the hoist decision, the priority slice length
void emitInit(const XlateProgram& prog, std::vector<uint32_t>* code, std::string* listing, int sliceBias = -1);

// copy generated code into a private image of the template, `offsetFromEntry` bytes behind the kernel entry (inside the hole)
void placeCode(const XlateTemplate& tmpl, std::vector<unsigned char>* elf, uint32_t offsetFromEntry, const std::vector<uint32_t>& code);

}  // namespace xl
}  // namespace fx
