// fx_xlate_elf.cpp — the template code objects of the translator: the embedded ELF images (one per VGPR build), their symbols
// and code hole, placing generated code into a private copy, and the fingerprint of a finished image (see fx_xlate.hpp).
#include <elf.h>

#include <cstring>
#include <mutex>

#include "fx_xlate_internal.hpp"

namespace fx {
namespace {

// the template code objects (XLATE flavour of fx_interp_gfx950.S, one per VGPR build), embedded by the Makefile
const unsigned char kBlobV64[] = {
#include "build/fx_xlate_v64_blob.inc"
};
const unsigned char kBlobV72[] = {
#include "build/fx_xlate_v72_blob.inc"
};
const unsigned char kBlobV80[] = {
#include "build/fx_xlate_v80_blob.inc"
};
const unsigned char kBlobV96[] = {
#include "build/fx_xlate_v96_blob.inc"
};
const unsigned char kBlobV128[] = {
#include "build/fx_xlate_v128_blob.inc"
};
const unsigned char kBlobV168[] = {
#include "build/fx_xlate_v168_blob.inc"
};
const unsigned char kBlobV256[] = {
#include "build/fx_xlate_v256_blob.inc"
};

struct BlobRef { const unsigned char* p; size_t n; const char* kernel; int vgprs; };
const BlobRef kBlobs[ASM_VARIANTS] = {
    {nullptr, 0, "", 0},
    {kBlobV64, sizeof(kBlobV64), "fx_xlate_v64", 64},
    {kBlobV72, sizeof(kBlobV72), "fx_xlate_v72", 72},
    {kBlobV80, sizeof(kBlobV80), "fx_xlate_v80", 80},
    {kBlobV96, sizeof(kBlobV96), "fx_xlate_v96", 96},
    {kBlobV128, sizeof(kBlobV128), "fx_xlate_v128", 128},
    {kBlobV168, sizeof(kBlobV168), "fx_xlate_v168", 168},
    {kBlobV256, sizeof(kBlobV256), "fx_xlate_v256", 256},
};

// ---- ELF: value and file offset of a named symbol ------------------------------------------------------
struct SymbolAt { uint64_t value = 0; size_t fileOff = 0; bool found = false; };

SymbolAt findSymbol(const unsigned char* img, size_t n, const std::string& name) {
    SymbolAt r;
    if (n < sizeof(Elf64_Ehdr)) return r;
    Elf64_Ehdr eh;
    std::memcpy(&eh, img, sizeof(eh));
    if (std::memcmp(eh.e_ident, ELFMAG, SELFMAG) != 0 || eh.e_ident[EI_CLASS] != ELFCLASS64) return r;
    if (eh.e_shoff == 0 || eh.e_shentsize != sizeof(Elf64_Shdr) || eh.e_shoff + (uint64_t)eh.e_shnum * sizeof(Elf64_Shdr) > n) return r;
    std::vector<Elf64_Shdr> sh(eh.e_shnum);
    std::memcpy(sh.data(), img + eh.e_shoff, sh.size() * sizeof(Elf64_Shdr));
    for (const Elf64_Shdr& s : sh) {
        if (s.sh_type != SHT_SYMTAB && s.sh_type != SHT_DYNSYM) continue;
        if (s.sh_link >= sh.size() || s.sh_entsize != sizeof(Elf64_Sym) || s.sh_offset + s.sh_size > n) continue;
        const Elf64_Shdr& str = sh[s.sh_link];
        if (str.sh_offset + str.sh_size > n) continue;
        const size_t count = s.sh_size / sizeof(Elf64_Sym);
        for (size_t i = 0; i < count; ++i) {
            Elf64_Sym sym;
            std::memcpy(&sym, img + s.sh_offset + i * sizeof(Elf64_Sym), sizeof(sym));
            if (sym.st_name >= str.sh_size) continue;
            const char* nm = reinterpret_cast<const char*>(img + str.sh_offset + sym.st_name);
            const size_t maxLen = str.sh_size - sym.st_name;
            if (strnlen(nm, maxLen) == maxLen || name != nm) continue;
            if (sym.st_shndx == SHN_UNDEF || sym.st_shndx >= sh.size()) continue;
            const Elf64_Shdr& sec = sh[sym.st_shndx];
            if (sym.st_value < sec.sh_addr || sym.st_value > sec.sh_addr + sec.sh_size) continue;
            r.value = sym.st_value;
            r.fileOff = (size_t)(sec.sh_offset + (sym.st_value - sec.sh_addr));
            r.found = true;
            return r;
        }
    }
    return r;
}

std::mutex g_mu;
XlateTemplate g_templates[ASM_VARIANTS];
bool g_parsed[ASM_VARIANTS] = {};
std::string g_parseErr[ASM_VARIANTS];

}  // namespace

const XlateTemplate* xlateTemplate(AsmVariant variant, std::string* err) {
    if (variant <= ASM_LDS || variant >= ASM_VARIANTS) {
        if (err) *err = "no translation template for this build";
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_parsed[variant]) {
        g_parsed[variant] = true;
        const BlobRef& b = kBlobs[variant];
        XlateTemplate t;
        t.image = b.p;
        t.imageBytes = b.n;
        t.kernelName = b.kernel;
        t.vgprs = b.vgprs;
        const SymbolAt kn = findSymbol(b.p, b.n, t.kernelName);
        const SymbolAt tab = findSymbol(b.p, b.n, t.kernelName + "_table");
        const SymbolAt hole = findSymbol(b.p, b.n, t.kernelName + "_hole");
        if (!kn.found || !tab.found || !hole.found || tab.fileOff + (kAsmSlots + 2) * 4 > b.n) {
            g_parseErr[variant] = "translation template " + t.kernelName + ": symbols not found in the code object";
        } else {
            uint32_t table[kAsmSlots + 2];
            std::memcpy(table, b.p + tab.fileOff, sizeof(table));
            std::memcpy(t.handlerOff, table, sizeof(t.handlerOff));
            t.holeOff = table[kAsmSlots];
            t.holeBytes = table[kAsmSlots + 1];
            t.holeFileOff = hole.fileOff;
            if ((uint64_t)t.holeOff != hole.value - kn.value || t.holeFileOff + t.holeBytes > b.n || (t.holeBytes & 3u))
                g_parseErr[variant] = "translation template " + t.kernelName + ": inconsistent hole";
            else
                g_templates[variant] = t;
        }
    }
    if (!g_parseErr[variant].empty()) {
        if (err) *err = g_parseErr[variant];
        return nullptr;
    }
    return &g_templates[variant];
}

namespace xl {
void placeCode(const XlateTemplate& tmpl, std::vector<unsigned char>* elf, uint32_t offsetFromEntry, const std::vector<uint32_t>& code) {
    if (!code.empty()) std::memcpy(elf->data() + tmpl.holeFileOff + (offsetFromEntry - tmpl.holeOff), code.data(), code.size() * 4);
}
}  // namespace xl

uint64_t imageHash(const XlateImage& image) {
    uint64_t h = 0xcbf29ce484222325ull;   // FNV-1a over 8-byte words
    const size_t n = image.elf.size() / 8;
    for (size_t k = 0; k < n; ++k) {
        uint64_t w;
        std::memcpy(&w, image.elf.data() + 8 * k, 8);
        h = (h ^ w) * 0x100000001b3ull;
    }
    for (size_t k = 8 * n; k < image.elf.size(); ++k) h = (h ^ image.elf[k]) * 0x100000001b3ull;
    h = (h ^ (uint64_t)image.stages ^ ((uint64_t)image.ldsBytes << 8)) * 0x100000001b3ull;
    return h & 0x7fffffffffffffffull;
}

bool buildXlateImage(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords,
                     const XlateTemplate& tmpl, const XlateProgram& prog, XlateImage* out, std::string* err) {
    std::vector<uint32_t> code[5];
    if (!planXlate(steadyRecords, lastRecords, tmpl, prog, out, code, nullptr, err)) return false;
    out->elf.assign(tmpl.image, tmpl.image + tmpl.imageBytes);
    const uint32_t offs[5] = {out->base[0], out->base[1], out->base[2], out->base[3], out->initOff};
    for (int k = 0; k < 5; ++k) xl::placeCode(tmpl, &out->elf, offs[k], code[k]);
    return true;
}

}  // namespace fx
