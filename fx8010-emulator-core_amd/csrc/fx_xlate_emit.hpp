// fx_xlate_emit.hpp — the gfx950 instruction encoder of the translator (internal to csrc/: fx_xlate.cpp, fx_xlate_stages.cpp).
//
// Operand values (Src: VGPR / SGPR / inline constant / literal), the Emitter - one method per instruction FORMAT (VOP1 / VOP2 /
// VOP3 / VOPC / SOP1 / SOP2 / SOPC / SOPK / SOPP / SMEM / DS / GLOBAL), which appends the machine words, keeps the listing line
// that llvm-mc re-assembles to the same bytes (tests/test_xlate.py) and tallies the vector instructions with their issue cost -
// the opcode numbers used, and the register conventions shared with the template (fx_interp_gfx950.S).  Nothing here knows about
// FX8010 programs.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "fx_asm.hpp"

namespace fx {
namespace xl {

// ---- operands --------------------------------------------------------------------------------------------
// operand texts are only needed for a listing; building them costs more than the encoding itself
inline thread_local bool tlsWantText = false;

struct Src {
    uint32_t code = 0;  // 9-bit source operand
    uint32_t lit = 0;
    bool hasLit = false;
    std::string text;
};

inline Src vreg(int n) {
    Src s;
    s.code = 256u + (uint32_t)n;
    if (tlsWantText) s.text = "v" + std::to_string(n);
    return s;
}
inline Src vreg64(int n) {
    Src s;
    s.code = 256u + (uint32_t)n;
    if (tlsWantText) s.text = "v[" + std::to_string(n) + ":" + std::to_string(n + 1) + "]";
    return s;
}
inline Src sreg(int n) {
    Src s;
    s.code = (uint32_t)n;
    if (tlsWantText) s.text = "s" + std::to_string(n);
    return s;
}
inline Src sreg64(int n) {
    Src s;
    s.code = (uint32_t)n;
    if (tlsWantText) s.text = "s[" + std::to_string(n) + ":" + std::to_string(n + 1) + "]";
    return s;
}
inline Src named(uint32_t code, const char* text) {
    Src s;
    s.code = code;
    if (tlsWantText) s.text = text;
    return s;
}

struct InlineF { uint32_t bits; uint32_t code; const char* text; };
inline const InlineF kInlineF32[] = {
    {0x3f000000u, 240, "0.5"}, {0xbf000000u, 241, "-0.5"}, {0x3f800000u, 242, "1.0"},  {0xbf800000u, 243, "-1.0"},      {0x40000000u, 244, "2.0"},
    {0xc0000000u, 245, "-2.0"}, {0x40800000u, 246, "4.0"},  {0xc0800000u, 247, "-4.0"}, {0x3e22f983u, 248, "0.15915494"},
};
struct InlineD { uint64_t bits; uint32_t code; const char* text; };
inline const InlineD kInlineF64[] = {
    {0x0000000000000000ull, 128, "0"},   {0x3fe0000000000000ull, 240, "0.5"}, {0xbfe0000000000000ull, 241, "-0.5"},
    {0x3ff0000000000000ull, 242, "1.0"}, {0xbff0000000000000ull, 243, "-1.0"}, {0x4000000000000000ull, 244, "2.0"},
    {0xc000000000000000ull, 245, "-2.0"}, {0x4010000000000000ull, 246, "4.0"},  {0xc010000000000000ull, 247, "-4.0"},
};

// a 32-bit value as a source operand: inline constant when the bit pattern has one, else a literal
inline Src imm32(uint32_t bits, bool forceLiteral = false) {
    Src s;
    const int32_t iv = (int32_t)bits;
    if (!forceLiteral) {
        if (iv >= 0 && iv <= 64) { s.code = 128u + (uint32_t)iv; if (tlsWantText) s.text = std::to_string(iv); return s; }
        if (iv >= -16 && iv <= -1) { s.code = 192u + (uint32_t)(-iv); if (tlsWantText) s.text = std::to_string(iv); return s; }
        for (const InlineF& k : kInlineF32)
            if (k.bits == bits) { s.code = k.code; if (tlsWantText) s.text = k.text; return s; }
    }
    s.code = 255;
    s.lit = bits;
    s.hasLit = true;
    if (tlsWantText) {
        char buf[16];
        std::snprintf(buf, sizeof(buf), "0x%x", bits);
        s.text = buf;
    }
    return s;
}

// ---- instruction emitter -----------------------------------------------------------------------------------
class Emitter {
    static constexpr uint32_t DS_READ2_B32_OP = 0x37, VOP2_ADDC_OP = 0x1c;

  public:
    Emitter(std::vector<uint32_t>* words, std::string* listing) : w_(*words), text_(listing) { tlsWantText = listing != nullptr; }

    size_t bytes() const { return w_.size() * 4; }
    int count() const { return count_; }
    // vector-ALU instructions on the path a finite, in-domain wave takes every sample (cold(true) brackets code that
    // such a wave does not execute: entry stubs, out-of-range paths, the second LUT trip)
    int valu() const { return valu_; }
    int valuSlow() const { return valuSlow_; }
    int valuClocks() const { return (int)((valuClocksX100_ + 50) / 100); }
    void cold(bool on) { coldDepth_ += on ? 1 : -1; cold_ = coldDepth_ > 0; }
    // uniform constants kept in VGPRs for the whole launch (XlateProgram::vconst)
    void constants(const std::vector<std::pair<uint32_t, int>>* pool) { pool_ = pool; }
    int pooled(uint32_t bits) const {
        if (pool_)
            for (const auto& c : *pool_)
                if (c.first == bits) return c.second;
        return -1;
    }
    // the listing is kept as lines until the stream is complete (forward branches are patched in place)
    void finish() {
        if (!text_) return;
        for (const std::string& l : lines_) { *text_ += l; *text_ += '\n'; }
        lines_.clear();
    }

    // a SOPP branch whose target is not known yet; bind() fixes it to the then-current position
    struct Fixup { size_t word = 0, lineNo = 0; std::string name; };
    Fixup branchForward(uint32_t op, const char* name) {
        Fixup f;
        f.word = w_.size();
        f.lineNo = lines_.size();
        f.name = name;
        w_.push_back(0xbf800000u | (op << 16));
        ++count_;
        if (text_) lines_.push_back(std::string(name) + " 0");
        return f;
    }
    void bind(const Fixup& f) {
        const size_t delta = w_.size() - (f.word + 1);
        w_[f.word] = (w_[f.word] & 0xffff0000u) | (uint32_t)(delta & 0xffffu);
        if (text_) lines_[f.lineNo] = f.name + " " + std::to_string(delta);
    }
    // branch to a position already emitted (word index inside this stream)
    bool branchBack(uint32_t op, const char* name, size_t targetWord) {
        const int64_t delta = (int64_t)targetWord - ((int64_t)w_.size() + 1);
        if (delta < -32768) return false;
        sopp(op, name, (uint32_t)delta & 0xffffu, true);
        return true;
    }
    size_t words() const { return w_.size(); }
    // a two-dword instruction given as its words and its listing line (SMEM loads)
    void raw2(uint32_t w0, uint32_t w1, const std::string& text) {
        w_.push_back(w0);
        w_.push_back(w1);
        ++count_;
        if (text_) line(text);
    }
    // s_waitcnt vmcnt(n), the other counters left alone (n <= 63: bits 3:0 and 15:14)
    void waitVmcnt(int n) {
        if (n > 63) n = 63;
        if (n < 0) n = 0;
        w_.push_back(0xbf8c0000u | 0x0f70u | (uint32_t)(n & 15) | ((uint32_t)(n >> 4) << 14));
        ++count_;
        if (text_) line("s_waitcnt vmcnt(" + std::to_string(n) + ")");
    }

    void vop2(uint32_t op, const char* name, int vdst, const Src& src0, int vsrc1, const char* tail = "") {
        tally(name);
        put((op << 25) | ((uint32_t)vdst << 17) | ((uint32_t)vsrc1 << 9) | src0.code, src0);
        if (text_) line(std::string(name) + " v" + std::to_string(vdst) + ", " + src0.text + ", v" + std::to_string(vsrc1) + tail);
    }
    void vop1(uint32_t op, const char* name, const Src& vdst, const Src& src0) {
        tally(name);
        put(0x7e000000u | ((vdst.code & 0xffu) << 17) | (op << 9) | src0.code, src0);
        if (text_) line(std::string(name) + " " + vdst.text + ", " + src0.text);
    }
    void vopc(uint32_t op, const char* name, const Src& src0, int vsrc1) {
        tally(name);
        put(0x7c000000u | (op << 17) | ((uint32_t)vsrc1 << 9) | src0.code, src0);
        if (text_) line(std::string(name) + " vcc, " + src0.text + ", v" + std::to_string(vsrc1));
    }
    // VOP3A: no literals on gfx9; neg = per-source negate bits
    void vop3(uint32_t op, const char* name, const Src& vdst, const Src& s0, const Src& s1, const Src* s2, uint32_t neg = 0) {
        tally(name);
        w_.push_back(0xd0000000u | (op << 16) | (vdst.code & 0xffu));
        w_.push_back(s0.code | (s1.code << 9) | ((s2 ? s2->code : 0u) << 18) | (neg << 29));
        ++count_;
        if (!text_) return;
        std::string t = std::string(name) + " " + vdst.text + ", " + ((neg & 1) ? "-" : "") + s0.text + ", " + ((neg & 2) ? "-" : "") + s1.text;
        if (s2) t += std::string(", ") + ((neg & 4) ? "-" : "") + s2->text;
        line(t);
    }
    // VOPC in its VOP3 form, result to VCC, |src0| when abs0
    void vop3cmp(uint32_t op, const char* name, const Src& s0, bool abs0, const Src& s1) {
        tally(name);
        w_.push_back(0xd0000000u | (op << 16) | (abs0 ? 0x100u : 0u) | 106u);
        w_.push_back(s0.code | (s1.code << 9));
        ++count_;
        if (text_) line(std::string(name) + " vcc, " + (abs0 ? "|" + s0.text + "|" : s0.text) + ", " + s1.text);
    }
    void sop1(uint32_t op, const char* name, const Src& sdst, const Src& ssrc) {
        put(0xbe800000u | ((sdst.code & 0x7fu) << 16) | (op << 8) | ssrc.code, ssrc);
        if (text_) line(std::string(name) + " " + sdst.text + ", " + ssrc.text);
    }
    void sop1NoDst(uint32_t op, const char* name, const Src& ssrc) {
        put(0xbe800000u | (op << 8) | ssrc.code, ssrc);
        if (text_) line(std::string(name) + " " + ssrc.text);
    }
    void sop2(uint32_t op, const char* name, const Src& sdst, const Src& s0, const Src& s1) {
        // at most one literal, which then follows the instruction word
        const Src& l = s1.hasLit ? s1 : s0;
        put(0x80000000u | (op << 23) | ((sdst.code & 0x7fu) << 16) | (s1.code << 8) | s0.code, l);
        if (text_) line(std::string(name) + " " + sdst.text + ", " + s0.text + ", " + s1.text);
    }
    // global_{load,store}_dword with an SGPR base pair and a VGPR byte offset, no immediate offset
    void global(uint32_t op, bool load, int vdata, int vaddr, int sbase, bool nt = false) {
        w_.push_back(0xdc008000u | (op << 18) | (nt ? 1u << 17 : 0u));  // (offset field 0)
        w_.push_back((uint32_t)vaddr | (load ? 0u : (uint32_t)vdata << 8) | ((uint32_t)sbase << 16) | (load ? (uint32_t)vdata << 24 : 0u));
        ++count_;
        const std::string base = "s[" + std::to_string(sbase) + ":" + std::to_string(sbase + 1) + "]";
        if (!text_) return;
        if (load) line("global_load_dword v" + std::to_string(vdata) + ", v" + std::to_string(vaddr) + ", " + base + (nt ? " nt" : ""));
        else line("global_store_dword v" + std::to_string(vaddr) + ", v" + std::to_string(vdata) + ", " + base + (nt ? " nt" : ""));
    }
    void waitVmcnt0() { waitVmcnt(0); }
    // global_load_dwordx2 / x4 into v[vdata ..], VGPR byte offset, SGPR base pair
    void globalLoadWide(uint32_t op, int dwords, int vdata, int vaddr, int sbase) {
        w_.push_back(0xdc008000u | (op << 18));
        w_.push_back((uint32_t)vaddr | ((uint32_t)sbase << 16) | ((uint32_t)vdata << 24));
        ++count_;
        if (text_) line("global_load_dwordx" + std::to_string(dwords) + " v[" + std::to_string(vdata) + ":" + std::to_string(vdata + dwords - 1) + "], v" +
             std::to_string(vaddr) + ", s[" + std::to_string(sbase) + ":" + std::to_string(sbase + 1) + "]");
    }
    // LDS: reads return into v[vdst..], byte offset in the instruction (read2: two dword offsets)
    void dsRead(uint32_t op, const char* name, int dwords, int vdst, int vaddr, uint32_t offset) {
        w_.push_back(0xd8000000u | (op << 17) | (offset & 0xffffu));
        w_.push_back((uint32_t)vaddr | ((uint32_t)vdst << 24));
        ++count_;
        if (text_) line(std::string(name) + " v[" + std::to_string(vdst) + ":" + std::to_string(vdst + dwords - 1) + "], v" + std::to_string(vaddr) +
             (offset ? " offset:" + std::to_string(offset) : ""));
    }
    void dsRead2B32(int vdst, int vaddr, uint32_t dword0, uint32_t dword1) {
        w_.push_back(0xd8000000u | (DS_READ2_B32_OP << 17) | (dword1 << 8) | dword0);
        w_.push_back((uint32_t)vaddr | ((uint32_t)vdst << 24));
        ++count_;
        if (text_) line("ds_read2_b32 v[" + std::to_string(vdst) + ":" + std::to_string(vdst + 1) + "], v" + std::to_string(vaddr) +
             (dword0 ? " offset0:" + std::to_string(dword0) : "") + " offset1:" + std::to_string(dword1));
    }
    void dsWrite2B32(int vaddr, int vdata0, int vdata1, uint32_t dword0, uint32_t dword1) {
        w_.push_back(0xd8000000u | (0x0eu << 17) | (dword1 << 8) | dword0);
        w_.push_back((uint32_t)vaddr | ((uint32_t)vdata0 << 8) | ((uint32_t)vdata1 << 16));
        ++count_;
        if (text_) line("ds_write2_b32 v" + std::to_string(vaddr) + ", v" + std::to_string(vdata0) + ", v" + std::to_string(vdata1) +
             (dword0 ? " offset0:" + std::to_string(dword0) : "") + " offset1:" + std::to_string(dword1));
    }
    void dsReadB32(int vdst, int vaddr, uint32_t offset) {
        w_.push_back(0xd8000000u | (0x36u << 17) | (offset & 0xffffu));
        w_.push_back((uint32_t)vaddr | ((uint32_t)vdst << 24));
        ++count_;
        if (text_) line("ds_read_b32 v" + std::to_string(vdst) + ", v" + std::to_string(vaddr) + (offset ? " offset:" + std::to_string(offset) : ""));
    }
    void dsWriteB32(int vaddr, int vdata, uint32_t offset) {
        w_.push_back(0xd8000000u | (0x0du << 17) | (offset & 0xffffu));
        w_.push_back((uint32_t)vaddr | ((uint32_t)vdata << 8));
        ++count_;
        if (text_) line("ds_write_b32 v" + std::to_string(vaddr) + ", v" + std::to_string(vdata) + (offset ? " offset:" + std::to_string(offset) : ""));
    }
    void barrier() { sopp(0x0au, "s_barrier", 0, false); }
    void dsWriteB128(int vaddr, int vdata, uint32_t offset) {
        w_.push_back(0xd8000000u | (0xdfu << 17) | (offset & 0xffffu));
        w_.push_back((uint32_t)vaddr | ((uint32_t)vdata << 8));
        ++count_;
        if (text_) line("ds_write_b128 v" + std::to_string(vaddr) + ", v[" + std::to_string(vdata) + ":" + std::to_string(vdata + 3) + "]" +
             (offset ? " offset:" + std::to_string(offset) : ""));
    }
    void dsWriteB64(int vaddr, int vdata, uint32_t offset) {
        w_.push_back(0xd8000000u | (0x4du << 17) | (offset & 0xffffu));
        w_.push_back((uint32_t)vaddr | ((uint32_t)vdata << 8));
        ++count_;
        if (text_) line("ds_write_b64 v" + std::to_string(vaddr) + ", v[" + std::to_string(vdata) + ":" + std::to_string(vdata + 1) + "]" +
                        (offset ? " offset:" + std::to_string(offset) : ""));
    }
    // v = v + carry (VCC in and out)
    void addCarry(int v) {
        tally("v_addc_co_u32");
        w_.push_back((VOP2_ADDC_OP << 25) | ((uint32_t)v << 17) | ((uint32_t)v << 9) | 128u);
        ++count_;
        if (text_) line("v_addc_co_u32_e32 v" + std::to_string(v) + ", vcc, 0, v" + std::to_string(v) + ", vcc");
    }
    // v = v - borrow, borrow in from the SGPR pair `sin`, borrow out to the pair `sout`
    void subBorrow(int v, int sin, int sout) {
        tally("v_subbrev_co_u32");
        w_.push_back(0xd0000000u | (0x11eu << 16) | ((uint32_t)sout << 8) | (uint32_t)v);
        w_.push_back(128u | ((256u + (uint32_t)v) << 9) | ((uint32_t)sin << 18));
        ++count_;
        if (text_) line("v_subbrev_co_u32_e64 v" + std::to_string(v) + ", s[" + std::to_string(sout) + ":" + std::to_string(sout + 1) + "], 0, v" +
                        std::to_string(v) + ", s[" + std::to_string(sin) + ":" + std::to_string(sin + 1) + "]");
    }
    // VGPR index mode on: M0 = s<n>, mode 1 = src0 relative (SOPC encoding, the mode nibble in the src1 field)
    void setGprIdxOn(int sreg, uint32_t mode) {
        w_.push_back(0xbf000000u | (0x11u << 16) | (mode << 8) | (uint32_t)sreg);
        ++count_;
        if (text_) line("s_set_gpr_idx_on s" + std::to_string(sreg) + ", gpr_idx(" + (mode == 1u ? "SRC0" : mode == 2u ? "SRC1" : "DST") + ")");
    }
    // s_memrealtime s[sdata:sdata+1] (the 100 MHz clock; diagnostics)
    void memRealTime(int sdata) {
        w_.push_back(0xc0000000u | (0x25u << 18) | ((uint32_t)sdata << 6));
        w_.push_back(0u);
        ++count_;
        if (text_) line("s_memrealtime s[" + std::to_string(sdata) + ":" + std::to_string(sdata + 1) + "]");
    }
    void waitLgkm0() { waitLgkm(0); }
    // s_waitcnt lgkmcnt(n), the other counters left alone (n <= 15: bits 11:8)
    void waitLgkm(int n) {
        if (n > 15) n = 15;
        w_.push_back(0xbf8cc07fu | ((uint32_t)n << 8));
        ++count_;
        if (text_) line("s_waitcnt lgkmcnt(" + std::to_string(n) + ")");
    }
    // VOPC in its VOP3 form: destination VCC or an SGPR pair, optional |src0|
    void vop3cmpG(uint32_t op, const char* name, const Src& sdst, const Src& s0, bool abs0, const Src& s1) {
        tally(name);
        w_.push_back(0xd0000000u | (op << 16) | (abs0 ? 0x100u : 0u) | (sdst.code & 0xffu));
        w_.push_back(s0.code | (s1.code << 9));
        ++count_;
        if (text_) line(std::string(name) + " " + sdst.text + ", " + (abs0 ? "|" + s0.text + "|" : s0.text) + ", " + s1.text);
    }
    // VOPC in its VOP3 form with an SGPR-pair destination
    void vop3cmpTo(uint32_t op, const char* name, int sdst, const Src& s0, const Src& s1) {
        tally(name);
        w_.push_back(0xd0000000u | (op << 16) | (uint32_t)sdst);
        w_.push_back(s0.code | (s1.code << 9));
        ++count_;
        if (text_) line(std::string(name) + " s[" + std::to_string(sdst) + ":" + std::to_string(sdst + 1) + "], " + s0.text + ", " + s1.text);
    }
    void sopc(uint32_t op, const char* name, const Src& s0, const Src& s1) {
        w_.push_back(0xbf000000u | (op << 16) | (s1.code << 8) | s0.code);
        if (s1.hasLit) w_.push_back(s1.lit);
        else if (s0.hasLit) w_.push_back(s0.lit);
        ++count_;
        if (text_) line(std::string(name) + " " + s0.text + ", " + s1.text);
    }
    // SOPK with an SGPR destination (s_getreg_b32: simm16 = {size - 1, offset, register id})
    void sopk(uint32_t op, const char* name, int sdst, uint32_t simm, const std::string& operandText) {
        w_.push_back(0xb0000000u | (op << 23) | ((uint32_t)sdst << 16) | (simm & 0xffffu));
        ++count_;
        if (text_) line(std::string(name) + " s" + std::to_string(sdst) + ", " + operandText);
    }
    void sopp(uint32_t op, const char* name, uint32_t simm, bool showImm, const std::string& text = std::string()) {
        w_.push_back(0xbf800000u | (op << 16) | (simm & 0xffffu));
        ++count_;
        if (text_) line(!text.empty() ? text : (showImm ? std::string(name) + " " + std::to_string(simm) : std::string(name)));
    }

  private:
    void put(uint32_t word, const Src& maybeLit) {
        w_.push_back(word);
        if (maybeLit.hasLit) w_.push_back(maybeLit.lit);
        ++count_;
    }
    void line(const std::string& t) {
        if (text_) lines_.push_back(t);
    }

public:
    // a comment line of the listing (no code): the exact streams mark their sync points with it (tests/test_xlate.py walks them)
    void note(const char* what) {
        if (text_) lines_.push_back(std::string("; ") + what);
    }

private:
    bool listing() const { return text_ != nullptr; }
    std::vector<uint32_t>& w_;
    std::string* text_;
    std::vector<std::string> lines_;
    int count_ = 0, valu_ = 0, valuSlow_ = 0;
    long valuClocksX100_ = 0;
    bool cold_ = false;
    int coldDepth_ = 0;
    const std::vector<std::pair<uint32_t, int>>* pool_ = nullptr;
    // Issue cost of a wave64 VALU instruction on a busy SIMD, in clocks x 100.  Measured on MI355X as the time an instruction
    // adds to a realistic mix at four waves per SIMD (tools/micro/mix_cost.hip; the homogeneous loops of valu_rate.hip
    // bound it from above) and scaled so that the table reproduces that mix's own time (12 instructions in 14.8 ns at
    // 2.35 GHz): plain fp32 add / sub / mul, moves, 32-bit integer add / sub / and / or / xor 2.05; v_fma_f32 2.4;
    // v_med3 / min / max 2.6; conversions to and from fp64 and all fp64 arithmetic 4.25; everything else (compares,
    // integer conversions, left shifts, selects, carries) 3.95.
    static bool startsWith(const char* name, const char* prefix) { return std::strncmp(name, prefix, std::strlen(prefix)) == 0; }
    static int issueCost(const char* name) {
        static const char* const fast[] = {"v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_and_b32",
                                           "v_or_b32", "v_xor_b32", "v_lshrrev_b32"};
        for (const char* f : fast)
            if (startsWith(name, f)) return 205;
        if (startsWith(name, "v_fma_f32")) return 240;
        if (startsWith(name, "v_med3_f32") || startsWith(name, "v_max_f32") || startsWith(name, "v_min_f32")) return 260;
        if (std::strstr(name, "f64")) return 425;
        return 395;
    }
    void tally(const char* name) {
        if (cold_) return;
        ++valu_;
        const int cost = issueCost(name);
        if (cost > 300) ++valuSlow_;
        valuClocksX100_ += cost;
    }
};

// gfx950 opcodes used (checked against llvm-mc by tests/test_xlate.py, which re-assembles the listing)
enum : uint32_t {
    VOP2_CNDMASK = 0, VOP2_ADD_F32 = 1, VOP2_SUB_F32 = 2, VOP2_SUBREV_F32 = 3, VOP2_MUL_F32 = 5, VOP2_MAX_I32 = 0x0d, VOP2_ADD_U32 = 0x34,
    VOP1_MOV = 1, VOP1_CVT_F32_F64 = 0x0f, VOP1_CVT_F64_F32 = 0x10,
    VOPC_CMP_U_F32 = 0x48, VOPC_CMP_EQ_U32 = 0xca, VOPC_CMP_NE_U32 = 0xcd, VOPC_CMP_LE_U32 = 0xcb, VOPC_CMP_LE_F32 = 0x43, VOPC_CMP_GT_U32 = 0xcc,
    SOPC_CMP_LG_U64 = 0x13, SOPP_CBRANCH_SCC1 = 5, SOPP_CBRANCH_VCCZ = 6, SOPP_CBRANCH_VCCNZ = 7,
    SOP2_ADD_I32 = 2, SOP2_SUB_I32 = 3, SOP2_MIN_I32 = 6, SOP2_MIN_U32 = 7, SOP2_CSELECT_B32 = 0x0a, SOP2_OR_B32 = 0x0e, SOP2_OR_B64 = 0x0f, SOP2_LSHL_B32 = 0x1c, SOP2_LSHR_B32 = 0x1e,
    SOPC_CMP_GT_I32 = 2, SOPC_CMP_GE_I32 = 3, SOPC_CMP_LT_I32 = 4, SOPC_CMP_EQ_U32 = 6, SOPC_CMP_GE_U32 = 9, SOPC_CMP_LT_U32 = 0x0a, SOP2_MUL_I32 = 0x24, SOPP_BRANCH = 2, SOPP_CBRANCH_SCC0 = 4, SOPP_WAITCNT = 0x0c,
    VOP3_CMP_NLE_F32 = 0x4c, VOP1_READFIRSTLANE = 2,
    VOP1_CVT_F32_U32 = 6, VOPC_CMP_LT_F32 = 0x41, VOPC_CMP_EQ_F32 = 0x42, VOPC_CMP_GT_F32 = 0x44, VOP3_CMP_EQ_F32 = 0x42, VOP3_CMP_GT_F32 = 0x44,
    SOP2_AND_B32 = 0x0c, SOP2_AND_B64 = 0x0d, SOP2_ANDN2_B64 = 0x13,
    VOP2_ADDC_CO_U32 = 0x1c, VOP3B_SUBBREV_CO_U32 = 0x11e,
    VOP1_CVT_I32_F32 = 8, VOP2_LSHLREV_B32 = 0x12, VOP2_SUB_U32 = 0x35, VOP3_MED3_I32 = 0x1d7, VOPC_CMP_GE_F32 = 0x46, VOPC_CMP_NGE_F32 = 0x49, VOPC_CMP_NGT_F32 = 0x4b, VOPC_CMP_NLE_F32 = 0x4c,
    VOP3_CMP_LT_F32 = 0x41, VOP3_CMP_NLT_F32 = 0x4e, GLOBAL_LOAD_DWORDX2 = 0x15, GLOBAL_LOAD_DWORDX4 = 0x17, DS_READ_B64 = 0x76, DS_READ_B128 = 0xff, VOP2_OR_B32 = 0x14, VOP2_AND_B32 = 0x13, VOP2_XOR_B32 = 0x15, VOP2_LSHRREV_B32 = 0x10, VOPC_CMP_CLASS_F32 = 0x10, GLOBAL_LOAD_DWORD = 0x14, GLOBAL_STORE_DWORD = 0x1c,
    VOP3_CNDMASK = 0x100, VOP3_MED3_F32 = 0x1d6, VOP3_FMA_F32 = 0x1cb, VOP3_FMA_F64 = 0x1cc, VOP3_ADD_F64 = 0x280, VOP3_MUL_F64 = 0x281,
    SOP1_MOV_B32 = 0, SOP1_MOV_B64 = 1, SOP1_SETPC = 0x1d, SOP1_FLBIT_I32_B32 = 0x12, SOP2_MAX_I32 = 8,
    SOP2_ADD_U32 = 0, SOP2_SUB_U32 = 1, SOP2_ADDC_U32 = 4,
    SOPP_NOP = 0, SOPP_IDX_OFF = 0x1c,
    VOP2_ASHRREV_I32 = 0x11, VOP1_FLOOR_F32 = 0x1f, VOP1_CVT_F32_I32 = 5, VOPC_CMP_EQ_U32_ = 0xca, VOPC_CMP_GT_I32 = 0xc4,
};

// register conventions shared with fx_interp_gfx950.S
constexpr int kRegFileBase = 32;  // v32 = row 0
constexpr int kVNumSkip = 14, kVShadowCount = 15;
constexpr int kSRecord = 16;      // s16.. = record window of handler set _a: s18..s23 = w2..w7
constexpr int kSReturn = 24;      // s[24:25] = where a handler of set _a continues
constexpr int kSEntry = 32;       // s[32:33] = address of the kernel entry
constexpr int kSEndSample = 34;   // s[34:35] = end-of-sample frame
constexpr int kSTemp = 62;        // s[62:63] scratch of the handlers, free between them
constexpr int kSTaint = 78;       // s[78:79] lanes that hold a non-finite value (template prologue)
constexpr int kVLane4 = 1;        // v1 = lane * 4
constexpr int kVClassMask = 29;   // v29 = v_cmp_class mask of NaN and +-Inf
constexpr int kVOod = 22;         // v22 = out-of-domain flags of the lane
constexpr int kVCursor = 16;      // v16..v19 = TRAM cursors: iTRAM write, iTRAM read, xTRAM write, xTRAM read
constexpr int kSCursor = 80;      // s80..s83 = the same cursors while a stream with uniform cursors runs
constexpr int kSPos = 84, kSOod = 85, kSAddr = 86;  // scratch of the inline TRAM code (s[86:87] = slot address)
// LUT tables in LDS (the tables a program uses, copied once per workgroup by the run-once code).  Two layouts:
//   narrow (round 2 .. 4): every array indexed by segment * 8 bytes - {xthr[g], xthr[g+1]} fp32 pairs | x1[64] fp64 | per table
//           slope[64] fp64, y1[64] fp64: three ds_read_b64 per LOG / EXP (x1, slope, y1);
//   wide:   every array indexed by segment * 16 bytes - thresholds and x1 in the first half of a 16-byte cell each, per table ONE
//           array of {slope, y1} cells: a ds_read_b64 (x1) and a ds_read_b128 per LOG / EXP - two LDS instructions instead of
//           three for the same 24 bytes per lane.  Measured 0.5 % slower (three times the bank-conflict cycles:
//           profiles/r05_lut_wide_ab.txt): the narrow layout stays in force, the wide one is a switch of the diagnostics build.
struct LutLdsLayout {
    bool wide;
    uint32_t thr, x1, tables, tableBytes;   // byte offsets of the threshold pairs, of x1[], of the first table; bytes per table
    int shift;                              // log2 of the bytes per segment cell
    uint32_t bytes(size_t nTables) const { return tables + (uint32_t)nTables * tableBytes; }
};
constexpr LutLdsLayout kLutNarrow{false, 0, 512, 1024, 1024, 3}, kLutWide{true, 0, 1024, 2048, 1024, 4};
const LutLdsLayout& lutLds();   // the layout in force (fx_xlate.cpp)
constexpr int kSLut = 40;          // s[40:41] = LUT blob
constexpr int kSLutXthr = 88, kSLutX1 = 90, kSLutSeg = 92;  // s[88:93]: bases of the fp32 thresholds, x1[] and the current table's segments
constexpr int kSTramBase[2] = {36, 38}, kSTramSize[2] = {56, 57}, kSTramSlots[2] = {46, 47};  // [iTRAM, xTRAM]
// the sample loop (frame registers of fx_interp_gfx950.S)
constexpr int kSSample = 3, kSNumSamples = 9;       // sample index, block length
constexpr int kSPcmIn = 12, kSPcmOut = 14;          // s[12:13] / s[14:15]: PCM in / out of the current sample
constexpr int kSSampleBytes = 45, kSChannelBytes = 68;  // bytes per sample (channels * N * 4) and per channel-sample (N * 4)
constexpr int kSValidLanes = 58;                    // s[58:59]: lanes that hold an instance
constexpr int kVInput = 23;                         // v23..v26: PCM input of the current sample, channel 0..3 (requested one sample ahead)
constexpr int kVInstance4 = 27;                     // v27 = instance * 4: byte offset into a PCM / state row
constexpr int kSPrefetched = 94;                    // s94 = 1: the leading TRAM reads of this sample are already in flight
constexpr int kSEventNext = 28;                         // control tracks: the sample at which the next event of the block's list is due (0xFFFFFFFF: none left)
constexpr int kSEventPtr = 26;                          // s[26:27]: address of that event's record (fx_xlate.hpp TrackEvent)
constexpr int kKernargTracks = 0xb8;                    // AsmArgs.tracks (fx_asm.hpp)
constexpr int kSSliceShift = 8;                        // unstaged programs with time-sliced priorities: log2 of a slice in 100 MHz ticks (emitInit)
constexpr int kSHoistOk = 95;                       // s95 = 1: this launch may issue leading TRAM reads one sample ahead (emitInit)
constexpr int kVRing = 30;                          // staged programs: lane * 4 + the LDS buffer of this sample's packets (sent and requested, see stageRequest)
// staged programs (s4..s8 are the template's dispatch scratch and the interpreter's fetch offset: free in generated code)
// A steady stream: s7 counts down to the next event - the group's barrier or the end of the steady stream - from s5 - 1; when it
// borrows, s8 (samples left in the group) and s6 (steady samples left) both go down by s5 and whichever reached 0 is served.
// A last-sample stream: s7 = samples of the group still to come after this one (the barrier follows when it borrows).
constexpr int kSGroupLeft = 7;
constexpr int kSSteadyLeft = 6;
constexpr int kSLoaded = 5, kSGroupSamples = 8;

}  // namespace xl
}  // namespace fx
