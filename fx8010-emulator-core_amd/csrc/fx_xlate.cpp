// fx_xlate.cpp — FX8010 program -> gfx950 machine code (see fx_xlate.hpp): the Translator (one stream of records -> one sample
// loop), the run-once code, hoist planning, row classes (xlateProgramOf) and the layout of a program's four streams
// (planXlate).  The encoder is fx_xlate_emit.hpp, the template images fx_xlate_elf.cpp, the stage planner fx_xlate_stages.cpp.
#include "fx_xlate.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>

#include "fx_knobs.hpp"
#include "fx_xlate_emit.hpp"
#include "fx_xlate_internal.hpp"

namespace fx {
using namespace xl;

// the LDS layout of the LOG / EXP tables in force (fx_xlate_emit.hpp): the narrow one.  Measured (tools/lut_wide_ab.sh,
// profiles/r05_lut_wide_ab.txt: config4, alternating runs inside one call): the wide layout issues a third fewer LDS
// instructions (283 -> 189 per wave-sample) but its 16-byte cells triple the bank-conflict cycles (123 -> 375) and the launch
// is 0.5 % SLOWER (49.03 -> 49.27 ms; SQ_WAIT_ANY 31 -> 36 % of wavefront time).  The diagnostics build can still pick it
// (FX_XLATE_LUTWIDE=1) to repeat the A/B; the release library has the default only.
const LutLdsLayout& xl::lutLds() {
    static const bool wide = knobInt(FX_DIAG_KNOB("FX_XLATE_LUTWIDE"), 0) != 0;
    return wide ? kLutWide : kLutNarrow;
}

namespace {

// the four VGPR constants of the quick LOG/EXP index guess (Translator::lut): cell bytes * 31.5, that - 0.5, the rounding
// constant and the mask of the cell offset - for 8-byte cells (narrow layout) and 16-byte cells (wide layout)
struct LutGuessConstants {
    uint32_t scale, bias, magic, mask;
    bool available(const Emitter& e) const { return e.pooled(scale) >= 0 && e.pooled(bias) >= 0 && e.pooled(magic) >= 0 && e.pooled(mask) >= 0; }
};
constexpr LutGuessConstants kLutGuessNarrow{0x437c0000u /* 252.0 */, 0x437b8000u /* 251.5 */, 0x4b400000u /* 1.5 * 2^23 */, 0x000001f8u};
constexpr LutGuessConstants kLutGuessWide{0x43fc0000u /* 504.0 */, 0x43fbc000u /* 503.5 */, 0x4b400000u, 0x000003f0u};
const LutGuessConstants& lutGuess() { return lutLds().wide ? kLutGuessWide : kLutGuessNarrow; }

class Translator {
  public:
    // exactReturns == nullptr: the exact stream (NaN passes every saturation).  Otherwise the fast stream, which
    // assumes finite register contents and leaves for the exact stream - at the same point there,
    // exactReturns[key] - as soon as the wave is tainted.
    Translator(const XlateTemplate& t, const XlateProgram& prog, uint32_t codeBase, bool isLast, uint32_t nextBase, std::vector<uint32_t>* code,
               std::string* listing, const std::vector<uint32_t>* exactReturns)
        : tmpl_(t), prog_(prog), base_(codeBase), isLast_(isLast), nextBase_(nextBase), e_(code, listing), fast_(exactReturns != nullptr),
          exactReturns_(exactReturns) { e_.constants(&prog_.vconst); }

    // Layout of a stream:  head (hot entry) | program | PCM out, advance, loop branch / exit | cold entry stub
    bool run(const std::vector<MicroOp>& records, XlateStats* stats, std::vector<uint32_t>* returns, uint32_t* coldEntry, std::string* err) {
        // sync points: per record i [4i] after the wait for TRAM reads pending before it, [4i+1] behind its handler call /
        // inline LUT, [4i+2] after the wait in front of the early reads that follow it; [4n] the head of the loop body
        // (inputs and leading reads consumed, next input requested), [4n+3] after the wait for reads still pending behind
        // the last instruction
        returns_.assign(4 * records.size() + 4, 0);
        records_ = &records;
        const HoistPlan& H = prog_.hoist;
        const int channels = (int)prog_.latchRows.size();
        if (channels < 1 || prog_.inRows.size() != prog_.latchRows.size()) { if (err) *err = "internal: channel rows missing"; return false; }
        bool usesSkipCounter = false, anyLut = false;
        for (const MicroOp& r : records) {
            usesSkipCounter = usesSkipCounter || r.w[0] == AS_PRED || r.w[0] == AS_SKIP;
            anyLut = anyLut || r.w[0] == AS_LUT;
        }
        buildConstantPool(records, anyLut);
        // staged programs: when the record words s22:s23 are only ever one value - the (1 - X) of the stage's INTERPs with one
        // uniform X - and nothing calls a handler, the pair is loaded by the cold entry instead of in every sample
        bool hoistOmx = false;
        uint32_t omxLo = 0, omxHi = 0;
        if (prog_.stage.count > 1) {
            bool simple = true, any = false;
            for (const MicroOp& r : records) {
                const uint32_t slot = r.w[0];
                if (slot == AS_ENDSAMPLE) break;
                if (slot == AS_NOP || slot == AS_PRED || slot == AS_UNPRED || slot == AS_MOV || slot == AS_LIMIT || slot == AS_LIMITN) continue;
                if (slot == AS_SKIP) { float w; int32_t c; if (!uniformSkip(r, &w, &c)) simple = false; continue; }
                if (slot == AS_LUT) { if ((r.w[6] & 1u) || !(r.w[6] & 2u)) simple = false; continue; }
                if (slot < AS_MACS || slot >= (uint32_t)kAsmSlots) { simple = false; continue; }
                const uint32_t rel = slot - AS_MACS, family = rel / 16, kind = (rel % 16) / 2;
                if (family != 3 || !(kind & 2u) || kind == 7u) continue;
                bool inlineConst = false;
                const uint64_t omx = (uint64_t)r.w[6] | ((uint64_t)r.w[7] << 32);
                for (const InlineD& k : kInlineF64) inlineConst = inlineConst || k.bits == omx;
                if (inlineConst) continue;
                if (any && (r.w[6] != omxLo || r.w[7] != omxHi)) simple = false;
                any = true;
                omxLo = r.w[6];
                omxHi = r.w[7];
            }
            hoistOmx = simple && any;
            if (hoistOmx) {
                omxHoisted_ = true;
                hoistedLo_ = omxLo;
                hoistedHi_ = omxHi;
                known_[6] = known_[7] = true;
                value_[6] = omxLo;
                value_[7] = omxHi;
            }
        }
        // ---- head: this sample's operands that come from memory
        const size_t headWord = e_.words();
        const StageInfo& G = prog_.stage;
        const bool staged = G.count > 1;
        int storesPerSample = 0;
        for (int c = 0; c < channels; ++c) storesPerSample += (G.storeMask >> c) & 1u;
        const bool ring = staged && G.inRing >= 0;
        if (H.leadCount > 0) {
            // the leading TRAM reads were issued one sample ago (s94 = 1) or are issued here (first sample of a launch, or
            // a launch whose cursor distance rules the early issue out)
            e_.sopc(SOPC_CMP_EQ_U32, "s_cmp_eq_u32", sreg(kSPrefetched), imm32(1));
            Emitter::Fixup inPlace = e_.branchForward(SOPP_CBRANCH_SCC0, "s_cbranch_scc0");
            e_.waitVmcnt(H.vmemAfterHoist + storesPerSample);  // younger than the reads: the rest of that sample's TRAM traffic and its PCM stores
            // (the in-place reads sit behind the loop: a steady sample falls through)
            const int leadCount = H.leadCount;
            defer(inPlace, [this, &records, leadCount]() {
                e_.cold(true);
                for (int k = 0; k < leadCount; ++k)
                    if (!tramRead(records[(size_t)k], records[(size_t)k].w[0], false)) deferredFailed_ = true;
                e_.waitVmcnt(0);
                e_.cold(false);
            });
            e_.sop1(SOP1_MOV_B32, "s_mov_b32", sreg(kSPrefetched), imm32(0));
        } else if (!ring) {
            bool anyInput = false;
            for (int c = 0; c < channels; ++c) anyInput = anyInput || prog_.inRows[(size_t)c] >= 0;
            // the PCM input requested one sample ago (a stage without input has nothing to wait for: its PCM stores may take
            // longer than one of its steps)
            if (anyInput || !staged) e_.waitVmcnt(prog_.tramOpsInline + storesPerSample);
        }
        if (!prog_.trackRows.empty() && !trackStep()) { if (err) *err = err_; return false; }
        if (staged && !G.recvRows.empty()) {
            // the rows the previous stage handed over for this sample were read into spare registers one step ago; the
            // next sample's are requested now and land behind this step's work (StageInfo).  (Younger than that request:
            // the packet written at the end of the last step, which the next barrier's wait covers.)
            e_.waitLgkm(packetOps(G.sendRows.size(), G.sendOff - G.ptrBias, -1));
            for (size_t i = 0; i < G.recvRows.size(); ++i) {
                int v;
                if (!row((uint32_t)G.recvRows[i], &v)) { if (err) *err = err_; return false; }
                e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(v), vreg(G.recvTmp + (int)i));
            }
            // (what arrives is not checked row by row: a stage whose packets may hold anything but clean values - it runs its
            // exact stream - says so in its flag row, which the receiver reads behind every barrier: stageFlagCheck)
            if (!isLast_) stageRequest();
        }
        if (ring) {
            if (!inputFromRing(storesPerSample)) { if (err) *err = err_; return false; }
        } else {
            if (fast_)
                for (int c = 0; c < channels; ++c)
                    if (prog_.inRows[(size_t)c] >= 0) taintIfNonFinite(kVInput + c);
            for (int c = 0; c < channels; ++c) {
                int v;
                if (prog_.inRows[(size_t)c] < 0) continue;
                if (!row((uint32_t)prog_.inRows[(size_t)c], &v)) { if (err) *err = err_; return false; }
                e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(v), vreg(kVInput + c));
            }
            bool anyInput = false;
            for (int c = 0; c < channels; ++c) anyInput = anyInput || prog_.inRows[(size_t)c] >= 0;
            if (!isLast_ && (anyInput || !staged)) pcmAccess(true, kSPcmIn, true);  // next sample's input; it lands behind this sample's program
        }
        if (fast_)
            for (int k = 0; k < H.leadCount; ++k) taintCheckRow(vrow(records[(size_t)k].w[5]));
        if (usesSkipCounter) e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(kVNumSkip), imm32(0));  // numSkip is local to process() (FX8010.cpp:1030)
        index_ = records.size();
        syncPoint(syncIndex(0), base_ + (uint32_t)e_.bytes());
        // (a later stage of a pipelined program has nothing at its head that could taint it: its flag check sits behind the barriers)
        const bool cleanHead = staged && G.index > 0 && H.leadCount == 0 && !ring && prog_.trackRows.empty();
        if (fast_ && !cleanHead && !leaveIfTainted((*exactReturns_)[syncIndex(0)])) { if (err) *err = err_; return false; }

        // ---- the program
        // Results nobody reads: a row written again, unconditionally, before any instruction of the same sample reads it (the
        // end of the sample counts as a reader: state, packets, the next sample).  The canonical case is the DANE idiom for
        // "test a value": `macs tmp, x, 0, 0` + `skip ccr, ccr, <cond>, n` - tmp is written once per test and never read; the
        // instruction is there for its CCR, which the SKIP's predicate takes straight from x (one()).
        deadWrite_.assign(records.size(), 0);
        {
            bool shadow = false;
            std::vector<uint8_t> shadowed(records.size(), 0);
            for (size_t i = 0; i < records.size(); ++i) {
                if (records[i].w[0] == AS_PRED) shadow = true;
                else if (records[i].w[0] == AS_UNPRED) shadow = false;
                shadowed[i] = shadow;
            }
            for (size_t i = 0; i < records.size() && records[i].w[0] != AS_ENDSAMPLE; ++i) {
                const uint32_t slot = records[i].w[0];
                if (slot < AS_MACS || slot >= (uint32_t)kAsmSlots || shadowed[i]) continue;
                const Access a = accessOf(records[i]);
                if (a.write <= 0) continue;   // (row 0 is the CCR)
                for (size_t j = i + 1; j < records.size(); ++j) {
                    if (records[j].w[0] == AS_ENDSAMPLE) break;
                    const Access b = accessOf(records[j]);
                    bool reads = false;
                    for (int k = 0; k < b.nReads; ++k) reads = reads || (int)b.reads[k] == a.write;
                    if (reads || ((b.tram || b.noise) && b.write == a.write)) break;
                    if (b.write == a.write && !shadowed[j] && !b.tram && !b.noise) { deadWrite_[i] = 1; break; }
                }
            }
            for (int r : prog_.latchRows)   // (an output latch is stored as PCM every sample)
                for (size_t i = 0; i < records.size(); ++i)
                    if (deadWrite_[i] && (int)records[i].w[5] == r) deadWrite_[i] = 0;
        }
        products_.assign((size_t)(fast_ ? prog_.cseEntries : 0), Product());
        for (size_t k = 0; k < products_.size(); ++k) {
            products_[k].vP64 = prog_.cseBase + 2 * (int)k;
            products_[k].vP = prog_.cseBase + 2 * prog_.cseEntries + (int)k;
        }
        bool ended = false;
        for (size_t i = 0; i < records.size(); ++i) {
            const MicroOp& r = records[i];
            const uint32_t slot = r.w[0];
            index_ = i;
            if (slot == AS_ENDSAMPLE) { ended = true; break; }
            if ((int)i >= H.leadCount && i != consumed_ && !one(r, slot)) { if (err) *err = err_; return false; }
            if (slot != AS_NOP && slot != AS_PRED && slot != AS_UNPRED && slot != AS_SKIP) cseKill(r.w[5]);  // (its R row, if it has one)
            if (!isLast_ && H.leadCount > 0 && (int)i == std::max(H.hoistAfter, H.leadCount - 1)) {
                // the next sample's leading reads, unless this launch keeps them in place (s95 = 0, see emitInit)
                if (predOpen_ || regionPreds_ > 0) { if (err) *err = "internal: hoist point inside a SKIP shadow"; return false; }
                if (!flush(2)) { if (err) *err = err_; return false; }
                plainMode();
                e_.sopc(SOPC_CMP_EQ_U32, "s_cmp_eq_u32", sreg(kSHoistOk), imm32(0));
                Emitter::Fixup skip = e_.branchForward(SOPP_CBRANCH_SCC1, "s_cbranch_scc1");
                for (int k = 0; k < H.leadCount; ++k)
                    if (!tramRead(records[(size_t)k], records[(size_t)k].w[0], false, true)) { if (err) *err = err_; return false; }
                for (int k = 0; k < H.leadCount; ++k) cseKill(records[(size_t)k].w[5]);
                e_.sop1(SOP1_MOV_B32, "s_mov_b32", sreg(kSPrefetched), imm32(1));
                e_.bind(skip);
            }
        }
        if (!ended) { if (err) *err = "record stream without ENDSAMPLE"; return false; }
        index_ = records.size();
        // (sync point [4n + 3]: [4n] is the head's - a wave that leaves the fast stream at the head must not land here)
        if (!flush(3)) { if (err) *err = err_; return false; }

        // ---- PCM out, next sample
        plainMode();
        {
            // diagnostics: FX_XLATE_LOOPPAD=n / FX_XLATE_LOOPPAD_SLOW=n put n independent instructions (plain / 4-clock class, on a
            // spare register) into every sample of every stream - does a stage's loop pay for issue slots or for the latency of
            // its dependent chain?  (tools/stage_pad_probe.sh)
            static const int pad = knobInt(FX_DIAG_KNOB("FX_XLATE_LOOPPAD"), 0);
            static const int padSlow = knobInt(FX_DIAG_KNOB("FX_XLATE_LOOPPAD_SLOW"), 0);
            for (int k = 0; k < pad; ++k) e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", 31, vreg(31), 31);
            for (int k = 0; k < padSlow; ++k) e_.vop1(VOP1_CVT_F32_U32, "v_cvt_f32_u32_e32", vreg(31), vreg(31));
            // ... FX_XLATE_LOOPPAD_SALU=n: n scalar instructions (does a wavefront's scalar work - delay-line addresses, loop
            // control - cost the SIMD issue time, or do the other wavefronts' vector instructions go out beside it?)
            static const int padScalar = knobInt(FX_DIAG_KNOB("FX_XLATE_LOOPPAD_SALU"), 0);
            for (int k = 0; k < padScalar; ++k) e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSTemp), sreg(kSTemp), imm32(1));
        }
        if (!staged || usesSkipCounter) e_.sop1(SOP1_MOV_B64, "s_mov_b64", named(126, "exec"), named(193, "-1"));
        if (staged) {
            // hand this sample's live rows to the next stage; ONE barrier per step on every path through a sample (all
            // wavefronts of the workgroup execute the same number of them).  The wait in front of it covers LAST step's writes
            // and this step's request - everything but the writes just issued.
            // (an exact stream's packets may hold non-finite values: its flag row tells the next stage, before the packets do)
            if (!fast_ && G.index + 1 < G.count) flagAccess(false, kVClassMask, G.flagBase + 256u * (uint32_t)G.index, 5);
            {
                std::vector<int> regs;
                for (size_t i = 0; i < G.sendRows.size(); ++i) {
                    int v;
                    if (!row((uint32_t)G.sendRows[i], &v)) { if (err) *err = err_; return false; }
                    regs.push_back(v);
                }
                packetIo(false, regs, G.sendOff - G.ptrBias);
            }
            ringStep(kVRing);
        }
        // (a steady stream of a staged program: barrier and loop control by ONE down-counter, behind the PCM store - below)
        const bool oneCounter = staged && !isLast_;
        if (staged && !oneCounter) {
            // ... every group-th sample (the same samples in every wavefront: they all count from 0)
            if (G.group > 1) {
                e_.sop2(SOP2_SUB_U32, "s_sub_u32", sreg(kSGroupLeft), sreg(kSGroupLeft), imm32(1));   // SCC = borrow: this was the group's last sample
                Emitter::Fixup within = e_.branchForward(SOPP_CBRANCH_SCC0, "s_cbranch_scc0");
                if (!G.sendRows.empty()) e_.waitLgkm(packetOps(G.sendRows.size(), G.sendOff - G.ptrBias, -1));
                e_.barrier();
                e_.sop1(SOP1_MOV_B32, "s_mov_b32", sreg(kSGroupLeft), imm32((uint32_t)G.group - 1u));
                if (!isLast_ && !stageFlagCheck(1)) { if (err) *err = err_; return false; }
                e_.bind(within);
            } else {
                if (!G.sendRows.empty()) e_.waitLgkm(packetOps(G.sendRows.size(), G.sendOff - G.ptrBias, -1));
                e_.barrier();
                if (!isLast_ && !stageFlagCheck(1)) { if (err) *err = err_; return false; }
            }
        }
        if (storesPerSample > 0) pcmAccess(false, kSPcmOut, false);
#ifdef FX_DIAGNOSTICS
        {
            // diagnostics build only (FX_XLATE_ENDSTAMP=1): when does each wavefront finish?  Behind the last sample's PCM store the
            // low word of the 100 MHz clock goes to word [wavefront] of a buffer of ITS OWN - the batch hands it over in the kernarg
            // slot of the stage descriptors, which an unstaged launch does not use (fx_batch.cpp; read back with
            // fxb_diag_read_stamps, tools/wave_end_probe.py) - never into an output element (MI355X_MICROARCH.md on stamps)
            static const bool stamp = FX_DIAG_KNOB("FX_XLATE_ENDSTAMP") != nullptr;
            if (stamp && isLast_ && !staged) {
                constexpr int kKernargStages = 0xc0;   // AsmArgs.stages (fx_asm.hpp)
                smemLoad(2, kSTemp, 0, kKernargStages);
                e_.memRealTime(kSTemp + 2);
                e_.waitLgkm0();
                e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(2), sreg(kSTemp + 2));
                e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(3), sreg(2));                       // s2 = wavefront of the launch
                e_.vop2(VOP2_LSHLREV_B32, "v_lshlrev_b32_e32", 3, imm32(2), 3);
                e_.sop1(SOP1_MOV_B64, "s_mov_b64", named(126, "exec"), imm32(1));
                e_.global(GLOBAL_STORE_DWORD, false, 2, 3, kSTemp);
                e_.sop1(SOP1_MOV_B64, "s_mov_b64", named(126, "exec"), named(193, "-1"));
            }
        }
#endif
        for (int q : {kSPcmIn, kSPcmOut}) {
            bool anyInput = false;
            for (int c = 0; c < channels; ++c) anyInput = anyInput || prog_.inRows[(size_t)c] >= 0;
            if (staged && ((q == kSPcmIn && !anyInput) || (q == kSPcmOut && storesPerSample == 0))) continue;  // a stage that does not touch that stream
            e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(q), sreg(q), sreg(kSSampleBytes));
            e_.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(q + 1), sreg(q + 1), imm32(0));
        }
        if (!staged || ring) e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSSample), sreg(kSSample), imm32(1));   // (a later stage counts down instead)
        if (prog_.prioritySlices && !staged) {
            // Time slices for the wavefronts of a SIMD.  The SIMD's arbiter serves its oldest wavefront first: in a launch that
            // fills every slot once, the first wavefront of a SIMD runs as if alone, the others take what is left, and they
            // finish one after the other (tools/wave_end_probe.py: 18 ms apart in a 31 ms launch of the headline shape) - for
            // the last third of the launch the SIMD holds three, two, one wavefront and issues like a half-empty one.  So every
            // fourth sample a wavefront reads the 100 MHz clock and takes the priority ((clock >> shift) + its wave-buffer
            // slot) & 3: at any moment the (up to four) wavefronts of a SIMD hold different levels, each the top one for the
            // same share of the time, and they reach the end together (spread 3 ms; config5 17.8 -> 19.8, config4 7.5 -> 9.2
            // e12 instr/s).  shift (s8, from the run-once code: a slice is about 1/24 of the block) grows with the block.
            e_.sop2(SOP2_AND_B32, "s_and_b32", sreg(kSTemp), sreg(kSSample), imm32(3));
            e_.sopc(SOPC_CMP_EQ_U32, "s_cmp_eq_u32", sreg(kSTemp), imm32(0));
            Emitter::Fixup due = e_.branchForward(SOPP_CBRANCH_SCC1, "s_cbranch_scc1");
            defer(due, [this]() {
                e_.cold(true);
                e_.memRealTime(kSTemp);
                e_.sopk(0x11, "s_getreg_b32", kSTemp + 2, (3u << 11) | 4u, "hwreg(HW_REG_HW_ID, 0, 4)");
                e_.waitLgkm0();
                e_.sop2(SOP2_LSHR_B32, "s_lshr_b32", sreg(kSTemp), sreg(kSTemp), sreg(kSSliceShift));
                e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSTemp), sreg(kSTemp), sreg(kSTemp + 2));
                e_.sop2(SOP2_AND_B32, "s_and_b32", sreg(kSTemp), sreg(kSTemp), imm32(3));
                std::vector<Emitter::Fixup> out;
                for (uint32_t level = 0; level < 4; ++level) {
                    Emitter::Fixup next;
                    if (level < 3) {
                        e_.sopc(SOPC_CMP_EQ_U32, "s_cmp_eq_u32", sreg(kSTemp), imm32(level));
                        next = e_.branchForward(SOPP_CBRANCH_SCC0, "s_cbranch_scc0");
                    }
                    e_.sopp(0x0f, "s_setprio", level, true);
                    if (level < 3) {
                        out.push_back(e_.branchForward(SOPP_BRANCH, "s_branch"));
                        e_.bind(next);
                    }
                }
                for (const Emitter::Fixup& f : out) e_.bind(f);
                e_.cold(false);
            });
        }
        if (prog_.tramDane) { if (prog_.uniformCursors) daneStep(); else daneStepPerLane(); }
        if (oneCounter) {
            // s7 counts the samples up to the next event - the group's barrier or the end of the steady stream, whichever
            // comes first (s5 = what it was loaded with, s8 = samples of the group, s6 = steady samples still to run when it
            // was loaded): one subtract and one branch per sample; the bookkeeping runs once per group
            e_.sop2(SOP2_SUB_U32, "s_sub_u32", sreg(kSGroupLeft), sreg(kSGroupLeft), imm32(1));   // SCC = borrow: an event is due
            if (!e_.branchBack(SOPP_CBRANCH_SCC0, "s_cbranch_scc0", headWord)) { if (err) *err = "translated loop too long for a branch"; return false; }
            e_.cold(true);
            e_.sop2(SOP2_SUB_U32, "s_sub_u32", sreg(kSGroupSamples), sreg(kSGroupSamples), sreg(kSLoaded));
            e_.sop2(SOP2_SUB_U32, "s_sub_u32", sreg(kSSteadyLeft), sreg(kSSteadyLeft), sreg(kSLoaded));
            e_.sopc(SOPC_CMP_EQ_U32, "s_cmp_eq_u32", sreg(kSGroupSamples), imm32(0));
            Emitter::Fixup within = e_.branchForward(SOPP_CBRANCH_SCC0, "s_cbranch_scc0");
            // the group's barrier: ONE per `group` samples in every wavefront of the workgroup (they all count from 0).  The
            // wait in front of it covers the packets of the previous samples - everything but the writes just issued
            if (!G.sendRows.empty()) e_.waitLgkm(packetOps(G.sendRows.size(), G.sendOff - G.ptrBias, -1));
            e_.barrier();
            e_.sop1(SOP1_MOV_B32, "s_mov_b32", sreg(kSGroupSamples), imm32((uint32_t)G.group));
            if (!stageFlagCheck(1)) { if (err) *err = err_; return false; }
            e_.bind(within);
            e_.sopc(SOPC_CMP_EQ_U32, "s_cmp_eq_u32", sreg(kSSteadyLeft), imm32(0));
            Emitter::Fixup done = e_.branchForward(SOPP_CBRANCH_SCC1, "s_cbranch_scc1");
            e_.sop2(SOP2_MIN_U32, "s_min_u32", sreg(kSLoaded), sreg(kSSteadyLeft), sreg(kSGroupSamples));
            e_.sop2(SOP2_SUB_U32, "s_sub_u32", sreg(kSGroupLeft), sreg(kSLoaded), imm32(1));
            if (!e_.branchBack(SOPP_BRANCH, "s_branch", headWord)) { if (err) *err = "translated loop too long for a branch"; return false; }
            e_.bind(done);
            // (the last-sample stream counts the rest of the group down from its length - 1)
            e_.sop2(SOP2_SUB_U32, "s_sub_u32", sreg(kSGroupLeft), sreg(kSGroupSamples), imm32(1));
            e_.cold(false);
            const int64_t delta = ((int64_t)nextBase_ - ((int64_t)base_ + (int64_t)e_.bytes() + 4)) / 4;
            if (delta < -32768 || delta > 32767) { if (err) *err = "last-sample stream out of branch range"; return false; }
            e_.sopp(SOPP_BRANCH, "s_branch", (uint32_t)delta & 0xffffu, true);
        } else if (!isLast_) {
            // loop while the sample after this one is not the block's last, then on to the last-sample stream
            if (staged) {
                e_.sop2(SOP2_SUB_U32, "s_sub_u32", sreg(kSSteadyLeft), sreg(kSSteadyLeft), imm32(1));   // SCC = borrow: the next sample is the last
                if (!e_.branchBack(SOPP_CBRANCH_SCC0, "s_cbranch_scc0", headWord)) { if (err) *err = "translated loop too long for a branch"; return false; }
            } else {
                e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSTemp), sreg(kSSample), imm32(1));
                e_.sopc(SOPC_CMP_LT_U32, "s_cmp_lt_u32", sreg(kSTemp), sreg(kSNumSamples));
                if (!e_.branchBack(SOPP_CBRANCH_SCC1, "s_cbranch_scc1", headWord)) { if (err) *err = "translated loop too long for a branch"; return false; }
            }
            const int64_t delta = ((int64_t)nextBase_ - ((int64_t)base_ + (int64_t)e_.bytes() + 4)) / 4;
            if (delta < -32768 || delta > 32767) { if (err) *err = "last-sample stream out of branch range"; return false; }
            e_.sopp(SOPP_BRANCH, "s_branch", (uint32_t)delta & 0xffffu, true);
        } else {
            if (prog_.uniformCursors) {
                for (int c = 0; c < 4; ++c) e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(kVCursor + c), sreg(kSCursor + c));
                e_.vop2(VOP2_OR_B32, "v_or_b32_e32", kVOod, sreg(kSOod), kVOod);
            }
            if (staged) {
                if (!G.sendRows.empty()) e_.waitLgkm0();                   // (the last packet: complete before the next barrier)
                for (int k = 0; k < kStageDepth * (G.count - 1 - G.index); ++k) e_.barrier();  // the later stages are still at work: their steps' barriers
            }
            e_.sop1NoDst(SOP1_SETPC, "s_setpc_b64", sreg64(kSEndSample));  // the template's epilogue
        }

        if (!emitDeferred()) { if (err) *err = err_; return false; }
        // ---- cold entry (from the template): scalar copies of what the loop keeps in SGPRs
        e_.cold(true);
        if (coldEntry) *coldEntry = base_ + (uint32_t)e_.bytes();
        if (prog_.uniformCursors) {
            // every TRAM instruction runs on all lanes: the four cursors are the same in every lane, keep them in SGPRs
            for (int c = 0; c < 4; ++c) e_.vop1(VOP1_READFIRSTLANE, "v_readfirstlane_b32", sreg(kSCursor + c), vreg(kVCursor + c));
            e_.sop1(SOP1_MOV_B32, "s_mov_b32", sreg(kSOod), imm32(0));
        }
        if (staged) {
            // stage k starts 3k steps late; packet s lives in buffer s & 3.  The step in front of its first sample requests
            // that sample's rows.
            // (the request in front of the first sample: packet 0 of the cut in front of this stage, buffer `index`; then the
            // pointer stands at index + 1 for sample 0)
            if (G.index + 1 < G.count) {
                // this stage's flag row: clean so far (the next stage first reads it behind its last barrier in front of its first
                // sample - behind this wavefront's first barrier, whichever that is)
                e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(2), imm32(0));
                flagAccess(false, 2, G.flagBase + 256u * (uint32_t)G.index, 3);
                e_.waitLgkm0();
            }
            // (the ring has 4 * group buffers: fewer than stages when the LDS budget allows only a group of one or two)
            e_.vop2(VOP2_ADD_U32, "v_add_u32_e32", kVRing, imm32(((uint32_t)G.index % (4u * (uint32_t)G.group)) * G.bufStride + G.ptrBias), kVLane4);
            for (int k = 0; k + 1 < kStageDepth * G.index; ++k) e_.barrier();
            if (G.index > 0) {
                if (!G.recvRows.empty()) stageRequest();
                e_.barrier();
                index_ = records.size();
                if (!stageFlagCheck(2)) { if (err) *err = err_; return false; }
            }
            ringStep(kVRing);
            if (isLast_) {
                e_.sop1(SOP1_MOV_B32, "s_mov_b32", sreg(kSGroupLeft), imm32((uint32_t)G.group - 1u));
            } else {
                e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSSteadyLeft), sreg(kSNumSamples), imm32(0xffffffffu));   // nSamples - 1 steady samples (>= 1)
                e_.sop1(SOP1_MOV_B32, "s_mov_b32", sreg(kSGroupSamples), imm32((uint32_t)G.group));
                e_.sop2(SOP2_MIN_U32, "s_min_u32", sreg(kSLoaded), sreg(kSSteadyLeft), sreg(kSGroupSamples));
                e_.sop2(SOP2_SUB_U32, "s_sub_u32", sreg(kSGroupLeft), sreg(kSLoaded), imm32(1));
            }
            if (hoistOmx) {   // the one (1 - X) of the stage's INTERPs: set once, not once per sample
                e_.sop1(SOP1_MOV_B32, "s_mov_b32", sreg(kSRecord + 6), imm32(omxLo));
                e_.sop1(SOP1_MOV_B32, "s_mov_b32", sreg(kSRecord + 7), imm32(omxHi));
            }
            if (ring) inputBurst(0, true);
        }
        if (!prog_.trackRows.empty()) trackInit();
        // (inline constants where the bit pattern has one, as the assembler would choose: the listing must re-assemble to the same bytes)
        for (const auto& c : pool_) e_.sop1(SOP1_MOV_B32, "s_mov_b32", sreg(c.second), imm32(c.first));
        for (const auto& c : prog_.vconst) e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(c.second), imm32(c.first));
        if (anyLut && prog_.lutTables.empty()) {
            e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSLutXthr), sreg(kSLut), imm32((uint32_t)kLutXthrOff * 8, true));
            e_.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSLutXthr + 1), sreg(kSLut + 1), imm32(0));
            e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSLutX1), sreg(kSLut), imm32((uint32_t)kLutX1Off * 8, true));
            e_.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSLutX1 + 1), sreg(kSLut + 1), imm32(0));
        }
        if (!e_.branchBack(SOPP_BRANCH, "s_branch", headWord)) { if (err) *err = "translated loop too long for a branch"; return false; }
        e_.finish();
        if (returns) *returns = returns_;
        if (stats) { *stats = stats_; stats->instructions = e_.count(); stats->valu = e_.valu(); stats->valuSlow = e_.valuSlow(); stats->valuClocks = e_.valuClocks(); stats->nonFiniteImmediate = nonFinite_; }
        return true;
    }

  private:
    bool fail(const std::string& m) { err_ = m; return false; }
    // non-temporal hint per access class (1 TRAM load, 2 TRAM store, 4 PCM load, 8 PCM store); FX_XLATE_NT overrides (diagnostics)
    bool streaming(int cls) const {
        static const int mask = knobInt(FX_DIAG_KNOB("FX_XLATE_NT"), -1);
        if (mask >= 0) return (mask & cls) != 0;
        return prog_.tramStreaming && (cls & 3) != 0;
    }

    static int vrow(uint32_t r) { return kRegFileBase + (int)r; }

    bool row(uint32_t r, int* v) {
        if (vrow(r) >= tmpl_.vgprs) return fail("register-file row beyond the VGPR budget of the build");
        *v = vrow(r);
        for (int p : pending_)
            if (p == *v) return fail("internal: row with a TRAM read in flight");
        return true;
    }

    // the fast stream's entrance checks (fx_xlate.hpp "taint"): a value that came from memory into register v
    void taintIfNonFinite(int v) {
        e_.vopc(VOPC_CMP_CLASS_F32, "v_cmp_class_f32_e32", vreg(v), kVClassMask);
        e_.sop2(SOP2_OR_B64, "s_or_b64", sreg64(kSTaint), sreg64(kSTaint), named(106, "vcc"));
    }
    // a row of the bounded class must stay inside [-1, 1] (NaN fails the test as well); any other must stay finite
    void taintCheckRow(int v) {
        if (!isBoundedVgpr(v)) { taintIfNonFinite(v); return; }
        e_.vop3cmp(VOP3_CMP_NLE_F32, "v_cmp_nle_f32_e64", vreg(v), true, imm32(0x3f800000u));
        e_.sop2(SOP2_OR_B64, "s_or_b64", sreg64(kSTaint), sreg64(kSTaint), named(106, "vcc"));
    }

    // PCM rows of all channels: load v23.. (of the NEXT sample when `next`) or store the output latch rows
    void pcmAccess(bool load, int sbase, bool next) {
        const int channels = (int)prog_.latchRows.size();
        e_.sop1(SOP1_MOV_B64, "s_mov_b64", named(126, "exec"), sreg64(kSValidLanes));
        if (next) {
            e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSAddr), sreg(sbase), sreg(kSSampleBytes));
            e_.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSAddr + 1), sreg(sbase + 1), imm32(0));
        } else {
            e_.sop1(SOP1_MOV_B64, "s_mov_b64", sreg64(kSAddr), sreg64(sbase));
        }
        for (int c = 0; c < channels; ++c) {
            if (load) {
                if (prog_.inRows[(size_t)c] >= 0) e_.global(GLOBAL_LOAD_DWORD, true, kVInput + c, kVInstance4, kSAddr, streaming(4));
            } else if ((prog_.stage.storeMask >> c) & 1u) {
                e_.global(GLOBAL_STORE_DWORD, false, vrow((uint32_t)prog_.latchRows[(size_t)c]), kVInstance4, kSAddr, streaming(8));
            }
            if (c + 1 < channels) {
                e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSAddr), sreg(kSAddr), sreg(kSChannelBytes));
                e_.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSAddr + 1), sreg(kSAddr + 1), imm32(0));
            }
        }
        e_.sop1(SOP1_MOV_B64, "s_mov_b64", named(126, "exec"), named(193, "-1"));
    }

    // ---- staged programs (fx_xlate.hpp StageInfo)
    // v = lane * 4 + ((sample + 1) & 3) * stride: the next buffer of the ring
    void ringStep(int v) {
        const StageInfo& G = prog_.stage;
        e_.vop2(VOP2_ADD_U32, "v_add_u32_e32", v, imm32(G.bufStride), v);
        e_.vop2(VOP2_AND_B32, "v_and_b32_e32", v, imm32(4u * (uint32_t)G.group * G.bufStride - 1u), v);
    }
    // Behind a barrier: has the stage in front of this one left its fast stream?  Its packets of the step that just ended -
    // consumed three steps from now - may then hold non-finite values, and this wavefront continues in ITS exact stream, at the
    // same point there (sync point `key` of the end of the stream: 1 = behind the group's barrier, 2 = behind the last barrier
    // of the cold entry).  The exact stream only notes where that point is.
    bool stageFlagCheck(int key) {
        const StageInfo& G = prog_.stage;
        if (G.index == 0) return true;
        if (!fast_) { syncPoint(syncIndex(key), base_ + (uint32_t)e_.bytes()); return true; }
        flagAccess(true, 2, G.flagBase + 256u * (uint32_t)(G.index - 1), 3);
        e_.waitLgkm0();
        e_.vopc(VOPC_CMP_NE_U32, "v_cmp_ne_u32_e32", imm32(0), 2);
        e_.sop2(SOP2_OR_B64, "s_or_b64", sreg64(kSTaint), sreg64(kSTaint), named(106, "vcc"));
        return leaveIfTainted((*exactReturns_)[syncIndex(key)]);
    }
    // request the rows of the NEXT sample this stage will work on (the previous stage wrote them two steps ago).  Packet s of
    // the cut behind stage c lives in buffer (s + c + 1) mod ring: what stage k sends for its sample s and what it requests for
    // its sample s + 1 (from the cut in front of it) share one buffer index, s + k + 1 - one pointer, stepped once per sample.
    void stageRequest() {
        const StageInfo& G = prog_.stage;
        std::vector<int> regs;
        for (size_t i = 0; i < G.recvRows.size(); ++i) regs.push_back(G.recvTmp + (int)i);
        packetIo(true, regs, G.recvOff - G.ptrBias);
    }
    // The rows of a packet, regs[i] <-> LDS row at kVRing + bufBase + rel + 256 i.  Two neighbouring rows go in ONE ds_read2_b32 /
    // ds_write2_b32 where both dword offsets fit its 8-bit fields (the ring at LDS address 0: StageInfo::ptrBias) - a read2's
    // destination is a register pair, even-aligned on gfx950.  packetOps: how many instructions that makes (the counted
    // s_waitcnt lgkmcnt of the protocol count instructions, not rows).
    static bool pairable(uint32_t byteOff) { return (byteOff & 3u) == 0 && byteOff / 4u + 64u <= 255u; }
    int packetOps(size_t rows, uint32_t rel, int firstReg) const {   // firstReg: of a read's consecutive destination registers, -1 for writes
        const StageInfo& G = prog_.stage;
        int ops = 0;
        for (size_t i = 0; i < rows;) {
            const uint32_t off = G.bufBase + rel + 256u * (uint32_t)i;
            const bool two = i + 1 < rows && pairable(off) && (firstReg < 0 || ((firstReg + (int)i) & 1) == 0);
            i += two ? 2 : 1;
            ++ops;
        }
        return ops;
    }
    void packetIo(bool load, const std::vector<int>& regs, uint32_t rel) {
        const StageInfo& G = prog_.stage;
        for (size_t i = 0; i < regs.size();) {
            const uint32_t off = G.bufBase + rel + 256u * (uint32_t)i;
            const bool two = i + 1 < regs.size() && pairable(off) && (!load || ((regs[i] & 1) == 0 && regs[i + 1] == regs[i] + 1));
            if (two && load) e_.dsRead2B32(regs[i], kVRing, off / 4u, off / 4u + 64u);
            else if (two) e_.dsWrite2B32(kVRing, regs[i], regs[i + 1], off / 4u, off / 4u + 64u);
            else if (load) e_.dsReadB32(regs[i], kVRing, off);
            else e_.dsWriteB32(kVRing, regs[i], off);
            i += two ? 2 : 1;
        }
    }
    // a flag row (lane * 4 + offset): behind a ring at address 0 the offset can be beyond the 16 bits of a DS instruction
    void flagAccess(bool load, int vreg_, uint32_t offset, int vtmp) {
        int addr = kVLane4;
        if (offset > 0xffffu) {
            e_.vop2(VOP2_ADD_U32, "v_add_u32_e32", vtmp, imm32(offset), kVLane4);
            addr = vtmp;
            offset = 0;
        }
        if (load) e_.dsReadB32(vreg_, addr, offset);
        else e_.dsWriteB32(addr, vreg_, offset);
    }
    // PCM input in bursts of eight samples (StageInfo::inRing).  inputBurst(first, initial): loads of samples s + first ..
    // s + first + 7 (s = the current sample, s3) into the "next" half, each only if it exists; the initial one is waited for.
    void inputBurst(int first, bool initial) {
        const int channels = (int)prog_.latchRows.size();
        const StageInfo& G = prog_.stage;
        e_.sop2(SOP2_SUB_I32, "s_sub_i32", sreg(kSTemp + 2), sreg(kSNumSamples), sreg(kSSample));   // samples from s on
        if (first) {
            e_.sop2(SOP2_SUB_I32, "s_sub_i32", sreg(kSTemp + 2), sreg(kSTemp + 2), imm32((uint32_t)first));
            e_.sop2(SOP2_LSHL_B32, "s_lshl_b32", sreg(kSTemp + 3), sreg(kSSampleBytes), imm32(3));
            e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSAddr), sreg(kSPcmIn), sreg(kSTemp + 3));
            e_.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSAddr + 1), sreg(kSPcmIn + 1), imm32(0));
        } else {
            e_.sop1(SOP1_MOV_B64, "s_mov_b64", sreg64(kSAddr), sreg64(kSPcmIn));
        }
        e_.sop1(SOP1_MOV_B64, "s_mov_b64", named(126, "exec"), sreg64(kSValidLanes));
        std::vector<Emitter::Fixup> done;
        for (int j = 0; j < kInputBurst; ++j) {
            e_.sopc(SOPC_CMP_GT_I32, "s_cmp_gt_i32", sreg(kSTemp + 2), imm32((uint32_t)j));
            done.push_back(e_.branchForward(SOPP_CBRANCH_SCC0, "s_cbranch_scc0"));
            e_.sop1(SOP1_MOV_B64, "s_mov_b64", sreg64(kSTemp + 4), sreg64(kSAddr));
            int used = 0;
            for (int c = 0; c < channels; ++c) {
                if (prog_.inRows[(size_t)c] >= 0) {
                    e_.global(GLOBAL_LOAD_DWORD, true, G.inRing + 2 * kInputBurst * used + kInputBurst + j, kVInstance4, kSTemp + 4, streaming(4));
                    ++used;
                }
                if (c + 1 < channels) {
                    e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSTemp + 4), sreg(kSTemp + 4), sreg(kSChannelBytes));
                    e_.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSTemp + 5), sreg(kSTemp + 5), imm32(0));
                }
            }
            e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSAddr), sreg(kSAddr), sreg(kSSampleBytes));
            e_.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSAddr + 1), sreg(kSAddr + 1), imm32(0));
        }
        for (const Emitter::Fixup& f : done) e_.bind(f);
        e_.sop1(SOP1_MOV_B64, "s_mov_b64", named(126, "exec"), named(193, "-1"));
        if (initial) e_.waitVmcnt(0);
    }
    // head of a sample: at the first sample of every eight, the burst requested eight samples ago becomes the current one and
    // the next is requested; then this sample's value is picked from the current eight (VGPR index mode, src0 relative)
    bool inputFromRing(int storesPerSample) {
        const int channels = (int)prog_.latchRows.size();
        const StageInfo& G = prog_.stage;
        plainMode();
        e_.sop2(SOP2_AND_B32, "s_and_b32", sreg(kSTemp), sreg(kSSample), imm32(kInputBurst - 1));
        e_.sopc(SOPC_CMP_EQ_U32, "s_cmp_eq_u32", sreg(kSTemp), imm32(0));
        // (one sample in eight: behind the loop)
        defer(e_.branchForward(SOPP_CBRANCH_SCC1, "s_cbranch_scc1"), [this, storesPerSample, channels]() {
            const StageInfo& g = prog_.stage;
            e_.cold(true);
            e_.waitVmcnt(std::min(63, kInputBurst * storesPerSample));   // younger than that burst: eight samples' PCM stores
            int n = 0;
            for (int c = 0; c < channels; ++c) {
                if (prog_.inRows[(size_t)c] < 0) continue;
                for (int j = 0; j < kInputBurst; ++j)
                    e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(g.inRing + 2 * kInputBurst * n + j), vreg(g.inRing + 2 * kInputBurst * n + kInputBurst + j));
                ++n;
            }
            if (!isLast_) inputBurst(kInputBurst, false);
            e_.sop2(SOP2_AND_B32, "s_and_b32", sreg(kSTemp), sreg(kSSample), imm32(kInputBurst - 1));  // (the burst code used the scratch register)
            e_.cold(false);
        });
        e_.setGprIdxOn(kSTemp, 1u);
        int used = 0;
        for (int c = 0; c < channels; ++c) {
            int v;
            if (prog_.inRows[(size_t)c] < 0) continue;
            if (!row((uint32_t)prog_.inRows[(size_t)c], &v)) return false;
            e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(v), vreg(G.inRing + 2 * kInputBurst * used));
            ++used;
        }
        e_.sopp(SOPP_IDX_OFF, "s_set_gpr_idx_off", 0, false);
        if (fast_)
            for (int c = 0; c < channels; ++c)
                if (prog_.inRows[(size_t)c] >= 0) taintIfNonFinite(vrow((uint32_t)prog_.inRows[(size_t)c]));
        return true;
    }

    // ---- control tracks (fx_xlate.hpp TrackEvent): scalar bookkeeping, one vector move / load per change
    void smemLoad(int dwords, int sdst, int sbase, uint32_t offset) {  // s_load_dword / x2 / x4 with a 20-bit immediate offset
        const uint32_t op = dwords == 1 ? 0u : (dwords == 2 ? 1u : 2u);
        e_.raw2(0xc0020000u | (op << 18) | ((uint32_t)sdst << 6) | ((uint32_t)sbase >> 1), offset,
                std::string(dwords == 1 ? "s_load_dword s" + std::to_string(sdst) : "s_load_dwordx" + std::to_string(dwords) + " s[" + std::to_string(sdst) + ":" +
                            std::to_string(sdst + dwords - 1) + "]") + ", s[" + std::to_string(sbase) + ":" + std::to_string(sbase + 1) + "], " +
                    (offset ? "0x" + hexOf(offset) : std::string("0x0")));
    }
    static std::string hexOf(uint32_t v) {
        char buf[16];
        std::snprintf(buf, sizeof(buf), "%x", v);
        return buf;
    }
    // cold entry: the first event of the block's list (fx_xlate.hpp TrackEvent)
    void trackInit() {
        smemLoad(2, kSTemp, 0, kKernargTracks);                       // s[62:63] = tracks buffer
        e_.waitLgkm0();
        e_.sop1(SOP1_MOV_B64, "s_mov_b64", sreg64(kSEventPtr), sreg64(kSTemp));
        smemLoad(1, kSEventNext, kSEventPtr, 0);                       // the sample of the first event (0xFFFFFFFF: none)
        e_.waitLgkm0();
    }
    // head of a sample: ONE compare whatever the number of schedules; the events due at this sample are applied behind the loop
    bool trackStep() {
        std::vector<int> rows;
        for (int r : prog_.trackRows) {
            int v;
            if (!row((uint32_t)r, &v)) return false;
            rows.push_back(v);
        }
        e_.sopc(SOPC_CMP_EQ_U32, "s_cmp_eq_u32", sreg(kSSample), sreg(kSEventNext));
        defer(e_.branchForward(SOPP_CBRANCH_SCC1, "s_cbranch_scc1"), [this, rows]() {
            e_.cold(true);
            const size_t again = e_.words();
            smemLoad(4, kSTemp + 2, kSEventPtr, 0);                    // s[64:67] = sample, slot, value offset, stride
            smemLoad(2, kSTemp, 0, kKernargTracks);
            e_.waitLgkm0();
            e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSTemp), sreg(kSTemp), sreg(kSTemp + 4));   // s[62:63] = address of the value(s)
            e_.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSTemp + 1), sreg(kSTemp + 1), imm32(0));
            std::vector<Emitter::Fixup> to(rows.size()), done;
            for (size_t t = 0; t < rows.size(); ++t) {
                e_.sopc(SOPC_CMP_EQ_U32, "s_cmp_eq_u32", sreg(kSTemp + 3), imm32((uint32_t)t));
                to[t] = e_.branchForward(SOPP_CBRANCH_SCC1, "s_cbranch_scc1");
            }
            done.push_back(e_.branchForward(SOPP_BRANCH, "s_branch"));   // (a slot this code does not know: skipped)
            for (size_t t = 0; t < rows.size(); ++t) {
                e_.bind(to[t]);
                e_.sopc(SOPC_CMP_EQ_U32, "s_cmp_eq_u32", sreg(kSTemp + 5), imm32(4));
                Emitter::Fixup perInstance = e_.branchForward(SOPP_CBRANCH_SCC0, "s_cbranch_scc0");
                smemLoad(1, kSTemp + 2, kSTemp, 0);                    // one value for every instance
                e_.waitLgkm0();
                e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(rows[t]), sreg(kSTemp + 2));
                Emitter::Fixup loaded = e_.branchForward(SOPP_BRANCH, "s_branch");
                e_.bind(perInstance);
                e_.sop1(SOP1_MOV_B64, "s_mov_b64", named(126, "exec"), sreg64(kSValidLanes));
                e_.global(GLOBAL_LOAD_DWORD, true, rows[t], kVInstance4, kSTemp);
                e_.sop1(SOP1_MOV_B64, "s_mov_b64", named(126, "exec"), named(193, "-1"));
                e_.waitVmcnt(0);
                e_.bind(loaded);
                if (fast_) taintIfNonFinite(rows[t]);
                if (t + 1 < rows.size()) done.push_back(e_.branchForward(SOPP_BRANCH, "s_branch"));
            }
            for (const Emitter::Fixup& f : done) e_.bind(f);
            e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSEventPtr), sreg(kSEventPtr), imm32(16));
            e_.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSEventPtr + 1), sreg(kSEventPtr + 1), imm32(0));
            smemLoad(1, kSEventNext, kSEventPtr, 0);
            e_.waitLgkm0();
            e_.sopc(SOPC_CMP_EQ_U32, "s_cmp_eq_u32", sreg(kSSample), sreg(kSEventNext));
            if (!e_.branchBack(SOPP_CBRANCH_SCC1, "s_cbranch_scc1", again)) deferredFailed_ = true;
            e_.cold(false);
        });
        return true;
    }

    // TRAM reads in the middle of a program are issued without waiting; the wait (and, in the fast stream, the taint
    // check of what arrived and the hand-over to the exact stream) happens here, before the first use of such a row
    bool flush(int key = 0) {
        if (pending_.empty()) return true;
        e_.waitVmcnt(0);
        syncPoint(syncIndex(key), base_ + (uint32_t)e_.bytes());
        if (fast_) {
            plainMode();
            for (int v : pending_) taintCheckRow(v);
            if (!leaveIfTainted((*exactReturns_)[syncIndex(key)])) return false;
        }
        pending_.clear();
        return true;
    }
    // sync points of a record: per half, [0] after the wait for pending TRAM reads, [1] after its call / inline LUT
    size_t syncIndex(int kind) const { return 4 * index_ + (size_t)kind; }
    bool leaveIfTainted(uint32_t target) {
        if (target == 0) return fail("internal: fast and exact streams differ in their sync points");
        e_.sopc(SOPC_CMP_LG_U64, "s_cmp_lg_u64", sreg64(kSTaint), imm32(0));
        const int64_t delta = ((int64_t)target - ((int64_t)base_ + (int64_t)e_.bytes() + 4)) / 4;
        if (delta < -32768 || delta > 32767) return fail("exact stream out of branch range of the fast stream");
        e_.sopp(SOPP_CBRANCH_SCC1, "s_cbranch_scc1", (uint32_t)delta & 0xffffu, true);
        return true;
    }
    // the record is about to read / write these register-file rows: none may still be in flight
    bool touch(const MicroOp& r, bool a, bool x, bool y, bool dst) {
        if (pending_.empty()) return true;
        const uint32_t rows[4] = {r.w[2], r.w[3], r.w[4], r.w[5]};
        const bool use[4] = {a, x, y, dst};
        for (int k = 0; k < 4; ++k)
            if (use[k])
                for (int p : pending_)
                    if (p == vrow(rows[k])) return flush();
        return true;
    }

    static int32_t x86Trunc(uint32_t bits) {  // cvttss2si: 0x80000000 for NaN and anything outside int32
        float f;
        std::memcpy(&f, &bits, 4);
        if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT32_MIN;
        return (int32_t)f;
    }

    // LOG / EXP with a per-lane operand and a uniform table (FX8010.cpp:1113-1125, linearInterpolate :283-296), from
    // the host tables of fx_model.hpp.  The segment index is a monotone step function of the fp32 operand: guess
    // g = (int)((x + 1) * 31.5), then y = slope[g] * ((double)x - x1[g]) + y1[g] with the reference's two roundings.
    // The guess is off by one only within a few ulp of a threshold, so everything segment g needs - its two fp32
    // thresholds included - is fetched in ONE round trip (LDS: four conflict-light ds_read_b64), the thresholds
    // check the guess, and the rare miss takes a second trip with the corrected index.
    // Bit-identical to the interpreter's h_lut (dense sweep in tests/test_gpu_parity.py).
    // one LOG / EXP site: what its fetch and its miss path need to know
    struct LutSite { int vA = 0; bool lds = false, guarded = false, quick = false; uint32_t window = 0, slopeOff = 0; };
    bool lut(const MicroOp& r) {
        int vA, vR;
        if (!touch(r, true, false, false, true) || !row(r.w[2], &vA) || !row(r.w[5], &vR)) return false;
        plainMode();
        const bool operandWild = r.w[2] >= prog_.wildRow.size() || prog_.wildRow[r.w[2]];
        int ldsTable = -1;
        for (size_t k = 0; k < prog_.lutTables.size(); ++k)
            if (prog_.lutTables[k] == r.w[3]) ldsTable = (int)k;
        const bool lds = ldsTable >= 0;
        if (!lds && (!segKnown_ || segOff_ != r.w[3])) {
            e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSLutSeg), sreg(kSLut), imm32(r.w[3], true));
            e_.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSLutSeg + 1), sreg(kSLut + 1), imm32(0));
            segKnown_ = true;
            segOff_ = r.w[3];
        }
        // An operand of the BOUNDED class lies in [-1, 1] while the wave runs the fast stream: the guess and the corrected
        // index are in 0..63 by construction and nothing can be out of the domain.  Everywhere else (wild operand, or the
        // exact stream, which a wave enters precisely when that invariant broke) the index is clamped and the flag derived.
        LutSite site;
        site.vA = vA;
        site.lds = lds;
        site.guarded = operandWild || !fast_;
        site.slopeOff = lutLds().tables + (uint32_t)(lds ? ldsTable : 0) * lutLds().tableBytes;
        site.window = (lds && fast_) ? lutGuessWindowHi() : 0;
        site.quick = site.window != 0 && lutGuess().available(e_);
        if (fast_ && operandWild && site.quick) {
            // A wild operand is nearly always inside the table too (the PCM input, a wrap-around result): one compare sends the
            // wave to the guarded form - behind the loop, with the out-of-domain flag and the taint check of the result - only
            // when some lane holds |x| > 1 or a NaN; otherwise it runs the bounded form, whose result is finite.
            e_.vop3cmpG(VOP3_CMP_NLE_F32, "v_cmp_nle_f32_e64", named(106, "vcc"), vreg(vA), true, imm32(0x3f800000u));
            Emitter::Fixup outside = e_.branchForward(SOPP_CBRANCH_VCCNZ, "s_cbranch_vccnz");
            site.guarded = false;
            lutBody(site);
            e_.vop1(VOP1_CVT_F32_F64, "v_cvt_f32_f64_e32", vreg(vR), vreg64(12));
            LutSite slow = site;
            slow.guarded = true;
            slow.quick = false;
            slow.window = 0;
            const uint32_t exactSync = (*exactReturns_)[syncIndex(1)];
            defer(outside, [this, slow, vR, exactSync]() {
                e_.cold(true);
                lutBody(slow);
                e_.vop1(VOP1_CVT_F32_F64, "v_cvt_f32_f64_e32", vreg(vR), vreg64(12));
                taintIfNonFinite(vR);
                if (!leaveIfTainted(exactSync)) deferredFailed_ = true;
                e_.cold(false);
            });
            syncPoint(syncIndex(1), base_ + (uint32_t)e_.bytes());
            return true;
        }
        if (site.guarded) site.window = 0, site.quick = false;
        lutBody(site);
        e_.vop1(VOP1_CVT_F32_F64, "v_cvt_f32_f64_e32", vreg(vR), vreg64(12));
        syncPoint(syncIndex(1), base_ + (uint32_t)e_.bytes());
        if (operandWild && fast_) {  // a wild operand can be Inf / NaN, and then so is the result
            taintIfNonFinite(vR);
            if (!leaveIfTainted((*exactReturns_)[syncIndex(1)])) return false;
        }
        return true;
    }

    // guess, fetch, check (second trip behind the loop, or inline for tables in global memory), out-of-domain flag of the
    // guarded form: leaves slope * (x - x1) + y1 in v[12:13]
    void lutBody(const LutSite& site) {
        const int vA = site.vA;
        const bool lds = site.lds, guarded = site.guarded, quick = site.quick;
        const uint32_t window = site.window;
        Src zero = imm32(0), top = imm32(63), vcc = named(106, "vcc");
        // The guess.  Operand inside [-1, 1], tables in LDS and the four constants in VGPRs: three instructions of the double-rate
        // class give 8 * floor((x + 1) * 31.5) directly - t = fma(x, 252, 251.5) = 8 * (x + 1) * 31.5 - 0.5; adding
        // 1.5 * 2^23 rounds t to the nearest integer q, which lands in the low mantissa bits (q = floor(8 * ...) unless the
        // fraction is within 2^-15 of a whole number; at x = +-1, the only floats where the tie can matter, round-to-even
        // picks 504 and 0); q & 0x1f8 is the byte offset of the segment.  Otherwise (x + 1) * 31.5 truncated by v_cvt.
        if (quick) {
            Src half = vreg(e_.pooled(lutGuess().bias));
            e_.vop3(VOP3_FMA_F32, "v_fma_f32", vreg(6), vreg(vA), vreg(e_.pooled(lutGuess().scale)), &half);
            e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", 6, vreg(e_.pooled(lutGuess().magic)), 6);
            e_.vop2(VOP2_AND_B32, "v_and_b32_e32", 7, vreg(e_.pooled(lutGuess().mask)), 6);
        } else {
            e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", 6, imm32(0x3f800000u), vA);
            e_.vop2(VOP2_MUL_F32, "v_mul_f32_e32", 6, imm32(0x41fc0000u), 6);              // * 31.5
            e_.vop1(VOP1_CVT_I32_F32, "v_cvt_i32_f32_e32", vreg(6), vreg(6));                // saturating, NaN -> 0
        }
        if (guarded) e_.vop3(VOP3_MED3_I32, "v_med3_i32", vreg(6), vreg(6), zero, &top);
        // Where the operand is known to be in the table (not guarded) the guess is checked without its thresholds: d = x - x1[g]
        // must lie in [0, W), W a constant of the grid (fx_frontend.cpp lutGuessWindowHi) - one unsigned compare of d's high
        // word, three LDS reads instead of four.  A miss (one lane in ~10^5 within reach of a threshold) reads the thresholds
        // after all and corrects the index as the guarded form does.
        static const int prio = knobInt(FX_DIAG_KNOB("FX_XLATE_LUTPRIO"), 0);   // diagnostics
        // diagnostics, WRONG RESULTS (timing only): 1 = no branch to the miss path, 2 = no LDS reads and no wait, 4 = reads but no wait
        static const int probe = knobInt(FX_DIAG_KNOB("FX_XLATE_LUTPROBE_WRONG_RESULTS"), 0);
        if (prio > 0 && lds) e_.sopp(0x0fu, "s_setprio", (uint32_t)prio & 3u, true);
        if (!(lds && (probe & 2))) lutFetch(site, window == 0, quick);
        e_.vop1(VOP1_CVT_F64_F32, "v_cvt_f64_f32_e32", vreg64(12), vreg(vA));
        {
            // diagnostics: how much independent work fits into the LDS round trip for nothing?  FX_XLATE_LUTPAD=n pads every
            // LOG / EXP with n plain (double-rate class) instructions on a spare register, FX_XLATE_LUTPAD_SLOW=n with n
            // conversions (the 4-clock class) between the reads and their wait (tools/lut_pad_probe.sh)
            static const int pad = knobInt(FX_DIAG_KNOB("FX_XLATE_LUTPAD"), 0);
            static const int padSlow = knobInt(FX_DIAG_KNOB("FX_XLATE_LUTPAD_SLOW"), 0);
            static const int padAfter = knobInt(FX_DIAG_KNOB("FX_XLATE_LUTPAD_AFTER"), 0);
            if (lds && !padAfter) {
                for (int k = 0; k < pad; ++k) e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", 31, vreg(31), 31);
                for (int k = 0; k < padSlow; ++k) e_.vop1(VOP1_CVT_F32_U32, "v_cvt_f32_u32_e32", vreg(31), vreg(31));
            }
            if (lds && (probe & 6)) {} else if (lds) e_.waitLgkm0(); else e_.waitVmcnt(0);
            if (lds && padAfter) {   // the same instructions BEHIND the wait: what they cost when nothing hides them
                for (int k = 0; k < pad; ++k) e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", 31, vreg(31), 31);
                for (int k = 0; k < padSlow; ++k) e_.vop1(VOP1_CVT_F32_U32, "v_cvt_f32_u32_e32", vreg(31), vreg(31));
            }
        }
        if (prio > 0 && lds) e_.sopp(0x0fu, "s_setprio", 0, true);
        // the segment arithmetic goes ahead on the guess while the scalar unit makes up its mind (the compares' results
        // reach it a pipeline later: checking first would stall the wave twice per LOG/EXP); a miss redoes it
        if (window) {
            e_.vop3(VOP3_ADD_F64, "v_add_f64", vreg64(12), vreg64(12), vreg64(10), nullptr, 2);         // d = x - x1
            e_.vopc(VOPC_CMP_LE_U32, "v_cmp_le_u32_e32", imm32(window, true), 13);                     // lanes outside [0, W)
            lutSegmentMath(1);
        } else {
            e_.vopc(VOPC_CMP_GE_F32, "v_cmp_ge_f32_e32", vreg(vA), 9);                    // x >= xthr[g+1]: one up
            e_.vop3cmpTo(VOP3_CMP_LT_F32, "v_cmp_lt_f32_e64", kSTemp, vreg(vA), vreg(8));  // x <  xthr[g]  : one down
            lutSegmentMath(0);
            e_.sop2(SOP2_OR_B64, "s_or_b64", vcc, vcc, sreg64(kSTemp));
        }
        if (lds && (probe & 1)) {
        } else if (lds) {
            // the miss path lives behind the loop (emitDeferred): the hit path falls through its branch
            defer(e_.branchForward(SOPP_CBRANCH_VCCNZ, "s_cbranch_vccnz"), [this, site]() { lutMiss(site); });
        } else {
            Emitter::Fixup hit = e_.branchForward(SOPP_CBRANCH_VCCZ, "s_cbranch_vccz");
            lutMiss(site);
            e_.bind(hit);
        }
        if (guarded) {
            // the index can leave 0..63 (x outside the table, or NaN): out-of-domain flag, as h_lut sets it.  The two
            // bounds are constants of the table grid (fx_model.hpp xdom): !(lo <= x) or !(hi > x)
            float dom[2];
            lutDomainBounds(dom);
            uint32_t lo, hi;
            std::memcpy(&lo, &dom[0], 4);
            std::memcpy(&hi, &dom[1], 4);
            e_.vopc(VOPC_CMP_NLE_F32, "v_cmp_nle_f32_e32", imm32(lo, true), vA);
            e_.sop1(SOP1_MOV_B64, "s_mov_b64", sreg64(kSTemp), vcc);
            e_.vopc(VOPC_CMP_NGT_F32, "v_cmp_ngt_f32_e32", imm32(hi, true), vA);
            e_.sop2(SOP2_OR_B64, "s_or_b64", vcc, vcc, sreg64(kSTemp));
            Src flag = imm32(16);
            e_.vop3(VOP3_CNDMASK, "v_cndmask_b32_e64", vreg(7), zero, flag, &vcc);
            e_.vop2(VOP2_OR_B32, "v_or_b32_e32", kVOod, vreg(kVOod), 7);
        }
    }

    // segment v6 (index) or, first fetch of the quick form, v7 (byte offset): x1 -> v[10:11], slope -> v[2:3], y1 -> v[4:5],
    // thresholds -> v[8:9]
    void lutFetch(const LutSite& s, bool withThresholds, bool offsetReady) {
        if (s.lds) {
            const LutLdsLayout& L = lutLds();
            if (!offsetReady) e_.vop2(VOP2_LSHLREV_B32, "v_lshlrev_b32_e32", 7, imm32((uint32_t)L.shift), 6);
            if (withThresholds) e_.dsRead(DS_READ_B64, "ds_read_b64", 2, 8, 7, L.thr);     // xthr[g], xthr[g+1]
            e_.dsRead(DS_READ_B64, "ds_read_b64", 2, 10, 7, L.x1);                         // x1[g]
            if (L.wide) {
                e_.dsRead(DS_READ_B128, "ds_read_b128", 4, 2, 7, s.slopeOff);              // {slope, y1} of segment g: one cell
            } else {
                e_.dsRead(DS_READ_B64, "ds_read_b64", 2, 2, 7, s.slopeOff);
                e_.dsRead(DS_READ_B64, "ds_read_b64", 2, 4, 7, s.slopeOff + 512);
            }
        } else {
            if (withThresholds) {
                e_.vop2(VOP2_LSHLREV_B32, "v_lshlrev_b32_e32", 7, imm32(2), 6);
                e_.globalLoadWide(GLOBAL_LOAD_DWORDX2, 2, 8, 7, kSLutXthr);                // xthr[g], xthr[g+1]
            }
            e_.vop2(VOP2_LSHLREV_B32, "v_lshlrev_b32_e32", 7, imm32(3), 6);
            e_.globalLoadWide(GLOBAL_LOAD_DWORDX2, 2, 10, 7, kSLutX1);                     // x1[g]
            e_.vop2(VOP2_LSHLREV_B32, "v_lshlrev_b32_e32", 7, imm32(4), 6);
            e_.globalLoadWide(GLOBAL_LOAD_DWORDX4, 4, 2, 7, kSLutSeg);                     // slope, y1
        }
    }
    void lutSegmentMath(int from) {
        if (from <= 0) e_.vop3(VOP3_ADD_F64, "v_add_f64", vreg64(12), vreg64(12), vreg64(10), nullptr, 2);   // x - x1
        e_.vop3(VOP3_MUL_F64, "v_mul_f64", vreg64(12), vreg64(2), vreg64(12), nullptr);
        e_.vop3(VOP3_ADD_F64, "v_add_f64", vreg64(12), vreg64(12), vreg64(4), nullptr);
    }
    // the guess was off by one in some lane: thresholds (read now if the window test stood in for them), corrected index,
    // second trip.  Leaves the segment arithmetic's result in v[12:13] like the hit path.
    void lutMiss(const LutSite& s) {
        const Src zero = imm32(0), top = imm32(63);
        e_.cold(true);
        if (s.window) {
            e_.dsRead(DS_READ_B64, "ds_read_b64", 2, 8, 7, lutLds().thr);
            if (s.quick) e_.vop2(VOP2_LSHRREV_B32, "v_lshrrev_b32_e32", 6, imm32((uint32_t)lutLds().shift), 7);
            e_.waitLgkm0();
            e_.vop3cmpTo(VOP3_CMP_LT_F32, "v_cmp_lt_f32_e64", kSTemp, vreg(s.vA), vreg(8));
        }
        e_.vopc(VOPC_CMP_GE_F32, "v_cmp_ge_f32_e32", vreg(s.vA), 9);                      // (VCC again: the carry of the correction)
        e_.sopp(SOPP_NOP, "s_nop", 1, true);
        e_.addCarry(6);
        e_.subBorrow(6, kSTemp, kSTemp + 2);
        if (s.guarded) e_.vop3(VOP3_MED3_I32, "v_med3_i32", vreg(6), vreg(6), zero, &top);
        lutFetch(s, false, false);
        e_.vop1(VOP1_CVT_F64_F32, "v_cvt_f64_f32_e32", vreg64(12), vreg(s.vA));
        if (s.lds) e_.waitLgkm0(); else e_.waitVmcnt(0);
        lutSegmentMath(0);
        e_.cold(false);
    }
    // code that almost never runs, kept out of the loop body: entered by a forward branch, returns by a backward one
    struct Deferred { Emitter::Fixup entry; size_t resume = 0; std::function<void()> body; };
    // the hot path continues at the current position; `body` runs behind the loop when the branch at `entry` is taken
    void defer(const Emitter::Fixup& entry, std::function<void()> body) {
        Deferred d;
        d.entry = entry;
        d.resume = e_.words();
        d.body = std::move(body);
        deferred_.push_back(std::move(d));
    }
    bool emitDeferred() {
        for (size_t k = 0; k < deferred_.size(); ++k) {  // (a body may defer paths of its own)
            const Deferred d = deferred_[k];
            e_.bind(d.entry);
            d.body();
            if (deferredFailed_) return false;
            if (!e_.branchBack(SOPP_BRANCH, "s_branch", d.resume)) return fail("translated loop too long for a branch");
        }
        deferred_.clear();
        return true;
    }

    // IDELAY / XDELAY with uniform cursors (fx_interp_handlers.inc TRAM_READ / TRAM_WRITE are the per-lane versions;
    // FX8010.cpp:909-967).  Slot arithmetic, bounds and cursor update are scalar; the lanes only move data.
    // Every path issues exactly one vector memory operation or waits for all of them (s_waitcnt vmcnt(0) on the
    // out-of-range path), so that "operations issued after this one" can be counted when the code is generated.
    static int tramOf(uint32_t slot) { return (slot == AS_TRAM_IR || slot == AS_TRAM_IW) ? 0 : 1; }
    int32_t tramOffset(const MicroOp& r, int32_t size) const {
        int32_t p = x86Trunc(r.w[4]);
        p = p > size - 1 ? size - 1 : p;
        return p < 0 ? 0 : p;
    }
    void advanceCursor(int cursor, int t) {  // cursor = (cursor + 1) % size
        e_.sop2(SOP2_ADD_I32, "s_add_i32", sreg(cursor), sreg(cursor), imm32(1));
        e_.sopc(SOPC_CMP_GE_I32, "s_cmp_ge_i32", sreg(cursor), sreg(kSTramSize[t]));
        e_.sop2(SOP2_CSELECT_B32, "s_cselect_b32", sreg(cursor), imm32(0), sreg(cursor));
    }
    void slotAddress(const Src& pos, int t) {
        e_.sop2(SOP2_LSHL_B32, "s_lshl_b32", sreg(kSPos), pos, imm32(8));
        e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSAddr), sreg(kSTramBase[t]), sreg(kSPos));
        e_.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSAddr + 1), sreg(kSTramBase[t] + 1), imm32(0));
    }
    // deferred: a read in the middle of the program (its row is waited for at its first use); otherwise a leading read,
    // whose wait is the head of the sample loop
    // opt-in DANE model: slot = (counter + position) mod size, counter in s80 (iTRAM) / s82 (xTRAM); `ahead`: the read belongs
    // to the NEXT sample, whose counter is one lower
    void daneSlot(const MicroOp& r, int t, int32_t size, bool ahead) {
        int32_t q = danePosition(r.w[4], (r.w[6] & 32u) != 0, size);
        if (ahead) q = (q - 1 + size) % size;
        const int counter = kSCursor + 2 * t;
        e_.sop2(SOP2_ADD_I32, "s_add_i32", sreg(kSPos), sreg(counter), imm32((uint32_t)q));
        e_.sopc(SOPC_CMP_GE_I32, "s_cmp_ge_i32", sreg(kSPos), sreg(kSTramSize[t]));
        Emitter::Fixup inRing = e_.branchForward(SOPP_CBRANCH_SCC0, "s_cbranch_scc0");
        e_.sop2(SOP2_SUB_I32, "s_sub_i32", sreg(kSPos), sreg(kSPos), sreg(kSTramSize[t]));
        e_.bind(inRing);
    }
    // end of a sample period in the DANE model: both address counters step down
    void daneStep() {
        for (int t = 0; t < 2; ++t) {
            const int32_t size = t == 0 ? prog_.iSize : prog_.xSize;
            if (size < 1) continue;
            const int counter = kSCursor + 2 * t;
            e_.sop2(SOP2_SUB_I32, "s_sub_i32", sreg(counter), sreg(counter), imm32(1));
            e_.sopc(SOPC_CMP_LT_I32, "s_cmp_lt_i32", sreg(counter), imm32(0));
            Emitter::Fixup fine = e_.branchForward(SOPP_CBRANCH_SCC0, "s_cbranch_scc0");
            e_.sop2(SOP2_ADD_I32, "s_add_i32", sreg(counter), sreg(counter), sreg(kSTramSize[t]));
            e_.bind(fine);
        }
    }

    // ... and where a tap's position is a per-instance value in whole samples the taps are the interpreter's handlers (called as
    // subroutines, their DANE path: fx_interp_handlers.inc) and the counters are the lanes' own, v16 / v18: v <= 0 ? size - 1 : v - 1
    void daneStepPerLane() {
        for (int t = 0; t < 2; ++t) {
            const int32_t size = t == 0 ? prog_.iSize : prog_.xSize;
            if (size < 1) continue;
            const int counter = kVCursor + 2 * t;
            e_.vopc(VOPC_CMP_GT_I32, "v_cmp_gt_i32_e32", imm32(1), counter);
            e_.vop2(VOP2_ADD_U32, "v_add_u32_e32", 5, imm32(0xffffffffu), counter);
            e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(6), imm32((uint32_t)(size - 1)));
            e_.sopp(SOPP_NOP, "s_nop", 0, true);
            e_.vop2(VOP2_CNDMASK, "v_cndmask_b32_e32", counter, vreg(5), 6, ", vcc");
        }
    }
    // the low 11 bits of a uniform DANE address (FX_OPT_TRAM_INTERP: the weight of the next sample)
    static int32_t daneFraction(uint32_t bits) {
        float f;
        std::memcpy(&f, &bits, 4);
        const float scaled = f * 2147483648.0f;
        uint32_t sb;
        std::memcpy(&sb, &scaled, 4);
        return x86Trunc(sb) & 0x7ff;
    }
    // Opt-in DANE taps that are not one scalar slot: the position is a per-instance DANE address (FX_OPT_TRAM_ADDR_SHIFT: a
    // fixed-point fraction, |position| < 2^20 samples), and / or the read interpolates between two slots (FX_OPT_TRAM_INTERP).
    // Per lane: a = cvttss2si(value * 2^31); p = a >> 11; slot = (counter + p) mod size - the modulo in fp32, where every
    // quantity is an integer below 2^23 and therefore exact: q = floor(x / size) by the reciprocal is off by at most one, two
    // selects repair it; the lanes then gather / scatter with their own byte offsets.  Reads wait for their data at once.
    bool daneGather(const MicroOp& r, int t, int32_t size, bool isRead, int vData) {
        const bool perLane = !(r.w[6] & 4u), interp = isRead && (r.w[6] & 64u) != 0;
        if (perLane && !(r.w[6] & 32u)) return fail("DANE tap with a per-instance position in whole samples (HIP C++ kernel)");
        const Src vcc = named(106, "vcc");
        const int counter = kSCursor + 2 * t;
        if (isRead && !flush()) return false;   // (the wait below drains everything: earlier reads are checked first)
        // v6 = the DANE address a
        if (perLane) {
            int vY;
            if (!row(r.w[4], &vY)) return false;
            e_.vop2(VOP2_MUL_F32, "v_mul_f32_e32", 7, imm32(0x4f000000u), vY);                     // value * 2^31
            e_.vop1(VOP1_CVT_I32_F32, "v_cvt_i32_f32_e32", vreg(6), vreg(7));
            e_.vopc(VOPC_CMP_NGT_F32, "v_cmp_ngt_f32_e32", imm32(0x4f000000u), 7);                 // cvttss2si: NaN and >= 2^31 give 0x80000000 (v_cvt saturates upwards)
            e_.sopp(SOPP_NOP, "s_nop", 1, true);
            e_.vop2(VOP2_CNDMASK, "v_cndmask_b32_e32", 6, vreg(6), 28, ", vcc");                    // v28 = 0x80000000
        } else {
            float f;
            std::memcpy(&f, &r.w[4], 4);
            const float scaled = f * 2147483648.0f;
            uint32_t sb;
            std::memcpy(&sb, &scaled, 4);
            e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(6), imm32((uint32_t)x86Trunc(sb)));
        }
        // slot = (counter + (a >> 11)) mod size, as a float in v8; v10 = size
        e_.vop2(VOP2_ASHRREV_I32, "v_ashrrev_i32_e32", 7, imm32(11), 6);
        e_.vop2(VOP2_ADD_U32, "v_add_u32_e32", 7, sreg(counter), 7);
        e_.vop1(VOP1_CVT_F32_I32, "v_cvt_f32_i32_e32", vreg(8), vreg(7));
        const float sizef = (float)size, rcp = 1.0f / (float)size;
        uint32_t sizeBits, rcpBits;
        std::memcpy(&sizeBits, &sizef, 4);
        std::memcpy(&rcpBits, &rcp, 4);
        e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(10), imm32(sizeBits));
        e_.vop2(VOP2_MUL_F32, "v_mul_f32_e32", 9, imm32(rcpBits), 8);
        e_.vop1(VOP1_FLOOR_F32, "v_floor_f32_e32", vreg(9), vreg(9));
        {
            Src addend = vreg(8);
            e_.vop3(VOP3_FMA_F32, "v_fma_f32", vreg(8), vreg(9), vreg(10), &addend, 1);           // x - q * size
        }
        auto wrap = [&](int v) {   // v in [-size, 2 size) -> [0, size)
            e_.vopc(VOPC_CMP_GT_F32, "v_cmp_gt_f32_e32", imm32(0), v);
            e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", 11, vreg(10), v);
            e_.sopp(SOPP_NOP, "s_nop", 0, true);
            e_.vop2(VOP2_CNDMASK, "v_cndmask_b32_e32", v, vreg(v), 11, ", vcc");
            e_.vopc(VOPC_CMP_LE_F32, "v_cmp_le_f32_e32", vreg(10), v);
            e_.vop2(VOP2_SUB_F32, "v_sub_f32_e32", 11, vreg(v), 10);
            e_.sopp(SOPP_NOP, "s_nop", 0, true);
            e_.vop2(VOP2_CNDMASK, "v_cndmask_b32_e32", v, vreg(v), 11, ", vcc");
        };
        wrap(8);
        auto address = [&](int vSlotF, int vAddr) {   // byte offset of the lane's slot: slot * 256 + lane * 4
            e_.vop1(VOP1_CVT_I32_F32, "v_cvt_i32_f32_e32", vreg(vAddr), vreg(vSlotF));
            e_.vop2(VOP2_LSHLREV_B32, "v_lshlrev_b32_e32", vAddr, imm32(8), vAddr);
            e_.vop2(VOP2_ADD_U32, "v_add_u32_e32", vAddr, vreg(vAddr), kVLane4);
        };
        // the ring is allocated whole (slots >= size: fx_batch ensureTram) - checked once per tap, scalar
        e_.sopc(SOPC_CMP_GE_I32, "s_cmp_ge_i32", sreg(kSTramSlots[t]), imm32((uint32_t)size));
        Emitter::Fixup outside = e_.branchForward(SOPP_CBRANCH_SCC0, "s_cbranch_scc0");
        address(8, 7);
        if (!isRead) {
            e_.global(GLOBAL_STORE_DWORD, false, vData, 7, kSTramBase[t], streaming(2));
            defer(outside, [this]() {
                e_.waitVmcnt(0);
                e_.sop2(SOP2_OR_B32, "s_or_b32", sreg(kSOod), sreg(kSOod), imm32(2));
            });
            return true;
        }
        const int vR = vData;
        e_.global(GLOBAL_LOAD_DWORD, true, vR, 7, kSTramBase[t], streaming(1));
        if (interp) {
            e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", 12, imm32(0x3f800000u), 8);                      // the next slot
            e_.vopc(VOPC_CMP_LE_F32, "v_cmp_le_f32_e32", vreg(10), 12);
            e_.vop2(VOP2_SUB_F32, "v_sub_f32_e32", 11, vreg(12), 10);
            e_.sopp(SOPP_NOP, "s_nop", 0, true);
            e_.vop2(VOP2_CNDMASK, "v_cndmask_b32_e32", 12, vreg(12), 11, ", vcc");
            address(12, 7);
            e_.global(GLOBAL_LOAD_DWORD, true, 4, 7, kSTramBase[t], streaming(1));
        }
        e_.waitVmcnt(0);
        defer(outside, [this, vR, interp]() {
            e_.waitVmcnt(0);
            e_.cold(true);
            e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(vR), imm32(0));
            if (interp) e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(4), imm32(0));
            e_.cold(false);
        });
        if (interp) {
            // x0 + f * (x1 - x0), f = (a & 0x7ff) / 2048; x1 - x0 as x1 + (-1.0 * x0); lanes with f == 0 keep x0 itself
            e_.vop2(VOP2_AND_B32, "v_and_b32_e32", 13, imm32(0x7ffu), 6);
            e_.vop1(VOP1_CVT_F32_I32, "v_cvt_f32_i32_e32", vreg(9), vreg(13));
            e_.vop2(VOP2_MUL_F32, "v_mul_f32_e32", 9, imm32(0x3a000000u), 9);                       // * 2^-11
            e_.vop2(VOP2_MUL_F32, "v_mul_f32_e32", 3, imm32(0xbf800000u), vR);
            e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", 3, vreg(4), 3);                                   // x1 first
            e_.vop2(VOP2_MUL_F32, "v_mul_f32_e32", 3, vreg(9), 3);                                   // f first
            e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", 3, vreg(vR), 3);                                  // x0 first
            e_.vopc(VOPC_CMP_EQ_U32_, "v_cmp_eq_u32_e32", imm32(0), 13);
            e_.sopp(SOPP_NOP, "s_nop", 1, true);
            e_.vop2(VOP2_CNDMASK, "v_cndmask_b32_e32", vR, vreg(3), vR, ", vcc");
        }
        // what arrived (and what the interpolation made of it) is checked like any value from memory; both streams define the
        // sync point
        syncPoint(syncIndex(1), base_ + (uint32_t)e_.bytes());
        if (fast_) {
            taintCheckRow(vR);
            if (!leaveIfTainted((*exactReturns_)[syncIndex(1)])) return false;
        }
        (void)vcc;
        return true;
    }

    bool tramRead(const MicroOp& r, uint32_t slot, bool deferred, bool ahead = false) {
        const int t = tramOf(slot);
        const int cursor = kSCursor + 2 * t + 1;
        const int32_t size = t == 0 ? prog_.iSize : prog_.xSize;
        if (deferred && !touch(r, false, false, false, true)) return false;
        plainMode();
        int vR = 0;
        if (deferred ? !row(r.w[5], &vR) : false) return false;
        if (!deferred) vR = vrow(r.w[5]);
        if (vR >= tmpl_.vgprs) return fail("register-file row beyond the VGPR budget of the build");
        if (size < 1) {  // the reference would divide by zero: flagged, the read yields 0
            e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(vR), imm32(0));
            e_.sop2(SOP2_OR_B32, "s_or_b32", sreg(kSOod), sreg(kSOod), imm32(4));
            return true;
        }
        if (prog_.tramDane && (!(r.w[6] & 4u) || ((r.w[6] & 64u) && daneFraction(r.w[4]) != 0))) {
            // a tap whose position is a per-instance value (a modulated delay), or an interpolated read between two slots
            if (!deferred) return fail("internal: a leading read must be a plain tap");
            return daneGather(r, t, size, true, vR);
        }
        const int32_t p = prog_.tramDane ? 1 : tramOffset(r, size);
        const Src pos = p == 0 ? sreg(cursor) : sreg(kSPos);
        if (prog_.tramDane) {
            daneSlot(r, t, size, ahead);
        } else if (p != 0) {
            e_.sop2(SOP2_SUB_I32, "s_sub_i32", sreg(kSPos), sreg(cursor), imm32((uint32_t)p));
            e_.sopc(SOPC_CMP_LT_I32, "s_cmp_lt_i32", sreg(kSPos), imm32(0));
            Emitter::Fixup inRange = e_.branchForward(SOPP_CBRANCH_SCC0, "s_cbranch_scc0");
            e_.sop2(SOP2_ADD_I32, "s_add_i32", sreg(kSPos), sreg(kSPos), sreg(kSTramSize[t]));  // C's % keeps it negative: flagged
            e_.sop2(SOP2_OR_B32, "s_or_b32", sreg(kSOod), sreg(kSOod), imm32(1));
            e_.bind(inRange);
        }
        e_.sopc(SOPC_CMP_LT_I32, "s_cmp_lt_i32", pos, sreg(kSTramSlots[t]));
        Emitter::Fixup outside = e_.branchForward(SOPP_CBRANCH_SCC0, "s_cbranch_scc0");
        slotAddress(pos, t);
        e_.global(GLOBAL_LOAD_DWORD, true, vR, kVLane4, kSAddr, streaming(1));
        defer(outside, [this, vR]() {  // slot beyond the allocation (behind the loop): no load, everything drained instead
            e_.waitVmcnt(0);
            e_.cold(true);
            e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(vR), imm32(0));
            e_.cold(false);
        });
        if (deferred) pending_.push_back(vR);
        if (!prog_.tramDane) advanceCursor(cursor, t);
        return true;
    }
    bool tramWrite(const MicroOp& r, uint32_t slot) {
        const int t = tramOf(slot);
        const int cursor = kSCursor + 2 * t;
        const int32_t size = t == 0 ? prog_.iSize : prog_.xSize;
        if (!touch(r, !(r.w[6] & 1u), false, false, false)) return false;
        plainMode();
        if (size < 1) {
            e_.sop2(SOP2_OR_B32, "s_or_b32", sreg(kSOod), sreg(kSOod), imm32(4));
            return true;
        }
        const int32_t p = prog_.tramDane ? 1 : tramOffset(r, size);
        const Src pos = p == 0 ? sreg(cursor) : sreg(kSPos);
        int vA = 2;
        if (r.w[6] & 1u) e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(2), value(r.w[2]));
        else if (!row(r.w[2], &vA)) return false;
        if (prog_.tramDane && !(r.w[6] & 4u)) return daneGather(r, t, size, false, vA);
        if (prog_.tramDane) daneSlot(r, t, size, false);
        else if (p != 0) e_.sop2(SOP2_ADD_I32, "s_add_i32", sreg(kSPos), sreg(cursor), imm32((uint32_t)p));
        e_.sop2(SOP2_MIN_I32, "s_min_i32", sreg(kSAddr), sreg(kSTramSlots[t]), imm32(t == 0 ? 8192u : 1048576u, true));
        e_.sopc(SOPC_CMP_LT_I32, "s_cmp_lt_i32", pos, sreg(kSAddr));
        Emitter::Fixup outside = e_.branchForward(SOPP_CBRANCH_SCC0, "s_cbranch_scc0");
        slotAddress(pos, t);
        e_.global(GLOBAL_STORE_DWORD, false, vA, kVLane4, kSAddr, streaming(2));
        defer(outside, [this]() {
            e_.waitVmcnt(0);
            e_.sop2(SOP2_OR_B32, "s_or_b32", sreg(kSOod), sreg(kSOod), imm32(2));
        });
        if (!prog_.tramDane) advanceCursor(cursor, t);
        return true;
    }
    bool tram(const MicroOp& r, uint32_t slot) {
        return (slot == AS_TRAM_IR || slot == AS_TRAM_XR) ? tramRead(r, slot, true) : tramWrite(r, slot);
    }

    // operand word -> source: a register-file row (VGPR) or the uniform's bit pattern
    bool operand(uint32_t word, bool uniform, Src* s) {
        if (uniform) { *s = value(word); return true; }
        int v;
        if (!row(word, &v)) return false;
        *s = vreg(v);
        return true;
    }

    bool isBoundedVgpr(int v) const {
        const int r = v - kRegFileBase;
        return v >= kRegFileBase && r >= 0 && r < (int)prog_.wildRow.size() && !prog_.wildRow[r];
    }
    // upper bound of |operand| that the fast stream may rely on: |c| of a uniform, 1 for a row of the bounded
    // class (the taint checks keep that invariant), +inf for any other row
    double bound(uint32_t word, bool uniform) const {
        if (uniform) {
            float f;
            std::memcpy(&f, &word, 4);
            return f == f ? std::fabs((double)f) : HUGE_VAL;
        }
        return (word < prog_.wildRow.size() && !prog_.wildRow[word]) ? 1.0 : HUGE_VAL;
    }
    // Can the saturation of this instruction's result be dropped in the fast stream?  Every rounding step is
    // monotone and the bounds are floats, so |exact bound| <= 1 carries through the fp32 (fp64 for INTERP) roundings.
    bool resultWithinUnit(uint32_t family, uint32_t kind, const MicroOp& r) const {
        if (!fast_ || kind == 7) return false;
        const bool uA = kind & 1, uX = kind & 2, uY = kind & 4;
        if (family <= 1) {  // A +- X*Y: a folded product sits in the X word
            const double bp = (uX && uY) ? bound(r.w[3], true) : bound(r.w[3], uX) * bound(r.w[4], uY);
            return bound(r.w[2], uA) + bp <= 1.0;
        }
        if (family == 2) {  // (A + X) + Y: a folded A + X sits in the A word
            const double bs = (uA && uX) ? bound(r.w[2], true) : bound(r.w[2], uA) + bound(r.w[3], uX);
            return bs + bound(r.w[4], uY) <= 1.0;
        }
        // INTERP: a convex combination of A and Y when X is a constant in [2^-20, 1]
        if (!uX) return false;
        double omx;
        const uint64_t bitsOmx = (uint64_t)r.w[6] | ((uint64_t)r.w[7] << 32);
        std::memcpy(&omx, &bitsOmx, 8);
        if (!(omx >= 0.0 && omx <= 1.0 - 9.5367431640625e-07)) return false;
        if (bound(r.w[2], uA) > 1.0) return false;
        if (uY) {  // folded product X*Y in the X word: |p| <= X = 1 - omx must hold
            return bound(r.w[3], true) <= 1.0 - omx;
        }
        return bound(r.w[4], false) <= 1.0;
    }

    // a uniform operand's bit pattern; a NaN or Inf among them rules the fast stream out
    Src value(uint32_t bits) {
        if ((bits & 0x7f800000u) == 0x7f800000u) nonFinite_ = true;
        return imm32(bits);
    }

    // plain VALU code follows: leave the VGPR index mode a handler may have left on
    void plainMode() {
        if (!indexModeUnknown_) return;
        e_.sopp(SOPP_IDX_OFF, "s_set_gpr_idx_off", 0, false);
        e_.sopp(SOPP_NOP, "s_nop", 3, true);
        indexModeUnknown_ = false;
    }

    // v2 -> saturate (NaN passes, FX8010.cpp:275-279) -> row R
    void satStore(int vR) {
        Src m1 = imm32(0xbf800000u), p1 = imm32(0x3f800000u);
        if (fast_) {  // no NaN can be here (taint discipline): the median is the saturation
            e_.vop3(VOP3_MED3_F32, "v_med3_f32", vreg(vR), vreg(2), m1, &p1);
            return;
        }
        e_.vopc(VOPC_CMP_U_F32, "v_cmp_u_f32_e32", vreg(2), 2);
        e_.vop3(VOP3_MED3_F32, "v_med3_f32", vreg(5), vreg(2), m1, &p1);
        e_.sopp(SOPP_NOP, "s_nop", 0, true);  // 2 wait states between the VALU write of VCC and its VALU read
        e_.vop2(VOP2_CNDMASK, "v_cndmask_b32_e32", vR, vreg(5), 2, ", vcc");
    }

    // NaN operands (exact streams only; a fast stream never sees one).  The x86 build hands on the NaN of the FIRST operand
    // of each SSE instruction, quieted, sign untouched - and which operand is first is g++'s choice per expression, pinned by
    // tests/golden/nan_collisions.json: MACS / MACINTS / MACINTW  X, Y, A;  MACSN / ACC3 / MACW / MACWN  A, X, Y;  INTERP  X, A, Y.
    // gfx950 hands on the NaN of the first SOURCE (src0, src1, src2), quieted - but a negated source (v_sub_f32, neg
    // modifiers) has its sign flipped first (tools/micro/nanrules.hip).  So the exact streams order their sources like
    // the x86 build and never subtract: A - p is A + (-1.0 * p), 1 - X is fma(X, -1.0, 1.0).
    static bool isNanBits(uint32_t bits) { return (bits & 0x7fffffffu) > 0x7f800000u; }
    // a uniform NaN is put into a temporary VGPR so that it can take any source position
    bool orderedOperand(uint32_t word, bool uniform, int temp, int* v) {
        if (!uniform) return row(word, v);
        e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(temp), value(word));
        *v = temp;
        return true;
    }

    // ---- products of a uniform multiplier and a row, kept while the row is not rewritten (fast streams only).
    // "interp s1, s1, k, a ; interp s2, s2, k, a ; ..." or a bank of "macs t, s_i, a, 0.03" multiply the same row by the same
    // constant over and over: the product (and, for INTERP, its conversion to fp64) is computed once into a spare VGPR above
    // the register file and reused until an instruction writes the row.  Entries are created outside SKIP shadows only (all
    // lanes valid) and only when a later instruction asks for the same product; every sample starts with an empty cache.
    struct Product { uint32_t c = 0, row = 0; int vP = -1, vP64 = -1; bool live = false, wide = false; };
    bool inShadow() const { return predOpen_ || regionPreds_ > 0; }
    // the (constant, row) a record multiplies, if it multiplies at all (mirrors macs() / interp())
    bool productKey(const MicroOp& r, uint32_t* c, uint32_t* row) const {
        const uint32_t slot = r.w[0];
        if (slot < AS_MACS || slot >= (uint32_t)kAsmSlots) return false;
        const uint32_t rel = slot - AS_MACS, family = rel / 16, kind = (rel % 16) / 2;
        if (family == 2 || kind == 7) return false;
        uint32_t cb, rw;
        if ((kind & 6u) == 2u) { cb = r.w[3]; rw = r.w[4]; }
        else if ((kind & 6u) == 4u) { cb = r.w[4]; rw = r.w[3]; }
        else return false;
        if ((cb & 0x7fffffffu) == 0x3f800000u) return false;  // +-1.0: no multiplication (INTERP: only a unit Y)
        uint32_t zc;
        if (family == 0 && zeroPlusScaled(r, &zc, nullptr) && pooled(zc) >= 0) return false;  // one fma, no separate product
        *c = cb;
        *row = rw;
        return true;
    }
    bool usedAgain(size_t from, uint32_t c, uint32_t row) const {
        const std::vector<MicroOp>& rec = *records_;
        for (size_t k = from + 1; k < rec.size() && k < from + 600; ++k) {
            if (rec[k].w[0] == AS_ENDSAMPLE) return false;
            uint32_t c2, row2;
            if (productKey(rec[k], &c2, &row2) && c2 == c && row2 == row) return true;
            if (rec[k].w[0] != AS_NOP && rec[k].w[0] != AS_PRED && rec[k].w[0] != AS_UNPRED && rec[k].w[0] != AS_SKIP && rec[k].w[5] == row) return false;
        }
        return false;
    }
    void cseKill(uint32_t row) {
        for (Product& p : products_)
            if (p.live && p.row == row) p.live = false;
    }
    void cseReset() {
        for (Product& p : products_) p.live = false;
    }
    int cseAlloc() {
        for (size_t k = 0; k < products_.size(); ++k)
            if (!products_[k].live) return (int)k;
        for (size_t k = 0; k < products_.size(); ++k)
            if (!usedAgain(index_, products_[k].c, products_[k].row)) return (int)k;
        return -1;
    }

    // p = X * Y: *pv = the VGPR that holds it (v3, or a cache register: then *entry is its cache entry), unless both are
    // uniform (*inV3 false: the record's X word holds the folded product)
    bool product(const MicroOp& r, uint32_t kind, bool* inV3, int* pv = nullptr, int* entry = nullptr) {
        const bool uX = kind & 2, uY = kind & 4;
        *inV3 = !(uX && uY);
        if (pv) *pv = 3;
        if (entry) *entry = -1;
        if (uX && uY) return true;
        Src a;
        int b, dst = 3;
        if (!uX && !uY) {
            if (!operand(r.w[3], false, &a) || !row(r.w[4], &b)) return false;
        } else if (uX) {
            a = value(r.w[3]);
            if (!row(r.w[4], &b)) return false;
        } else if (!fast_ && isNanBits(r.w[4])) {  // X must stay the first source
            int vy;
            if (!operand(r.w[3], false, &a) || !orderedOperand(r.w[4], true, 4, &vy)) return false;
            b = vy;
        } else {
            a = value(r.w[4]);
            if (!row(r.w[3], &b)) return false;
        }
        // (row 0 is the CCR: every instruction with a live CCR writes it on the side - never cached)
        if (fast_ && pv && entry && !products_.empty() && (uX != uY) && (uX ? r.w[4] : r.w[3]) != 0u) {
            const uint32_t c = uX ? r.w[3] : r.w[4], rw = uX ? r.w[4] : r.w[3];
            for (size_t k = 0; k < products_.size(); ++k)
                if (products_[k].live && products_[k].c == c && products_[k].row == rw) {
                    *pv = products_[k].vP;
                    *entry = (int)k;
                    ++stats_.reusedProducts;
                    return true;
                }
            int k;
            if (!inShadow() && usedAgain(index_, c, rw) && (k = cseAlloc()) >= 0) {
                Product& p = products_[(size_t)k];
                p.c = c;
                p.row = rw;
                p.live = true;
                p.wide = false;
                dst = p.vP;
                *pv = dst;
                *entry = k;
            }
        }
        e_.vop2(VOP2_MUL_F32, "v_mul_f32_e32", dst, a, b);
        return true;
    }

    // ---- constants held in SGPRs for the whole loop (VOP3 takes no literal on gfx9): the multipliers of "R = 0 + X * c"
    // with |c| > 0.5.  s88..s93 are free unless the LOG/EXP tables are read from global memory (more than four tables).
    // (Held in VGPRs instead they issue no faster inside a mix and the chip, which runs config5 against its power limit,
    // clocks ~0.5 % lower; replacing the 32-bit literals of plain multiplications by VGPRs costs another ~1 % the same way.)
    void buildConstantPool(const std::vector<MicroOp>& records, bool anyLut) {
        pool_.clear();
        if (!fast_ || (anyLut && prog_.lutTables.empty())) return;
        std::vector<std::pair<uint32_t, int>> freq;
        for (const MicroOp& r : records) {
            uint32_t c;
            if (!zeroPlusScaled(r, &c, nullptr)) continue;
            bool seen = false;
            for (auto& f : freq)
                if (f.first == c) { ++f.second; seen = true; }
            if (!seen) freq.emplace_back(c, 1);
        }
        std::stable_sort(freq.begin(), freq.end(), [](const std::pair<uint32_t, int>& a, const std::pair<uint32_t, int>& b) { return a.second > b.second; });
        for (size_t k = 0; k < freq.size() && k < 6; ++k) pool_.emplace_back(freq[k].first, 88 + (int)k);
    }
    int pooled(uint32_t bits) const {
        for (const auto& c : pool_)
            if (c.first == bits) return c.second;
        return -1;
    }
    // MACS / MACINTS "R = sat(0 + X * c)" with a uniform finite |c| > 0.5 and a per-lane X: the reference's mul-then-add
    // equals ONE fma(X, c, +0).  The add of +0 only matters when the product is -0 (it becomes +0); with |c| > 0.5 a
    // non-zero X never underflows to zero (|X * c| > 2^-150 rounds to at least the smallest denormal), so the product is
    // a zero exactly when X is, and then fma gives (+-0) + (+0) = +0 as well.  Everywhere else fma rounds X * c once,
    // like the multiplication, and adding +0 changes nothing.
    static bool zeroPlusScaled(const MicroOp& r, uint32_t* c, uint32_t* xRow) {
        const uint32_t slot = r.w[0];
        if (slot < AS_MACS || slot >= AS_MACSN) return false;  // MACS family only (not MACSN: 0 - p)
        const uint32_t kind = ((slot - AS_MACS) % 16) / 2;
        if (!(kind & 1u) || r.w[2] != 0u) return false;        // A must be the uniform +0.0
        uint32_t cb, xr;
        if ((kind & 6u) == 2u) { cb = r.w[3]; xr = r.w[4]; }
        else if ((kind & 6u) == 4u) { cb = r.w[4]; xr = r.w[3]; }
        else return false;
        const uint32_t mag = cb & 0x7fffffffu;
        if (mag <= 0x3f000000u || mag >= 0x7f800000u || mag == 0x3f800000u) return false;  // |c| > 0.5, finite; +-1.0 needs no multiply at all
        if (c) *c = cb;
        if (xRow) *xRow = xr;
        return true;
    }

    bool macs(const MicroOp& r, uint32_t kind, bool neg) {
        int vR;
        if (!row(r.w[5], &vR)) return false;
        plainMode();
        {
            uint32_t c, xRow;
            int vX;
            if (fast_ && !neg && zeroPlusScaled(r, &c, &xRow) && pooled(c) >= 0) {
                if (!row(xRow, &vX)) return false;
                const bool within = resultWithinUnit(0, kind, r);
                Src zero = imm32(0);
                e_.vop3(VOP3_FMA_F32, "v_fma_f32", vreg(within ? vR : 2), vreg(vX), sreg(pooled(c)), &zero);
                ++stats_.fusedZeroAdds;
                if (within) ++stats_.unsaturated;
                else satStore(vR);
                return true;
            }
        }
        if (kind == 7) {  // folded on the host: the A word is the saturated result
            e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(vR), value(r.w[2]));
            return true;
        }
        // a multiplier of exactly +-1.0 needs no multiplication: x * 1.0 is x for every x (denormals kept, NaN stays NaN)
        int pv = 3;
        bool inV3;
        const uint32_t unitWord = ((kind & 6u) == 2u) ? r.w[3] : ((kind & 6u) == 4u) ? r.w[4] : 0u;
        if (fast_ && (unitWord & 0x7fffffffu) == 0x3f800000u) {  // (the exact streams multiply: the operand order decides which NaN is handed on)
            if (!row((kind & 6u) == 2u ? r.w[4] : r.w[3], &pv)) return false;
            if (unitWord >> 31) neg = !neg;
            inV3 = true;
            ++stats_.unitMultipliers;
        } else {
            int entry;
            if (!product(r, kind, &inV3, &pv, &entry)) return false;
        }
        const bool within = resultWithinUnit(neg ? 1 : 0, kind, r);  // then the sum goes straight to its row
        const int d = within ? vR : 2;
        if (!fast_) {
            // exact stream: sources in the x86 build's order, no subtraction (see product())
            int vp = pv, vA;
            if (!inV3) {
                e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(3), value(r.w[3]));  // the folded product
                vp = 3;
            }
            if (neg) {
                e_.vop2(VOP2_MUL_F32, "v_mul_f32_e32", 3, imm32(0xbf800000u), vp);  // -p; a NaN keeps its sign
                vp = 3;
            }
            if (!orderedOperand(r.w[2], kind & 1, 12, &vA)) return false;
            if (neg) e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", d, vreg(vA), vp);  // A first
            else e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", d, vreg(vp), vA);      // the product first
            satStore(vR);
            return true;
        }
        if (inV3) {
            Src a;
            if (!operand(r.w[2], kind & 1, &a)) return false;
            e_.vop2(neg ? VOP2_SUB_F32 : VOP2_ADD_F32, neg ? "v_sub_f32_e32" : "v_add_f32_e32", d, a, pv);
        } else {
            int vA;
            if (!row(r.w[2], &vA)) return false;
            e_.vop2(neg ? VOP2_SUBREV_F32 : VOP2_ADD_F32, neg ? "v_subrev_f32_e32" : "v_add_f32_e32", d, value(r.w[3]), vA);
        }
        if (within) ++stats_.unsaturated;
        else satStore(vR);
        return true;
    }

    bool acc3(const MicroOp& r, uint32_t kind) {
        int vR;
        if (!row(r.w[5], &vR)) return false;
        plainMode();
        if (kind == 7) {
            e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(vR), value(r.w[2]));
            return true;
        }
        const bool uA = kind & 1, uX = kind & 2, uY = kind & 4;
        const bool within = resultWithinUnit(2, kind, r);
        const int d = within ? vR : 2;
        if (!fast_) {
            // exact stream: (A + X) + Y with A, X, Y in that source order (NaN priority of the x86 build, see product())
            int vA, vX, vY;
            if (uA && uX) {
                e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(2), value(r.w[2]));  // A + X folded by the host
            } else {
                if (!orderedOperand(r.w[2], uA, 12, &vA) || !orderedOperand(r.w[3], uX, 4, &vX)) return false;
                e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", 2, vreg(vA), vX);
            }
            if (!orderedOperand(r.w[4], uY, 4, &vY)) return false;
            e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", d, vreg(2), vY);
            satStore(vR);
            return true;
        }
        if (uA && uX) {  // t = A + X folded into the A word
            int vY;
            if (!row(r.w[4], &vY)) return false;
            e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", d, value(r.w[2]), vY);
        } else {
            Src a;
            int b;
            if (uA) {
                a = value(r.w[2]);
                if (!row(r.w[3], &b)) return false;
            } else if (uX) {
                a = value(r.w[3]);
                if (!row(r.w[2], &b)) return false;
            } else {
                if (!operand(r.w[2], false, &a) || !row(r.w[3], &b)) return false;
            }
            e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", 2, a, b);
            Src y;
            if (!operand(r.w[4], uY, &y)) return false;
            e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", d, y, 2);
        }
        if (within) ++stats_.unsaturated;
        else satStore(vR);
        return true;
    }

    // INTERP (FX8010.cpp:1180-1187): R = sat((float)((1.0 - (double)X) * (double)A + (double)(X*Y)))
    bool interp(const MicroOp& r, uint32_t kind) {
        int vR;
        if (!row(r.w[5], &vR)) return false;
        plainMode();
        if (kind == 7) {
            e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(vR), value(r.w[2]));
            return true;
        }
        if ((kind & 6u) == 4u && r.w[4] == 0x3f800000u) {  // Y = 1.0: the product is X itself
            int vX;
            if (!row(r.w[3], &vX)) return false;
            ++stats_.unitMultipliers;
            return interpTail(r, kind, vreg(vX), vR);
        }
        bool inV3;
        int pv, entry;
        if (!product(r, kind, &inV3, &pv, &entry)) return false;
        return interpTail(r, kind, inV3 ? vreg(pv) : value(r.w[3]), vR, entry);
    }
    // the fp32 product as an fp64 operand: converted into v[tmp:tmp+1], or taken from / put into its cache entry
    Src wideProduct(const Src& p, int entry, int tmp) {
        if (entry >= 0 && products_[(size_t)entry].vP64 >= 0) {
            Product& e = products_[(size_t)entry];
            if (!e.wide && !inShadow()) {
                e_.vop1(VOP1_CVT_F64_F32, "v_cvt_f64_f32_e32", vreg64(e.vP64), p);
                e.wide = true;
            }
            if (e.wide) return vreg64(e.vP64);
        }
        e_.vop1(VOP1_CVT_F64_F32, "v_cvt_f64_f32_e32", vreg64(tmp), p);
        return vreg64(tmp);
    }
    // p = fp32 product X*Y (a VGPR or the folded constant); entry = its product-cache entry or -1
    bool interpTail(const MicroOp& r, uint32_t kind, const Src& p, int vR, int entry = -1) {
        const bool within = resultWithinUnit(3, kind, r);
        const int d = within ? vR : 2;
        if (within) ++stats_.unsaturated;
        Src a;
        if (!operand(r.w[2], kind & 1, &a)) return false;
        e_.vop1(VOP1_CVT_F64_F32, "v_cvt_f64_f32_e32", vreg64(8), a);
        if (kind & 2) {  // uniform X: the record carries (1.0 - (double)X)
            const uint64_t omx = (uint64_t)r.w[6] | ((uint64_t)r.w[7] << 32);
            const InlineD* inl = nullptr;
            for (const InlineD& k : kInlineF64)
                if (k.bits == omx) inl = &k;
            if (!inl) {
                setRecordWord(6, r.w[6]);
                setRecordWord(7, r.w[7]);
            }
            const Src m = inl ? named(inl->code, inl->text) : sreg64(kSRecord + 6);
            if (productWithFloatIsExact(omx)) {
                // (1-X)*A is exact in fp64 for every float A, so mul-then-add rounds once - exactly what one fma does
                Src addend = wideProduct(p, entry, 10);
                e_.vop3(VOP3_FMA_F64, "v_fma_f64", vreg64(6), m, vreg64(8), &addend);
                e_.vop1(VOP1_CVT_F32_F64, "v_cvt_f32_f64_e32", vreg(d), vreg64(6));
                if (!within) satStore(vR);
                return true;
            }
            e_.vop3(VOP3_MUL_F64, "v_mul_f64", vreg64(6), m, vreg64(8), nullptr);
        } else {
            Src x;
            if (!operand(r.w[3], false, &x)) return false;
            e_.vop1(VOP1_CVT_F64_F32, "v_cvt_f64_f32_e32", vreg64(6), x);
            // 1.0 - X as fma(X, -1.0, 1.0): one rounding as well, and a NaN in X keeps its sign (a negated source flips it)
            Src plusOne = named(242, "1.0");
            e_.vop3(VOP3_FMA_F64, "v_fma_f64", vreg64(6), vreg64(6), named(243, "-1.0"), &plusOne);
            e_.vop3(VOP3_MUL_F64, "v_mul_f64", vreg64(6), vreg64(6), vreg64(8), nullptr);
        }
        const Src wide = wideProduct(p, entry, 8);
        e_.vop3(VOP3_ADD_F64, "v_add_f64", vreg64(6), vreg64(6), wide, nullptr);
        e_.vop1(VOP1_CVT_F32_F64, "v_cvt_f32_f64_e32", vreg(d), vreg64(6));
        if (!within) satStore(vR);
        return true;
    }

    // Is d * (double)f exact for every float f?  A float has 24 significant bits, a double 53: yes when d has at
    // most 29 (zero, or a normal number whose low 24 mantissa bits are clear; subnormal d: no claim).
    static bool productWithFloatIsExact(uint64_t dbits) {
        if ((dbits << 1) == 0) return true;
        const uint32_t exponent = (uint32_t)(dbits >> 52) & 0x7ffu;
        if (exponent == 0 || exponent == 0x7ffu) return false;
        return (dbits & 0xffffffull) == 0;
    }

    // A sync point: where a wavefront of the FAST stream may arrive in the exact stream (a lane met a non-finite value).  What the
    // record words s16..s23 hold there is what the fast stream left in them - and the two streams do not set them alike (the
    // fast stream drops a dead INTERP whose (1 - X) the exact stream had counted on: API fuzz, control panel, seed 2605911) - so
    // the exact stream forgets what it knew about them.  (The pair a staged program's cold entries load once is the same in both.)
    void syncPoint(size_t index, uint32_t at) {
        returns_[index] = at;
        if (fast_) return;
        e_.note("sync point");
        for (int k = 0; k < 8; ++k) known_[k] = false;
        if (omxHoisted_) {
            known_[6] = known_[7] = true;
            value_[6] = hoistedLo_;
            value_[7] = hoistedHi_;
        }
    }
    // s(16+k) = word k of the record, unless it holds that value already
    void setRecordWord(int k, uint32_t value) {
        if (known_[k] && value_[k] == value) return;
        e_.sop1(SOP1_MOV_B32, "s_mov_b32", sreg(kSRecord + k), imm32(value));
        known_[k] = true;
        value_[k] = value;
    }

    // LIMIT / LIMITN (FX8010.cpp:1163-1174): R = (A >= Y) ? X : Y  resp.  (A < Y) ? X : Y - a compare and a select, no
    // arithmetic: nothing depends on the order of sources, so both streams generate the same code (h_limit / h_limitn are
    // the interpreter's versions).  An ordered compare is false for a NaN: Y is chosen, as in the reference.
    bool limitInline(const MicroOp& r, uint32_t slot, bool ccrLive) {
        const uint32_t kind = r.w[6] & 7u;
        int vR;
        if (!touch(r, !(kind & 1u), !(kind & 2u), !(kind & 4u), true) || !row(r.w[5], &vR)) return false;
        plainMode();
        Src a, x, y;
        if (!operand(r.w[2], kind & 1u, &a) || !operand(r.w[3], (kind & 2u) != 0, &x) || !operand(r.w[4], (kind & 4u) != 0, &y)) return false;
        int vX = 3, vY = 4;
        if (kind & 2u) e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(3), x);
        else if (!row(r.w[3], &vX)) return false;
        if (kind & 4u) e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(4), y);
        else if (!row(r.w[4], &vY)) return false;
        if (slot == AS_LIMIT) e_.vopc(VOPC_CMP_GE_F32, "v_cmp_ge_f32_e32", a, vY);
        else e_.vopc(VOPC_CMP_LT_F32, "v_cmp_lt_f32_e32", a, vY);
        e_.sopp(SOPP_NOP, "s_nop", 1, true);
        e_.vop2(VOP2_CNDMASK, "v_cndmask_b32_e32", vR, vreg(vY), vX, ", vcc");
        if (ccrLive) ccrOrSkip(vR);
        return true;
    }

    // MACW / MACWN / MACINTW (FX8010.cpp:1126-1143, wrapAround :299-328) inline in the FAST stream: every operand is finite
    // there, so the order of the sources does not matter; the exact stream keeps calling the interpreter's handler, whose
    // sources are ordered like the x86 build's.  The result is not saturated: it can overflow, so it is checked like a
    // value from memory, and the wave leaves for the exact stream behind the same record there - after the CCR, which the
    // handler writes before it returns.   w(a) = a >= 1 ? a - 2 : (a < -1 ? a + 2 : a)
    void wrapInto(int dst, int src) {   // dst may be src; v5, v6 scratch
        const Src vcc = named(106, "vcc");
        e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", 5, imm32(0xc0000000u), src);                   // a - 2
        e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", 6, imm32(0x40000000u), src);                   // a + 2
        e_.vopc(VOPC_CMP_GT_F32, "v_cmp_gt_f32_e32", imm32(0xbf800000u), src);                // -1 > a
        e_.sopp(SOPP_NOP, "s_nop", 1, true);
        e_.vop2(VOP2_CNDMASK, "v_cndmask_b32_e32", 6, vreg(src), 6, ", vcc");
        e_.vopc(VOPC_CMP_LE_F32, "v_cmp_le_f32_e32", imm32(0x3f800000u), src);                // 1 <= a
        e_.sopp(SOPP_NOP, "s_nop", 1, true);
        e_.vop2(VOP2_CNDMASK, "v_cndmask_b32_e32", dst, vreg(6), 5, ", vcc");
        (void)vcc;
    }
    bool wrapFamily(const MicroOp& r, uint32_t slot, bool ccrLive) {
        const uint32_t kind = r.w[6] & 7u;
        int vR;
        // (the exact stream calls the handler here and waits for every delay-line read in flight first: the two streams must
        // wait - and define their sync points - at the same records)
        if (!flush() || !row(r.w[5], &vR)) return false;
        // ... and leave the record words in s18..s22 as its call does: later calls of EITHER stream skip the words they know
        // to be there already, and a wave may change streams in between
        for (int k = 2; k < 7; ++k) setRecordWord(k, k == 6 && !ccrLive ? (r.w[6] & ~8u) : r.w[k]);
        plainMode();
        int pv = 3;
        if ((kind & 6u) == 6u) {  // both factors uniform: the product here (finite: a non-finite uniform disables the fast stream)
            const float p = asFloat(r.w[3]) * asFloat(r.w[4]);
            uint32_t pb;
            std::memcpy(&pb, &p, 4);
            if ((pb & 0x7f800000u) == 0x7f800000u) return fail("internal: non-finite product of uniforms in a fast stream");
            e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(3), imm32(pb));
        } else {
            bool inV3;
            int entry;
            if (!product(r, kind, &inV3, &pv, &entry)) return false;
        }
        Src a;
        if (!operand(r.w[2], kind & 1u, &a)) return false;
        if (slot == AS_MACINTW) {
            e_.vop2(VOP2_ADD_F32, "v_add_f32_e32", 2, a, pv);                                 // A + X*Y
            wrapInto(vR, 2);
        } else {
            wrapInto(3, pv);                                                                  // (never into a cached product)
            e_.vop2(slot == AS_MACW ? VOP2_ADD_F32 : VOP2_SUB_F32, slot == AS_MACW ? "v_add_f32_e32" : "v_sub_f32_e32", vR, a, 3);
        }
        if (ccrLive) ccrFrom(vR);
        syncPoint(syncIndex(1), base_ + (uint32_t)e_.bytes());
        taintIfNonFinite(vR);
        return leaveIfTainted((*exactReturns_)[syncIndex(1)]);
    }

    // run the interpreter's handler for this record: operands in s18..s23, return address in s[24:25]
    bool call(const MicroOp& r, uint32_t slot, uint32_t wordMask) {
        if (slot >= (uint32_t)kAsmSlots) return fail("record with an unknown handler slot");
        if (!flush()) return false;  // the handler addresses rows by index
        for (int k = 2; k < 8; ++k)
            if (wordMask & (1u << k)) setRecordWord(k, r.w[k]);
        const uint32_t at = base_ + (uint32_t)e_.bytes();
        const uint32_t ret = at + 28;  // 8 + 4 + 8 + 4 + 4 bytes below
        e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSReturn), sreg(kSEntry), imm32(ret, true));
        e_.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSReturn + 1), sreg(kSEntry + 1), imm32(0));
        e_.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSTemp), sreg(kSEntry), imm32(tmpl_.handlerOff[slot], true));
        e_.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSTemp + 1), sreg(kSEntry + 1), imm32(0));
        e_.sop1NoDst(SOP1_SETPC, "s_setpc_b64", sreg64(kSTemp));
        if (base_ + (uint32_t)e_.bytes() != ret) return fail("internal: call sequence length");
        syncPoint(syncIndex(1), ret);
        indexModeUnknown_ = true;
        ++stats_.called;
        const bool canTaint = slot == AS_MACW || slot == AS_MACWN || slot == AS_MACINTW || slot == AS_LUT || slot == AS_TRAM_IR || slot == AS_TRAM_XR;
        // the handler may have met a non-finite value: continue in the exact stream, after the same call there
        if (fast_ && canTaint && !leaveIfTainted((*exactReturns_)[syncIndex(1)])) return false;
        return true;
    }

    // ---- CCR, SKIP and its shadow ------------------------------------------------------------------------
    // setCCR (FX8010.cpp:211-232) of the value in vR into the CCR row (v32); same instruction sequence as the
    // interpreter's CCR_FROM
    void ccrFrom(int vR) {
        const Src vcc = named(106, "vcc"), t0 = sreg64(62), t1 = sreg64(64), t2 = sreg64(66), r = vreg(vR), one = imm32(0x3f800000u), zero = imm32(0);
        e_.vop3cmpG(VOP3_CMP_LT_F32, "v_cmp_lt_f32_e64", vcc, r, true, one);   // |r| < 1
        e_.vop3cmpG(VOP3_CMP_EQ_F32, "v_cmp_eq_f32_e64", t0, r, true, one);    // |r| == 1
        e_.vop3cmpG(VOP3_CMP_GT_F32, "v_cmp_gt_f32_e64", t1, zero, false, r);  // r < 0
        e_.vop3cmpG(VOP3_CMP_EQ_F32, "v_cmp_eq_f32_e64", t2, zero, false, r);  // r == 0
        Src two = imm32(2), sixteen = imm32(16), eight = imm32(8), v6 = vreg(6), v7 = vreg(7);
        e_.vop3(VOP3_CNDMASK, "v_cndmask_b32_e64", v6, zero, two, &vcc);         // normalised: 2
        e_.sop2(SOP2_OR_B64, "s_or_b64", vcc, vcc, t0);
        e_.vop3(VOP3_CNDMASK, "v_cndmask_b32_e64", v6, v6, sixteen, &t0);        // saturated: 16
        e_.sop2(SOP2_AND_B64, "s_and_b64", t1, t1, vcc);                        // negative and |r| <= 1: +4 (6 / 20)
        e_.vop2(VOP2_ADD_U32, "v_add_u32_e32", 7, imm32(4), 6);
        e_.vop3(VOP3_CNDMASK, "v_cndmask_b32_e64", v6, v6, v7, &t1);
        e_.vop3(VOP3_CNDMASK, "v_cndmask_b32_e64", v6, v6, eight, &t2);          // zero: 8
        e_.vop1(VOP1_CVT_F32_U32, "v_cvt_f32_u32_e32", vreg(vrow(0)), v6);
    }

    static float asFloat(uint32_t bits) {
        float f;
        std::memcpy(&f, &bits, 4);
        return f;
    }
    // SKIP with uniform X and Y (FX8010.cpp:1175-1179): the CCR value it waits for and the count it sets
    static bool uniformSkip(const MicroOp& r, float* want, int32_t* count) {
        if (r.w[0] != AS_SKIP || (r.w[6] & 6u) != 6u) return false;
        *want = (float)x86Trunc(r.w[3]);
        *count = x86Trunc(r.w[4]);
        return true;
    }
    // Does anything observe the CCR row after record j before an instruction overwrites it?  (Conservative: the end
    // of the stream counts as an observer.)
    bool ccrDeadAfter(size_t j) const {
        const std::vector<MicroOp>& rec = *records_;
        bool shadowed = false;  // a write inside a SKIP shadow happens in some lanes only: it kills nothing
        for (size_t k = j + 1; k < rec.size(); ++k) {
            const uint32_t slot = rec[k].w[0];
            if (slot == AS_ENDSAMPLE || slot == AS_SKIP) return false;
            if (slot == AS_PRED) shadowed = true;
            if (slot == AS_UNPRED) shadowed = false;
            if (slot == AS_NOP || slot == AS_PRED || slot == AS_UNPRED) continue;
            uint32_t kind, ccr;
            if (slot >= AS_MACS) { kind = ((slot - AS_MACS) % 16) / 2; ccr = (slot - AS_MACS) & 1u; }
            else { kind = rec[k].w[6] & 7u; ccr = (rec[k].w[6] >> 3) & 1u; }
            if ((!(kind & 1u) && rec[k].w[2] == 0) || (!(kind & 2u) && rec[k].w[3] == 0) || (!(kind & 4u) && rec[k].w[4] == 0)) return false;
            if (ccr && !shadowed) return true;
        }
        return false;
    }
    // A SKIP outside every shadow, followed by exactly the `count` shadowed instructions it can skip and then
    // unshadowed code: its whole shadow runs under one EXEC mask instead of a PRED per instruction.
    bool simpleShadow(size_t j, int32_t count) const {
        if (predOpen_ || count < 1) return false;
        const std::vector<MicroOp>& rec = *records_;
        int32_t groups = 0;
        for (size_t k = j + 1; k < rec.size(); ++k) {
            const uint32_t slot = rec[k].w[0];
            if (slot == AS_PRED) { ++groups; continue; }
            if (groups == 0) return false;  // the instruction after the SKIP is not shadowed
            if (slot == AS_SKIP) return false;
            if (slot == AS_UNPRED || slot == AS_ENDSAMPLE) return groups == count;
        }
        return false;
    }
    // vcc = lanes whose CCR would equal `want`, straight from the value in vR (setCCR inverted)
    bool ccrPredicate(float want, int vR) {
        const Src vcc = named(106, "vcc"), r = vreg(vR), zero = imm32(0), one = imm32(0x3f800000u), mone = imm32(0xbf800000u);
        if (want == 8.0f) e_.vopc(VOPC_CMP_EQ_F32, "v_cmp_eq_f32_e32", zero, vR);
        else if (want == 16.0f) e_.vopc(VOPC_CMP_EQ_F32, "v_cmp_eq_f32_e32", one, vR);
        else if (want == 20.0f) e_.vopc(VOPC_CMP_EQ_F32, "v_cmp_eq_f32_e32", mone, vR);
        else if (want == 0.0f) e_.vop3cmpG(VOP3_CMP_NLE_F32, "v_cmp_nle_f32_e64", vcc, r, true, one);  // |r| > 1 or NaN
        else if (want == 2.0f || want == 6.0f) {
            // 0 < r < 1 (bits 0x00000001 .. 0x3F7FFFFF) or -1 < r < 0 (0x80000001 .. 0xBF7FFFFF): an open interval of floats is a
            // closed interval of bit patterns - denormals inside, +-0, +-1, Inf and NaN outside - so one integer add (double-rate
            // class) and one unsigned compare do for the two float compares and the scalar AND
            e_.vop2(VOP2_ADD_U32, "v_add_u32_e32", 5, imm32(want == 2.0f ? 0xffffffffu : 0x7fffffffu), vR);
            e_.vopc(VOPC_CMP_GT_U32, "v_cmp_gt_u32_e32", imm32(0x3f7fffffu), 5);
        } else return false;  // CCR never takes this value
        return true;
    }
    // lanes in vcc take the skip: numSkip = count, or - for a simple shadow - leave EXEC until the UNPRED
    void takeSkip(size_t j, int32_t count) {
        const Src vcc = named(106, "vcc"), exec = named(126, "exec");
        const int32_t n = count < 0 ? 1 : count;  // a negative count skips exactly one instruction (FX8010.cpp:1238)
        if (simpleShadow(j, n)) {
            e_.sop2(SOP2_ANDN2_B64, "s_andn2_b64", exec, exec, vcc);
            e_.vop2(VOP2_ADD_U32, "v_add_u32_e32", kVShadowCount, imm32((uint32_t)n), kVShadowCount);  // the others execute all n
            regionPreds_ = n;
            ++stats_.regions;
            return;
        }
        Src c = imm32((uint32_t)count);
        if (c.hasLit) {
            e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(5), c);
            c = vreg(5);
        } else {
            e_.sopp(SOPP_NOP, "s_nop", 0, true);
        }
        e_.sopp(SOPP_NOP, "s_nop", 0, true);  // two wait states between the VALU write of VCC and its VALU read
        e_.vop3(VOP3_CNDMASK, "v_cndmask_b32_e64", vreg(kVNumSkip), vreg(kVNumSkip), c, &vcc);
    }
    bool skipInline(const MicroOp& r) {
        float want;
        int32_t count;
        if (!uniformSkip(r, &want, &count)) return false;
        plainMode();
        e_.vopc(VOPC_CMP_EQ_F32, "v_cmp_eq_f32_e32", imm32(asBitsOf(want)), kRegFileBase);
        takeSkip(index_, count);
        return true;
    }
    static uint32_t asBitsOf(float f) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        return u;
    }
    // after a saturating instruction (or MACMV) has left its result in vR with a live CCR: derive the CCR - or, when
    // the only observer is the SKIP that follows, that SKIP's predicate directly
    void ccrOrSkip(int vR) {
        const std::vector<MicroOp>& rec = *records_;
        size_t j = index_ + 1;
        while (j < rec.size() && rec[j].w[0] == AS_NOP) ++j;
        float want;
        int32_t count;
        if (j < rec.size() && uniformSkip(rec[j], &want, &count) && ccrDeadAfter(j)) {
            if (ccrPredicate(want, vR)) takeSkip(j, count);
            consumed_ = j;
            ++stats_.fusedSkips;
            return;
        }
        ccrFrom(vR);
    }

    bool one(const MicroOp& r, uint32_t slot) {
        // (generic records carry their CCR flag in w6; a write the stream overwrites before anything reads it is dead)
        const uint32_t ccrLive = (slot < AS_MACS && ((r.w[6] >> 3) & 1u) && !ccrDeadAfter(index_)) ? 1u : 0u;
        if (slot == AS_NOP) return true;  // END / NOP only count (staticCount)
        if (slot == AS_UNPRED) {
            e_.sop1(SOP1_MOV_B64, "s_mov_b64", named(126, "exec"), named(193, "-1"));
            ++stats_.inlined;
            predOpen_ = false;
            regionPreds_ = 0;
            return true;
        }
        if (slot == AS_PRED && regionPreds_ > 0) {  // inside a simple shadow: EXEC already excludes the skipping lanes
            --regionPreds_;
            predOpen_ = true;
            ++stats_.inlined;
            return true;
        }
        if (slot == AS_SKIP && skipInline(r)) {
            ++stats_.inlined;
            return true;
        }
        if (slot == AS_PRED) {
            predOpen_ = true;
            // instruction inside a SKIP shadow (FX8010.cpp:1037,1235-1241): lanes with numSkip == 0 execute it,
            // the others count their skip down; v15 counts the executed ones
            plainMode();
            e_.sop1(SOP1_MOV_B64, "s_mov_b64", named(126, "exec"), named(193, "-1"));
            e_.vopc(VOPC_CMP_EQ_U32, "v_cmp_eq_u32_e32", imm32(0), kVNumSkip);
            e_.vop2(VOP2_MAX_I32, "v_max_i32_e32", 5, imm32(1), kVNumSkip);
            e_.vop2(VOP2_ADD_U32, "v_add_u32_e32", kVNumSkip, imm32(0xffffffffu), 5);
            Src one1 = imm32(1), vcc = named(106, "vcc");
            e_.vop3(VOP3_CNDMASK, "v_cndmask_b32_e64", vreg(5), imm32(0), one1, &vcc);
            e_.vop2(VOP2_ADD_U32, "v_add_u32_e32", kVShadowCount, vreg(kVShadowCount), 5);
            e_.sop1(SOP1_MOV_B64, "s_mov_b64", named(126, "exec"), named(106, "vcc"));
            ++stats_.inlined;
            return true;
        }
        if (slot == AS_LUT && !(r.w[6] & 1u) && (r.w[6] & 2u)) {   // (a per-instance table number: the handler)
            ++stats_.inlined;
            if (!lut(r)) return false;
            if (ccrLive) ccrFrom(vrow(r.w[5]));
            return true;
        }
        if (prog_.uniformCursors && slot >= AS_TRAM_IR && slot <= AS_TRAM_XW) {
            ++stats_.inlined;
            return tram(r, slot);
        }
        if (slot >= AS_MACS && slot < (uint32_t)kAsmSlots) {
            const uint32_t rel = slot - AS_MACS, family = rel / 16, kind = (rel % 16) / 2;
            // a CCR write that the stream itself overwrites before anything can read it is dead (the last-sample
            // stream marks every write live; only the final one is state)
            const uint32_t ccr = (rel & 1u) && !ccrDeadAfter(index_) ? 1u : 0u;
            // a result nobody reads (deadWrite_): nothing to compute in the fast stream - unless its CCR is wanted, and then, for
            // the test idiom `macs tmp, x, 0, 0` in front of a SKIP, the SKIP's predicate comes straight from x: 0 + x differs from
            // x only for x = -0, which no CCR value tells from +0 (ccrPredicate: every compare treats them alike), and with x of
            // the bounded class the saturation is the identity.  (The exact stream computes everything: a wave is there because
            // some value left its class.)
            if (fast_ && deadWrite_[index_]) {
                // (the rows the record names are waited for all the same: both streams keep their sync points at the same records)
                if (!touch(r, !(kind & 1u), !(kind & 2u), !(kind & 4u), true)) return false;
                if (!ccr) { ++stats_.inlined; ++stats_.deadResults; return true; }
                size_t j = index_ + 1;
                while (j < records_->size() && (*records_)[j].w[0] == AS_NOP) ++j;
                float want;
                int32_t count;
                int vA;
                if (family == 0 && kind == 6u && r.w[3] == 0u && bound(r.w[2], false) <= 1.0 && j < records_->size() &&
                    uniformSkip((*records_)[j], &want, &count) && ccrDeadAfter(j) && row(r.w[2], &vA)) {
                    plainMode();
                    ++stats_.inlined;
                    ++stats_.deadResults;
                    ccrOrSkip(vA);
                    return true;
                }
            }
            if (!ccr) {
                if (!touch(r, !(kind & 1u), !(kind & 2u), !(kind & 4u), true)) return false;
                ++stats_.inlined;
                switch (family) {
                    case 0: return macs(r, kind, false);
                    case 1: return macs(r, kind, true);
                    case 2: return acc3(r, kind);
                    default: return interp(r, kind);
                }
            }
            // a live CCR: the same code as above, then setCCR of the stored result - or the predicate of the SKIP that
            // reads it.  (Last-sample streams mark every CCR write live, but all except the final ones are overwritten
            // before anything reads them - ccrDeadAfter - so this stays short there too.)
            if (!touch(r, !(kind & 1u), !(kind & 2u), !(kind & 4u), true)) return false;
            ++stats_.inlined;
            bool ok;
            switch (family) {
                case 0: ok = macs(r, kind, false); break;
                case 1: ok = macs(r, kind, true); break;
                case 2: ok = acc3(r, kind); break;
                default: ok = interp(r, kind); break;
            }
            if (!ok) return false;
            ccrOrSkip(vrow(r.w[5]));
            return true;
        }
        if (slot == AS_MOV) {
            int vR;
            Src a;
            if (!touch(r, !(r.w[6] & 1u), false, false, true)) return false;
            if (!row(r.w[5], &vR) || !operand(r.w[2], r.w[6] & 1u, &a)) return false;
            plainMode();
            e_.vop1(VOP1_MOV, "v_mov_b32_e32", vreg(vR), a);
            if (ccrLive) ccrOrSkip(vR);
            ++stats_.inlined;
            return true;
        }
        if (slot == AS_LIMIT || slot == AS_LIMITN) {
            ++stats_.inlined;
            return limitInline(r, slot, ccrLive != 0);
        }
        if (fast_ && (slot == AS_MACW || slot == AS_MACWN || slot == AS_MACINTW)) {
            ++stats_.inlined;
            return wrapFamily(r, slot, ccrLive != 0);
        }
        if (slot < AS_MACS && ((r.w[6] >> 3) & 1u) && !ccrLive) {
            MicroOp q = r;
            q.w[6] &= ~8u;  // the handler need not derive a CCR nobody sees
            return call(q, slot, 0x7cu);
        }
        return call(r, slot, 0x7cu);  // generic handlers read w2..w6
    }

    const XlateTemplate& tmpl_;
    const XlateProgram prog_;
    uint32_t base_;
    bool isLast_;
    uint32_t nextBase_;  // steady streams: where the last-sample stream of the same flavour starts
    Emitter e_;
    XlateStats stats_;
    std::string err_;
    bool fast_;
    const std::vector<uint32_t>* exactReturns_;
    std::vector<uint32_t> returns_;  // sync points of this stream (see run())
    std::vector<int> pending_;       // VGPRs with a TRAM read in flight
    std::vector<std::pair<uint32_t, int>> pool_;  // uniform constants kept in SGPRs for the whole loop: (bits, SGPR)
    std::vector<Deferred> deferred_;
    std::vector<Product> products_;
    std::vector<uint8_t> deadWrite_;   // per record: its result is overwritten before anything reads it
    bool deferredFailed_ = false;
    const std::vector<MicroOp>* records_ = nullptr;
    size_t consumed_ = (size_t)-1;   // record already translated together with its predecessor
    bool predOpen_ = false;          // EXEC is restricted by a PRED / a simple shadow
    int32_t regionPreds_ = 0;        // PRED records of the current simple shadow still to come
    bool segKnown_ = false;          // s[92:93] holds the segment base of table offset segOff_
    uint32_t segOff_ = 0;
    size_t index_ = 0;
    bool nonFinite_ = false;
    bool indexModeUnknown_ = false;  // the per-sample frame enters the stream with index mode off
    bool known_[8] = {};
    bool omxHoisted_ = false;
    uint32_t hoistedLo_ = 0, hoistedHi_ = 0;
    uint32_t value_[8] = {};
};

}  // namespace

bool translateStream(const std::vector<MicroOp>& records, const XlateTemplate& tmpl, const XlateProgram& prog, uint32_t codeBase,
                     bool isLast, uint32_t nextBase, const std::vector<uint32_t>* exactReturns, std::vector<uint32_t>* code,
                     std::string* listing, XlateStats* stats, std::vector<uint32_t>* returns, uint32_t* coldEntry, std::string* err) {
    code->clear();
    Translator t(tmpl, prog, codeBase, isLast, nextBase, code, listing, exactReturns);
    return t.run(records, stats, returns, coldEntry, err);
}

// Run-once code (one wavefront = one workgroup; entered from the template with the TRAM cursors in v16..v19 and
// s95 = 0, returns through s[24:25]):
//  * the LOG/EXP tables the program uses -> LDS, lane g writing segment g of every array (layout: kLds*);
//  * s95 = 1 when this launch may issue its leading TRAM reads one sample ahead (HoistPlan).
void xl::emitInit(const XlateProgram& prog, std::vector<uint32_t>* code, std::string* listing, int sliceBias) {
    Emitter e(code, listing);
    if (sliceBias >= 0) {
        // time-sliced priorities (Translator::run): a slice of 2^shift ticks of the 100 MHz clock is about 1/24 of the block -
        // shift = floor(log2(samples of the block)) + sliceBias (= log2 of a sample period in ticks / 24, from the code's modelled
        // issue time), kept between 2^16 (0.66 ms: a wavefront looks at the clock every fourth sample, and a low-priority one is
        // slow - shorter slices than that it misses; measured on blocks of 64 ... 512 samples) and 2^20 ticks
        e.sop1(SOP1_FLBIT_I32_B32, "s_flbit_i32_b32", sreg(kSSliceShift), sreg(kSNumSamples));
        e.sop2(SOP2_SUB_I32, "s_sub_i32", sreg(kSSliceShift), imm32((uint32_t)(31 + sliceBias)), sreg(kSSliceShift));
        static const int minShift = knobInt(FX_DIAG_KNOB("FX_XLATE_PRIO_MINSHIFT"), 16);   // (diagnostics)
        e.sop2(SOP2_MAX_I32, "s_max_i32", sreg(kSSliceShift), sreg(kSSliceShift), imm32((uint32_t)minShift));
        e.sop2(SOP2_MIN_I32, "s_min_i32", sreg(kSSliceShift), sreg(kSSliceShift), imm32(20));
    }
    if (!prog.lutTables.empty()) {
        const LutLdsLayout& L = lutLds();
        e.vop2(VOP2_LSHLREV_B32, "v_lshlrev_b32_e32", 2, imm32(2), 0);  // v2 = lane * 4 (v0 = lane)
        e.vop2(VOP2_LSHLREV_B32, "v_lshlrev_b32_e32", 3, imm32(3), 0);  // v3 = lane * 8
        e.vop2(VOP2_LSHLREV_B32, "v_lshlrev_b32_e32", 12, imm32(4), 0); // v12 = lane * 16
        const int cell = L.wide ? 12 : 3;                               // LDS address of lane's cell: lane * 16 / lane * 8
        e.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSAddr), sreg(kSLut), imm32((uint32_t)kLutXthrOff * 8, true));
        e.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSAddr + 1), sreg(kSLut + 1), imm32(0));
        e.globalLoadWide(GLOBAL_LOAD_DWORDX2, 2, 4, 2, kSAddr);        // xthr[lane], xthr[lane + 1]
        e.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSAddr), sreg(kSLut), imm32((uint32_t)kLutX1Off * 8, true));
        e.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSAddr + 1), sreg(kSLut + 1), imm32(0));
        e.globalLoadWide(GLOBAL_LOAD_DWORDX2, 2, 6, 3, kSAddr);        // x1[lane]
        e.waitVmcnt(0);
        e.dsWriteB64(cell, 4, L.thr);
        e.dsWriteB64(cell, 6, L.x1);
        for (size_t k = 0; k < prog.lutTables.size(); ++k) {
            const uint32_t table = L.tables + (uint32_t)k * L.tableBytes;
            e.sop2(SOP2_ADD_U32, "s_add_u32", sreg(kSAddr), sreg(kSLut), imm32(prog.lutTables[k], true));
            e.sop2(SOP2_ADDC_U32, "s_addc_u32", sreg(kSAddr + 1), sreg(kSLut + 1), imm32(0));
            e.globalLoadWide(GLOBAL_LOAD_DWORDX4, 4, 4, 12, kSAddr);   // {slope, y1} of segment lane
            e.waitVmcnt(0);
            if (L.wide) {
                e.dsWriteB128(12, 4, table);
            } else {
                e.dsWriteB64(3, 4, table);
                e.dsWriteB64(3, 6, table + 512);
            }
        }
        e.waitLgkm0();
    }
    const HoistPlan& H = prog.hoist;
    if (H.leadCount > 0) {
        e.sop1(SOP1_MOV_B32, "s_mov_b32", sreg(kSHoistOk), imm32(1));
        for (int t = 0; t < 2; ++t) {
            if (H.forbidden[t].empty()) continue;
            // distance = (read cursor - write cursor) mod size; any listed value: keep the reads in place
            e.vop1(VOP1_READFIRSTLANE, "v_readfirstlane_b32", sreg(kSPos), vreg(kVCursor + 2 * t + 1));
            e.vop1(VOP1_READFIRSTLANE, "v_readfirstlane_b32", sreg(kSAddr), vreg(kVCursor + 2 * t));
            e.sop2(SOP2_SUB_I32, "s_sub_i32", sreg(kSPos), sreg(kSPos), sreg(kSAddr));
            e.sopc(SOPC_CMP_LT_I32, "s_cmp_lt_i32", sreg(kSPos), imm32(0));
            Emitter::Fixup pos = e.branchForward(SOPP_CBRANCH_SCC0, "s_cbranch_scc0");
            e.sop2(SOP2_ADD_I32, "s_add_i32", sreg(kSPos), sreg(kSPos), sreg(kSTramSize[t]));
            e.bind(pos);
            for (uint32_t d : H.forbidden[t]) {
                e.sopc(SOPC_CMP_EQ_U32, "s_cmp_eq_u32", sreg(kSPos), imm32(d));
                e.sop2(SOP2_CSELECT_B32, "s_cselect_b32", sreg(kSHoistOk), imm32(0), sreg(kSHoistOk));
            }
        }
    }
    e.sop1NoDst(SOP1_SETPC, "s_setpc_b64", sreg64(kSReturn));
    e.finish();
}

namespace {
int32_t truncX86(uint32_t bits) {  // cvttss2si: 0x80000000 for NaN and anything outside int32
    float f;
    std::memcpy(&f, &bits, 4);
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT32_MIN;
    return (int32_t)f;
}

}  // namespace

// tap position of the opt-in DANE model from a uniform operand: whole samples, or a DANE address fraction (value * 2^31,
// 0x800 per sample); reduced to 0 .. size-1
int32_t xl::danePosition(uint32_t bits, bool shifted, int32_t size) {
    int64_t q;
    if (shifted) {
        float f;
        std::memcpy(&f, &bits, 4);
        const float scaled = f * 2147483648.0f;
        uint32_t sb;
        std::memcpy(&sb, &scaled, 4);
        q = truncX86(sb) >> 11;
    } else {
        q = truncX86(bits);
    }
    if (size < 1) return 0;
    q %= size;
    return (int32_t)(q < 0 ? q + size : q);
}

namespace {
// Which TRAM reads can be issued one sample ahead, where, and for which cursor distances that is unsafe (fx_xlate.hpp HoistPlan)
HoistPlan planHoist(const std::vector<MicroOp>& steady, const std::vector<MicroOp>& last, const XlateProgram& p) {
    HoistPlan H;
    if (!p.uniformCursors) return H;
    if (const char* knob = FX_DIAG_KNOB("FX_XLATE_HOIST"))  // diagnostics: 0 = delay-line reads stay in place
        if (std::atoi(knob) == 0) return H;
    auto isRead = [](uint32_t slot) { return slot == AS_TRAM_IR || slot == AS_TRAM_XR; };
    auto isWrite = [](uint32_t slot) { return slot == AS_TRAM_IW || slot == AS_TRAM_XW; };
    auto tramOf = [](uint32_t slot) { return (slot == AS_TRAM_IR || slot == AS_TRAM_IW) ? 0 : 1; };
    auto sizeOf = [&](int t) { return t == 0 ? p.iSize : p.xSize; };
    auto offsetOf = [&](const MicroOp& r) {
        const int32_t size = sizeOf(tramOf(r.w[0]));
        if (p.tramDane) return danePosition(r.w[4], (r.w[6] & 32u) != 0, size);
        int32_t q = truncX86(r.w[4]);
        q = q > size - 1 ? size - 1 : q;
        return q < 0 ? 0 : q;
    };
    // leading reads: the maximal prefix of offset-0 reads into distinct rows, the same in both streams
    int lead = 0;
    std::vector<uint32_t> rows;
    while ((size_t)lead < steady.size() && (size_t)lead < last.size() && lead < 32) {
        const MicroOp& r = steady[(size_t)lead];
        if (!isRead(r.w[0]) || last[(size_t)lead].w[0] != r.w[0] || last[(size_t)lead].w[5] != r.w[5] || last[(size_t)lead].w[4] != r.w[4]) break;
        if (!(r.w[6] & 4u)) break;   // a per-instance position: gathered where it stands
        if (p.tramDane && (r.w[6] & 64u)) {   // an interpolated read between two slots: likewise
            float f;
            std::memcpy(&f, &r.w[4], 4);
            const float scaled = f * 2147483648.0f;
            uint32_t sb;
            std::memcpy(&sb, &scaled, 4);
            if (truncX86(sb) & 0x7ff) break;
        }
        if (sizeOf(tramOf(r.w[0])) < 1 || (!p.tramDane && offsetOf(r) != 0) || std::find(rows.begin(), rows.end(), r.w[5]) != rows.end()) break;
        // a register with a control track takes its scheduled value at the head of the sample, BEFORE the program's first
        // instruction: a read into it must stay an ordinary instruction behind the head (api fuzz seed 50788)
        if (std::find(p.trackRows.begin(), p.trackRows.end(), (int)r.w[5]) != p.trackRows.end()) break;
        rows.push_back(r.w[5]);
        ++lead;
    }
    if (lead == 0) return H;
    for (const MicroOp& r : steady)   // a write whose slot is not known here may be the one an early read must not overtake
        if (isWrite(r.w[0]) && !(r.w[6] & 4u)) return H;
    // per TRAM: reads and writes per sample, leading reads
    int nRead[2] = {0, 0}, nWrite[2] = {0, 0}, nLead[2] = {0, 0};
    for (size_t i = 0; i < steady.size(); ++i) {
        const uint32_t slot = steady[i].w[0];
        if (isRead(slot)) ++nRead[tramOf(slot)];
        if (isWrite(slot)) ++nWrite[tramOf(slot)];
        if ((int)i < lead) ++nLead[tramOf(slot)];
    }
    // the hoist point: behind the last record that touches a destination row of a leading read, behind every other read
    // of a TRAM that has leading reads (the cursor order of its reads must not change), and outside every SKIP shadow
    auto touches = [&](const MicroOp& r) {
        const uint32_t slot = r.w[0];
        if (slot == AS_NOP || slot == AS_PRED || slot == AS_UNPRED || slot == AS_ENDSAMPLE) return false;
        uint32_t kind;
        if (slot >= AS_MACS) kind = ((slot - AS_MACS) % 16) / 2;
        else kind = r.w[6] & 7u;
        for (uint32_t row : rows) {
            if (!(kind & 1u) && r.w[2] == row) return true;
            if (!(kind & 2u) && r.w[3] == row) return true;
            if (!(kind & 4u) && r.w[4] == row) return true;
            if (r.w[5] == row) return true;
        }
        return false;
    };
    int at = lead - 1;
    for (size_t i = (size_t)lead; i < steady.size(); ++i) {
        const uint32_t slot = steady[i].w[0];
        if (slot == AS_ENDSAMPLE) break;
        if (touches(steady[i]) || (isRead(slot) && nLead[tramOf(slot)] > 0)) at = (int)i;
    }
    {
        bool shadow = false;
        for (size_t i = 0; i < steady.size() && steady[i].w[0] != AS_ENDSAMPLE; ++i) {
            const uint32_t slot = steady[i].w[0];
            if (slot == AS_PRED) shadow = true;
            else if (slot == AS_UNPRED) shadow = false;
            // a SKIP (or the instruction it is fused with) restricts EXEC before the first PRED of its shadow: never stop
            // between an instruction and the SKIP / PRED that follows it
            const uint32_t next = i + 1 < steady.size() ? steady[i + 1].w[0] : (uint32_t)AS_ENDSAMPLE;
            if ((int)i >= at && !shadow && slot != AS_SKIP && next != AS_SKIP && next != AS_PRED && next != AS_NOP) { at = (int)i; break; }
            if (next == AS_ENDSAMPLE) { at = (int)i; if (shadow) return H; break; }
        }
    }
    // writes the early reads overtake: those behind the hoist point
    if (p.tramDane) {
        // DANE model: next sample's read of position q uses slot (counter - 1 + q); a write of this sample at position pw
        // uses (counter + pw): the same slot iff pw == q - 1 (mod size) - all constants, decided here
        for (int k = 0; k < lead; ++k) {
            const int t = tramOf(steady[(size_t)k].w[0]);
            const int64_t size = sizeOf(t), q = offsetOf(steady[(size_t)k]);
            for (size_t i = (size_t)at + 1; i < steady.size(); ++i)
                if (isWrite(steady[i].w[0]) && tramOf(steady[i].w[0]) == t && offsetOf(steady[i]) == (q - 1 + size) % size) return H;
        }
    }
    for (int t = 0; t < 2 && !p.tramDane; ++t) {
        if (nLead[t] == 0) continue;
        std::vector<int> later;  // index among the TRAM's writes of a sample
        int w = 0;
        bool plainOffsets = true;
        for (size_t i = 0; i < steady.size(); ++i) {
            if (!isWrite(steady[i].w[0]) || tramOf(steady[i].w[0]) != t) continue;
            if ((int)i > at) { later.push_back(w); plainOffsets = plainOffsets && offsetOf(steady[i]) == 0; }
            ++w;
        }
        if (later.empty()) continue;
        if (nRead[t] != nWrite[t] || !plainOffsets) return H;  // the cursor distance drifts, or slots are offset: reads stay in place
        const int64_t size = sizeOf(t);
        for (int ir = 0; ir < nLead[t]; ++ir)
            for (int jw : later) {
                const uint32_t d = (uint32_t)((((int64_t)jw - nRead[t] - ir) % size + size) % size);
                if (std::find(H.forbidden[t].begin(), H.forbidden[t].end(), d) == H.forbidden[t].end()) H.forbidden[t].push_back(d);
            }
        if (H.forbidden[t].size() > 64) { H.forbidden[0].clear(); H.forbidden[1].clear(); return H; }
    }
    H.leadCount = lead;
    H.hoistAfter = at;
    for (size_t i = (size_t)at + 1; i < steady.size(); ++i) {
        const uint32_t slot = steady[i].w[0];
        if ((isRead(slot) || isWrite(slot)) && sizeOf(tramOf(slot)) >= 1) ++H.vmemAfterHoist;
    }
    return H;
}
}  // namespace

XlateProgram xlateProgramOf(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords, int iSize, int xSize,
                            int nRows, const std::vector<int>& inputRows, const std::vector<int>& latchRows, const std::vector<int>& trackRows) {
    XlateProgram p;
    p.trackRows = trackRows;
    p.iSize = iSize;
    p.xSize = xSize;
    p.inRows = inputRows;
    p.latchRows = latchRows;
    // uniform cursors: no TRAM instruction inside a SKIP shadow (every lane executes every one of them, so all
    // lanes' cursors move together) and every TRAM offset operand uniform
    bool any = false, ok = true;
    for (const std::vector<MicroOp>* recs : {&steadyRecords, &lastRecords}) {
        bool shadow = false;
        for (const MicroOp& r : *recs) {
            const uint32_t slot = r.w[0];
            if (slot == AS_PRED) shadow = true;
            else if (slot == AS_UNPRED) shadow = false;
            else if (slot >= AS_TRAM_IR && slot <= AS_TRAM_XW) {
                any = true;
                if (r.w[6] & 16u) p.tramDane = true;  // (all TRAM records of a program carry the flag, or none)
                // reference model: a shadowed TRAM instruction makes the lanes' cursors diverge; DANE model: the counter
                // steps per sample whatever executes, so only the position has to be uniform
                // ... a per-instance position is gathered per lane (generated for DANE addresses, daneGather)
                if ((shadow && !(r.w[6] & 16u)) || (!(r.w[6] & 4u) && (r.w[6] & 48u) != 48u)) ok = false;
            }
        }
    }
    p.uniformCursors = any && ok;
    if (p.uniformCursors) {
        for (const MicroOp& r : steadyRecords)
            if (r.w[0] >= AS_TRAM_IR && r.w[0] <= AS_TRAM_XW && ((r.w[0] == AS_TRAM_IR || r.w[0] == AS_TRAM_IW) ? iSize : xSize) >= 1) ++p.tramOpsInline;
        p.hoist = planHoist(steadyRecords, lastRecords, p);
    }

    // LOG/EXP tables the inline code uses (per-lane operand): up to 4 of them go to LDS (1 KB each per wavefront, + 1 KB shared)
    for (const std::vector<MicroOp>* recs : {&steadyRecords, &lastRecords})
        for (const MicroOp& r : *recs)
            if (r.w[0] == AS_LUT && !(r.w[6] & 1u) && (r.w[6] & 2u) && std::find(p.lutTables.begin(), p.lutTables.end(), r.w[3]) == p.lutTables.end())
                p.lutTables.push_back(r.w[3]);
    if (p.lutTables.size() > 4) p.lutTables.clear();

    // Row classes.  BOUNDED: every value the row can hold lies in [-1, 1] - its writers saturate, or pass a bounded
    // value on - given that it started there (the template checks the state rows of this class, and inline TRAM
    // reads into them, against 1.0).  WILD: anything else: CCR, PCM input rows, results of the wrap-around and
    // integer instructions, of a per-lane TRAM handler, or copies of wild values.  Optimistic fixpoint.
    p.wildRow.assign((size_t)std::max(nRows, 1), 0);
    auto wild = [&](uint32_t row) { return row >= p.wildRow.size() || p.wildRow[row] != 0; };
    auto big = [](uint32_t bits) {
        float f;
        std::memcpy(&f, &bits, 4);
        return !(std::fabs(f) <= 1.0f);
    };
    auto operandWild = [&](uint32_t word, bool uniform) { return uniform ? big(word) : wild(word); };
    p.wildRow[0] = 1;  // CCR holds 0, 2, 6, 8, 16, 20
    for (int r : inputRows)
        if (r >= 0 && (size_t)r < p.wildRow.size()) p.wildRow[(size_t)r] = 1;
    for (int r : trackRows)  // a schedule may hold any finite value
        if (r >= 0 && (size_t)r < p.wildRow.size()) p.wildRow[(size_t)r] = 1;
    for (bool changed = true; changed;) {
        changed = false;
        for (const std::vector<MicroOp>* recs : {&steadyRecords, &lastRecords}) {
            for (const MicroOp& r : *recs) {
                const uint32_t slot = r.w[0], dst = r.w[5];
                const bool uA = r.w[6] & 1u, uX = r.w[6] & 2u, uY = r.w[6] & 4u;
                bool makesWild = false;
                switch (slot) {
                    case AS_MOV: makesWild = operandWild(r.w[2], uA); break;
                    case AS_LIMIT:
                    case AS_LIMITN: makesWild = operandWild(r.w[3], uX) || operandWild(r.w[4], uY); break;
                    case AS_TSTNEG: makesWild = operandWild(r.w[3], uX); break;  // X or (~X scaled back): inside [-1, 1] when X is
                    case AS_LUT: makesWild = operandWild(r.w[2], uA); break;     // tables map [-1, 1] into [-1, 1]
                    case AS_MACW:
                    case AS_MACWN:
                    case AS_MACINTW:
                    case AS_ANDXOR: makesWild = true; break;
                    case AS_TRAM_IR:
                    case AS_TRAM_XR: makesWild = !p.uniformCursors; break;  // the inline read is checked, the handler's is not
                    default: continue;  // saturating families, NOISE (|n| <= 1), instructions without a result
                }
                if (makesWild && dst < p.wildRow.size() && !p.wildRow[dst]) {
                    p.wildRow[dst] = 1;
                    changed = true;
                }
            }
        }
    }
    return p;
}

bool planXlate(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords, const XlateTemplate& tmpl,
               const XlateProgram& program, XlateImage* out, std::vector<uint32_t> code[5], std::string listing[5], std::string* err) {
    // hole: [steady fast][last fast][steady exact][last exact][run-once], each on a cache line.  The steady streams
    // branch to their last-sample streams, the fast streams to the exact ones; sizes do not depend on the targets, so:
    // size everything with dummy targets, lay out, translate the exact streams (their sync points are the fast
    // streams' escape targets), then the fast ones.
    const std::vector<MicroOp>* recs[2] = {&steadyRecords, &lastRecords};
    // VGPRs above the register file: the four constants of the quick LOG/EXP index guess (Translator::lut), when there is room
    XlateProgram pooledProgram = program;
    pooledProgram.vconst.clear();
    {
        const char* knob = FX_DIAG_KNOB("FX_XLATE_VCONST");  // diagnostics: 0 = none
        const int firstFree = kRegFileBase + (int)program.wildRow.size();
        if (!program.lutTables.empty() && tmpl.vgprs - firstFree >= 4 && !(knob && std::atoi(knob) == 0)) {
            int v = tmpl.vgprs;
            for (uint32_t c : {lutGuess().scale, lutGuess().bias, lutGuess().magic, lutGuess().mask}) pooledProgram.vconst.emplace_back(c, --v);
        }
    }
    // ... a staged program's spare registers for the next sample's packet and for its PCM input bursts (StageInfo)
    int stageTop = tmpl.vgprs - (int)pooledProgram.vconst.size();
    if (program.stage.count > 1) {
        const int firstFree = kRegFileBase + (int)program.wildRow.size();
        const int need = (int)program.stage.recvRows.size();
        if (stageTop - need < firstFree) { if (err) *err = "no spare VGPRs for the stage's packets in this build"; return false; }
        stageTop -= need;
        pooledProgram.stage.recvTmp = stageTop;
        if (program.stage.inRing == -2) {
            int used = 0;
            for (int r : program.inRows) used += r >= 0;
            const int ringRegs = 2 * kInputBurst * used;
            pooledProgram.stage.inRing = -1;
            if (used > 0 && stageTop - ringRegs >= firstFree) { stageTop -= ringRegs; pooledProgram.stage.inRing = stageTop; }
        }
    }
    // ... and the product cache of the fast streams (Translator::product): per entry an even-aligned pair and a single
    {
        const char* knob = FX_DIAG_KNOB("FX_XLATE_CSE");  // diagnostics: number of entries (0 = none)
        const int top = stageTop;
        const int base = (kRegFileBase + (int)program.wildRow.size() + 1) & ~1;
        int entries = std::min(kProductCacheEntries, knob ? std::atoi(knob) : kProductCacheEntries);
        while (entries > 0 && base + 3 * entries > top) --entries;
        pooledProgram.cseBase = base;
        pooledProgram.cseEntries = entries;
    }
    XlateProgram prog[2] = {pooledProgram, pooledProgram};
    uint32_t bytes[4] = {0, 0, 0, 0};  // steady fast, steady exact, last fast, last exact
    bool fastOk = true;
    for (int k = 0; k < 2; ++k) {
        std::vector<uint32_t> scratch, dummy(4 * recs[k]->size() + 4, tmpl.holeOff);
        XlateStats st;
        if (!translateStream(*recs[k], tmpl, prog[k], tmpl.holeOff, k == 1, tmpl.holeOff, &dummy, &scratch, nullptr, &st, nullptr, nullptr, err)) return false;
        bytes[2 * k] = align64((uint32_t)scratch.size() * 4);
        fastOk = fastOk && !st.nonFiniteImmediate;
        if (!translateStream(*recs[k], tmpl, prog[k], tmpl.holeOff, k == 1, tmpl.holeOff, nullptr, &scratch, nullptr, &st, nullptr, nullptr, err)) return false;
        bytes[2 * k + 1] = align64((uint32_t)scratch.size() * 4);
    }
    uint32_t at = tmpl.holeOff;
    if (fastOk) { out->base[0] = at; at += bytes[0]; out->base[2] = at; at += bytes[2]; }
    out->base[1] = at; at += bytes[1];
    out->base[3] = at; at += bytes[3];
    if (!fastOk) { out->base[0] = out->base[1]; out->base[2] = out->base[3]; }
    std::vector<uint32_t> exactRet[2];
    XlateStats stats[4];
    uint32_t cold[4] = {0, 0, 0, 0};
    for (int k = 1; k >= 0; --k)
        if (!translateStream(*recs[k], tmpl, prog[k], out->base[2 * k + 1], k == 1, out->base[3], nullptr, &code[2 * k + 1],
                             listing ? &listing[2 * k + 1] : nullptr, &stats[2 * k + 1], &exactRet[k], &cold[2 * k + 1], err))
            return false;
    for (int k = 1; k >= 0; --k) {
        code[2 * k].clear();  // (a non-finite uniform operand: every wave runs the exact streams)
        if (!fastOk) { cold[2 * k] = cold[2 * k + 1]; stats[2 * k] = stats[2 * k + 1]; continue; }
        if (!translateStream(*recs[k], tmpl, prog[k], out->base[2 * k], k == 1, out->base[2], &exactRet[k], &code[2 * k],
                             listing ? &listing[2 * k] : nullptr, &stats[2 * k], nullptr, &cold[2 * k], err))
            return false;
        if (align64((uint32_t)code[2 * k].size() * 4) != bytes[2 * k]) { if (err) *err = "internal: fast stream changed size"; return false; }
    }
    for (int k = 0; k < 2; ++k)
        if (align64((uint32_t)code[2 * k + 1].size() * 4) != bytes[2 * k + 1]) { if (err) *err = "internal: exact stream changed size"; return false; }
    out->steadyFastOff = cold[0];
    out->steadyOff = cold[1];
    out->lastFastOff = cold[2];
    out->lastOff = cold[3];
    out->steady = stats[0];
    out->last = stats[2];
    out->wildRow = program.wildRow;
    out->vgprConstants = (int)pooledProgram.vconst.size();
    out->initOff = 0;
    out->ldsBytes = 0;
    code[4].clear();
    // time-sliced priorities: log2(ticks of a sample period / 24) from the modelled issue time of the steady stream on a SIMD
    // with four wavefronts at 2.2 GHz (100 MHz ticks: / 22)
    int sliceBias = -1;
    if (program.prioritySlices) {
        const double ticks = std::max(1.0, (double)stats[0].valuClocks * 4.0 / 22.0 / 24.0);
        sliceBias = std::max(0, (int)std::lround(std::log2(ticks)));
    }
    if (!program.lutTables.empty() || program.hoist.leadCount > 0 || sliceBias >= 0) {
        emitInit(program, &code[4], listing ? &listing[4] : nullptr, sliceBias);
        out->initOff = at;
        out->ldsBytes = program.lutTables.empty() ? 0 : lutLds().bytes(program.lutTables.size());
        at += align64((uint32_t)code[4].size() * 4);
    }
    out->codeBytes = at - tmpl.holeOff;
    if (out->codeBytes + 4 > tmpl.holeBytes) {
        if (err) *err = "translated program larger than the code hole of the template";
        return false;
    }
    return true;
}

}  // namespace fx
