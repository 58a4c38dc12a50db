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

#include <array>
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

// Software pipelining of delay-line reads.  The TRAM reads a program starts with ("leading reads": records
// [0, leadCount) of both streams, offset 0, uniform cursors) are issued one sample period AHEAD - in the steady
// stream, right after the last instruction that touches their destination rows - so that their HBM latency
// hides behind the rest of the current sample instead of stalling the next one.  A read issued early overtakes
// the writes that follow it in the current sample; it would see stale data if one of them hit the same slot.
// With as many reads as writes per sample the distance (read cursor - write cursor) mod size is the same at
// every sample start, so that is decided once per launch: run-once code compares it with `forbidden` and keeps
// the reads in place (s95 = 0) when a pair would collide.
struct HoistPlan {
    int leadCount = 0;    // 0 = no read is issued ahead
    int hoistAfter = -1;  // steady stream: the next sample's leading reads follow this record
    int vmemAfterHoist = 0;  // inline TRAM instructions of the steady stream behind the hoist point
    std::vector<uint32_t> forbidden[2];  // [iTRAM, xTRAM]: cursor distances for which a hoisted read and a later write share a slot
};

// Control tracks (the reference's caller changes a slider every 8 samples, source/main.cpp:107-114):
// registers can be given a schedule of values that the generated sample loop applies itself - at sample s of a block, when s
// is a multiple of the track's period, the register takes the next value - instead of the caller cutting the block.  The
// generated code only knows WHICH rows are trackable (at most kMaxTracks: each has its few instructions behind the loop);
// the schedules of a block come as ONE list of events, sorted by sample, in a device buffer (AsmArgs.tracks), so arming,
// changing or clearing a schedule needs no re-translation and the loop itself pays one scalar compare per sample whatever
// the number of schedules.  Buffer: the events of the block in order, a closing record with sample 0xFFFFFFFF, then the
// values: one float per step (for all instances) or nPad floats per step (one per instance).
constexpr int kMaxTracks = 16;
struct TrackEvent {
    uint32_t sample;       // the sample of the block at which the register changes (0xFFFFFFFF closes the list)
    uint32_t slot;         // which trackable row (index into XlateProgram::trackRows)
    uint32_t valueOffset;  // byte offset of the value (or of the row of per-instance values) from the start of the buffer
    uint32_t strideBytes;  // 4: one value for all instances (scalar load); else the values are a row, one per instance
};

// A program pipelined over the wavefronts of a workgroup ("stages", SURVEY.md section 8d: small batches leave the machine
// empty - 4096 instances are 64 wavefronts on 1024 SIMDs - and a lone wavefront issues one instruction every ~4.5 clocks
// whatever it is).  The instances offer no more lanes, but a program whose state registers are each read and written by one
// stretch of it can be CUT into K contiguous stages: wavefront k of a K-wavefront workgroup runs stage k, all K on the same 64
// instances, stage k working on sample T - k at step T; the values that are live across a cut travel through double-buffered
// LDS rows, one s_barrier per step.  Every wavefront keeps a full private copy of the register file in its VGPRs; a row is
// stored at the end of the block by the stage that writes it last.  planStages() decides where cuts are legal:
//   * a read whose reaching definition is the PREVIOUS sample's must sit in the stage of that definition (a later stage's
//     result of sample t-1 does not exist yet when an earlier stage starts sample t);
//   * a SKIP, its shadow and the instruction it is fused with stay in one stage; so do all delay-line and noise instructions
//     (stage 0: cursors and LFSR words are per-wavefront state), and programs with control tracks or the DANE model are not cut.
struct StageInfo {
    int index = 0, count = 1;             // this stream is stage `index` of `count`; count == 1: the whole program, no pipeline
    std::vector<int> recvRows, sendRows;  // rows received from stage index-1 at the head / handed to stage index+1 at the tail
    uint32_t recvOff = 0, sendOff = 0;    // LDS byte offset of those packets inside a buffer (row i at +256 i)
    uint32_t bufBase = 0, bufStride = 0;  // the 4 * group buffers: bufBase + (sample mod (4 * group)) * bufStride (stride a power of two)
    uint32_t flagBase = 0;                // LDS rows [stage][lane]: non-zero = that stage runs its exact stream (its packets may hold non-finite values)
    // A stage's ring pointer (kVRing) is lane * 4 + ptrBias + buffer * bufStride, and its packet accesses are offsets from there.
    // Where the ring lies at LDS address 0 (programs without LOG / EXP tables in LDS: stageLdsLayout), ptrBias = the byte offset
    // of the stage's received rows inside a buffer (stage 0: of the rows it sends): the first rows of a packet are then within
    // the 255-dword reach of ds_read2_b32 / ds_write2_b32 and go two per instruction.
    uint32_t ptrBias = 0;
    uint32_t storeMask = ~0u;             // channels whose PCM output this stage stores
    int group = 1;                        // samples between two barriers (a power of two); the ring has 4 * group buffers
    // Latencies stay off the step (a step = the `group` samples between two barriers): stage k runs 3k steps behind stage 0.
    // A packet written during step T is waited for at the end of step T+1 (s_waitcnt in front of that step's barrier: long
    // complete by then) and may be READ from step T+2 on; the consumer works on it in step T+3 and requests every packet one
    // SAMPLE ahead, into spare VGPRs (recvTmp ..), from where the next sample's head moves it into the rows.
    int recvTmp = -1;                     // first of recvRows.size() spare VGPRs
    // PCM input of a stage whose steps are shorter than a trip to memory: 8 samples per burst, two bursts in VGPRs (per used
    // channel inRing .. +7 = the samples being consumed, +8 .. +15 = the next eight, in flight), selected by VGPR index mode
    int inRing = -1;                      // first VGPR of the ring, -1 = the one-sample-ahead prefetch of the unstaged loop
};
// `group` samples between two barriers: a wavefront meets the others only every fourth sample (a barrier costs a lone
// wavefront ~200 clocks, arrival skew included); a packet is consumed at most three barrier intervals after it was written
constexpr int kStageDepth = 3, kStageGroupMax = 8, kInputBurst = 8;   // (group: StageInfo::group, the largest of 8, 4, 2, 1 whose ring fits the LDS)
struct StagePlan {
    std::vector<int> cuts;                    // record index at which stage k+1 begins (size K-1, ascending); empty: not cut
    std::vector<std::vector<int>> live;       // live[c]: rows handed over at cut c
    std::vector<int> storeStage;              // per register-file row: the stage that stores it at the end of a block
    std::vector<int> pcmStage;                // per channel: the stage that stores its PCM output
    std::vector<uint32_t> inMask;             // per stage: channels whose PCM input it reads
    std::string why;                          // why the program is not cut (diagnostics)
    // what the planner expects: cost of the whole program and of each stage (pipeline overhead included) in the units of its
    // balance - roughly vector instructions per sample - and the LOG / EXP round trips in them (Batch::chooseStages)
    int totalCost = 0, totalLuts = 0;
    std::vector<int> stageCost, stageLuts;
};

// What the translator needs to know about the program beyond its records.
struct XlateProgram {
    StageInfo stage;
    int iSize = 0, xSize = 0;     // itramsize / xtramsize (the cursors' modulus)
    bool uniformCursors = false;  // all lanes' TRAM cursors move together: kept in SGPRs, TRAM instructions inline
    bool tramDane = false;        // opt-in DANE delay-line model: s80 / s82 are per-sample address counters, taps at (counter + position) mod size
    // LOG/EXP tables in LDS: the fp32 thresholds, x1[] and the {slope, y1} arrays of the tables the program uses
    // (lutTables = their byte offsets in the LUT blob, in LDS order); empty = tables are read from global memory
    std::vector<uint32_t> lutTables;
    std::vector<uint8_t> wildRow; // per register-file row: 0 = BOUNDED class (always inside [-1, 1]), 1 = WILD
    // PCM I/O of the generated sample loop: row the input of channel c is copied to (-1: unused), row of its output latch
    std::vector<int> inRows, latchRows;
    int tramOpsInline = 0;        // inline TRAM instructions per sample of the steady stream (each issues one VMEM operation)
    std::vector<int> trackRows;   // register-file row of track slot t (at most kMaxTracks); such rows are of the WILD class
    HoistPlan hoist;
    // delay lines far larger than the caches (set by the batch from slots x instances): TRAM loads and stores carry the
    // non-temporal hint - every slot is written once and read once, a whole delay later (+2 % at the memory-bound probe)
    bool tramStreaming = false;
    // unstaged programs: the wavefronts of a SIMD take turns at the top priority by the clock (fx_xlate.cpp Translator::run) - for
    // batches of two or more wavefronts per SIMD, which run on a build of at most four wave slots (four priority levels)
    bool prioritySlices = false;
    // uniform constants kept in VGPRs above the register file for the whole launch: (bit pattern, VGPR), set by planXlate
    // (the constants of the LOG/EXP index guess, which must be VGPR sources to stay in the double-rate instruction class)
    std::vector<std::pair<uint32_t, int>> vconst;
    // product cache of the fast streams (fx_xlate.cpp Translator::product): cseEntries entries in the spare VGPRs from cseBase
    // (even) on - the fp64 pairs first, then the fp32 products; set by planXlate
    int cseBase = 0, cseEntries = 0;
};
constexpr int kMaxVgprConstants = 4;
constexpr int kProductCacheEntries = 2;
constexpr int kSpareVgprsWanted = kMaxVgprConstants + 3 * kProductCacheEntries + 1;  // what a VGPR build with room to spare is chosen for
// nRows = rows of the register file, inputRows = the rows the PCM input goes to (-1: unused channel), latchRows = the
// rows the PCM output comes from; one entry per channel
XlateProgram xlateProgramOf(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords, int iSize, int xSize,
                            int nRows, const std::vector<int>& inputRows, const std::vector<int>& latchRows,
                            const std::vector<int>& trackRows = std::vector<int>());

struct XlateStats {
    int inlined = 0;     // records translated to straight-line code
    int called = 0;      // records executed by a call to the interpreter's handler
    int instructions = 0;
    int valu = 0;        // vector-ALU instructions a finite, in-domain wave executes per sample period
    int valuSlow = 0;    // ... of which of the ~4-clock class (fp64 arithmetic and conversions, compares, shifts, selects)
    int valuClocks = 0;  // ... and their modelled issue time on a busy SIMD, clocks (cost table in fx_xlate.cpp Emitter::tally)
    int fusedSkips = 0;  // SKIPs translated as a predicate on the value that would have set their CCR
    int regions = 0;     // SKIP shadows run under one EXEC mask (no per-instruction PRED)
    int unitMultipliers = 0;  // multiplications by +-1.0 that were not emitted
    int reusedProducts = 0;   // products of a uniform multiplier and a row taken from the product cache
    int fusedZeroAdds = 0;    // "R = 0 + X * c", |c| > 0.5, emitted as one fma (bit-identical, see fx_xlate.cpp zeroPlusScaled)
    int deadResults = 0; // instructions whose result nothing reads (only their CCR, if anything, is computed)
    int unsaturated = 0; // saturating instructions whose result provably lies in [-1, 1]: no v_med3 in the fast stream
    bool nonFiniteImmediate = false;  // a NaN / Inf among the uniform operands: no fast stream for this program
};

// Translate one stream of records (encodeAsmStream(ops, nullptr, true): w0 = handler slot) into the code of a
// whole sample LOOP that starts `codeBase` bytes after the kernel entry: PCM input of the sample (prefetched one
// sample ahead), the program, PCM output, pointer advance and the branch back.  A steady stream loops while the
// next sample is not the block's last and then continues at `nextBase` (the hot entry of the matching last-sample
// stream); a last-sample stream (isLast) runs once and leaves through s[34:35] (the template's epilogue).  Entry
// points: offset 0 ("hot": from the previous sample) and *coldEntry ("cold": from the template; sets up the
// scalar TRAM cursors first).
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
                     bool isLast, uint32_t nextBase, const std::vector<uint32_t>* exactReturns, std::vector<uint32_t>* code,
                     std::string* listing, XlateStats* stats, std::vector<uint32_t>* returns, uint32_t* coldEntry, std::string* err);

// where cuts are legal and which rows cross them; `wanted` stages at most (>= 2), balanced by an estimate of each record's
// vector instructions.  cuts empty = the program runs as one stage (why says why).
StagePlan planStages(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords, const XlateProgram& prog, int nRows, int wanted);

// where a staged program's flag rows, packet ring and scratch lie in LDS, and how many samples lie between two barriers (the
// largest group of 8, 4, 2, 1 - at most maxGroup - whose ring of 4 * group buffers fits the budget); false: not even one fits
struct StageLds {
    std::vector<uint32_t> cutOff;   // byte offset of cut c's rows inside a buffer
    uint32_t bufStride = 0, flagBase = 0, bufBase = 0, scratchBytes = 0, scratchOff = 0, bytes = 0;
    int group = 1;
    bool ringFirst = false;         // [ring][flags][scratch] instead of [tables][flags][ring][scratch]
};
// pinGroup: 1, 2 or 4 = exactly that many samples between two barriers, whatever would fit (the FX_STAGES_GROUP knob: tests of short rings)
bool stageLdsLayout(const XlateProgram& program, const StagePlan& plan, uint32_t ldsBudget, int maxGroup, StageLds* out, int pinGroup = 0);

// what the template needs per stage (fx_interp_gfx950.S, KA_STAGES): 32 bytes each
struct StageDescriptor {
    uint32_t steadyFast, steadyExact, lastFast, lastExact;  // cold entries, byte offsets from the kernel entry
    uint32_t storeFirst, storeCount;                        // this stage's slice of the store-row table (entries behind the load rows)
    uint32_t scratchOff;                                    // LDS bytes [scratchOff, scratchOff + 512 * stages): the epilogue's, nothing else's (a stage
                                                            // that is done reaches its epilogue while later stages still read tables and packets)
    uint32_t reserved1;
};

// A loadable code object: the template with the four streams in its hole.
struct XlateImage {
    int stages = 1;                            // > 1: launch with 64 * stages threads per workgroup
    int slowestStageValu = 0;                  // stages > 1: vector instructions per sample of the stage that sets the pace (steady / last: sums over the stages)
    std::vector<StageDescriptor> stageDesc;    // stages > 1
    std::vector<std::vector<int>> stageStoreRows;  // stages > 1: register-file rows stage k stores at the end of a block
    std::vector<std::array<uint32_t, 4>> stageBases;  // stages > 1: where stage k's four streams start
    StagePlan plan;
    std::vector<unsigned char> elf;
    // COLD entry offsets from the kernel entry; AsmArgs.steady = steadyFastOff | steadyOff << 32, .last likewise
    uint32_t steadyFastOff = 0, steadyOff = 0, lastFastOff = 0, lastOff = 0;
    uint32_t base[4] = {0, 0, 0, 0};  // where the streams start: steady fast, steady exact, last fast, last exact
    uint32_t initOff = 0;   // run-once code (AsmArgs.initOff), 0 = none
    uint32_t ldsBytes = 0;  // dynamic LDS per workgroup
    uint32_t codeBytes = 0;
    XlateStats steady, last;  // of the stream a finite wave runs
    std::vector<uint8_t> wildRow;  // row classes the code relies on: the loader flags BOUNDED rows in the row table
    int vgprConstants = 0;         // uniform constants the code keeps in VGPRs above the register file
};
// a fingerprint of the code object (template and generated code): what a profile of a launch is a profile OF (bench.py ties the
// committed hardware-counter passes to it; 63 bits)
uint64_t imageHash(const XlateImage& image);
// Lays the four streams out ([steady fast][steady exact][last fast][last exact]) and translates them; code[k] /
// listing[k] in that order (listing may be nullptr); code[4] = the run-once code (LDS tables), empty when none.  Without a fast stream (non-finite uniform operand) the
// fast offsets equal the exact ones.
bool planXlate(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords, const XlateTemplate& tmpl,
               const XlateProgram& prog, XlateImage* out, std::vector<uint32_t> code[5], std::string listing[5], std::string* err);
bool buildXlateImage(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords,
                     const XlateTemplate& tmpl, const XlateProgram& prog, XlateImage* out, std::string* err);
// The same for a program cut into plan.cuts.size() + 1 stages: per stage the four streams ([fast][last fast][exact][last exact]),
// then the shared run-once code.  code / listing (optional): [stage * 4 + stream], the run-once code last.
bool buildStagedImage(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords, const XlateTemplate& tmpl,
                      const XlateProgram& prog, const StagePlan& plan, XlateImage* out, std::vector<std::vector<uint32_t>>* code,
                      std::vector<std::string>* listing, std::string* err, uint32_t ldsBudget = 144u * 1024u,   // LDS a workgroup may take (several per CU: less)
                      int maxGroup = kStageGroupMax,    // samples between two barriers at most: short blocks want short steps (the pipeline fills and drains in 3 (K - 1) of them)
                      int pinGroup = 0);                // stageLdsLayout

}  // namespace fx
