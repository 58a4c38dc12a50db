// fx_batch.cpp — device state management and launch orchestration for one batch of instances.
//
// Device-resident state (all instance-fastest so a wavefront touches contiguous 256-byte runs):
//   state  [rows][nPad] u32   one row per reference register (index = reference register index),
//                             then output latches, TRAM cursors, LFSR words, ood flags, counter
//   itram  [wave][iSlots][64] f32   reference smallDelayBuffer, include/FX8010.h:210
//   xtram  [wave][xSlots][64] f32   reference largeDelayBuffer, include/FX8010.h:211
//   lut    [64][65] f64             LOG tables 0..31, EXP tables 32..63
//   stream steady | last | row table
// Registers the decoder classifies as uniform live only in hostValue_ (and as immediates in the
// stream); their state rows are refreshed when they turn per-instance.
#include "fx_batch.hpp"

#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "../../include/fx8010_amd.h"
#include "fx_knobs.hpp"

namespace fx {

namespace {
const Luts& sharedLuts() {
    static const Luts l;
    return l;
}
constexpr size_t kScratchBytes = 1 << 16;
inline uint32_t bitsOf(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
}  // namespace

Batch::Batch(int64_t nInstances, int channels, int device) : prog_(channels), knobs_(ReleaseKnobs::fromEnvironment()) {
    if (nInstances < 1) throw std::runtime_error("n_instances must be >= 1");
    if (channels < 1 || channels > kMaxChannels) throw std::runtime_error("num_channels must be 1..4");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        throw std::runtime_error(std::string("no usable HIP device (hipGetDeviceCount: ") + hipGetErrorString(e) + "); this library has no CPU fallback");
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= count) throw std::runtime_error("HIP device ordinal out of range");
    device_ = device;
    n_ = nInstances;
    nPad_ = (nInstances + 255) / 256 * 256;  // whole wavefronts for every K in {1,2,4}
    auto chk = [&](hipError_t r, const char* what) {
        if (r != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(r));
    };
    chk(hipSetDevice(device_), "hipSetDevice");
    chk(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
    chk(hipEventCreate(&ev0_), "hipEventCreate");
    chk(hipEventCreate(&ev1_), "hipEventCreate");
    chk(hipMalloc(reinterpret_cast<void**>(&dLut_), sizeof(double) * kLutBlobDoubles), "hipMalloc lut");
    chk(hipMalloc(reinterpret_cast<void**>(&dScratch_), kScratchBytes), "hipMalloc scratch");
    static const LutDevice lutDev(sharedLuts());
    chk(hipMemcpy(dLut_, lutDev.blob.data(), sizeof(double) * kLutBlobDoubles, hipMemcpyHostToDevice), "lut upload");
}

Batch::~Batch() {
    (void)hipSetDevice(device_);
    drainBuilder(true);
    if (stream_) (void)hipStreamSynchronize(stream_);
    (void)hipFree(dState_);
    (void)hipFree(dITram_);
    (void)hipFree(dXTram_);
    (void)hipFree(dLut_);
    clearCodeCache();
    (void)hipFree(dScratch_);
    (void)hipFree(dTracks_);
    (void)hipFree(dIn_);
    (void)hipFree(dOut_);
#ifdef FX_DIAGNOSTICS
    (void)hipFree(dStamps_);
#endif
    if (hPinIn_) (void)hipHostFree(hPinIn_);
    if (hPinOut_) (void)hipHostFree(hPinOut_);
    for (int k = 0; k < kHostPieces; ++k) {
        if (evIn_[k]) (void)hipEventDestroy(evIn_[k]);
        if (evDone_[k]) (void)hipEventDestroy(evDone_[k]);
    }
    if (copyIn_) (void)hipStreamDestroy(copyIn_);
    if (copyOut_) (void)hipStreamDestroy(copyOut_);
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

// The caller's stream may be gone by the time we need the previous launch to have finished (a torch stream that was
// garbage-collected): wait on the event recorded behind that launch instead of holding the foreign handle.
void Batch::waitLastLaunch() {
    if (launched_) (void)hipEventSynchronize(ev1_);
}

int Batch::fail(int code, const std::string& what) {
    lastError_ = what;
    return code;
}
int Batch::hipFail(hipError_t e, const char* where) {
    // The runtime also keeps the error as the calling thread's "last error", and the launch helpers of fx_kernel.hip report
    // hipGetLastError() after their kernel: an allocation that failed here would come back as the result of the next launch that
    // works (found by tests/test_gpu_boundary.py::test_delay_memory_that_cannot_be_allocated: fxb_ood_flags said ~0 after a
    // recovered out-of-memory).  It has been reported: clear it.
    (void)hipGetLastError();
    return fail(e == hipErrorOutOfMemory ? FX_E_MEMORY : FX_E_NODEVICE, std::string(where) + ": " + hipGetErrorString(e));
}

bool Batch::loadFile(const std::string& path) { drainBuilder(false); return afterLoad(prog_.loadFile(path)) == 1; }
bool Batch::loadText(const std::string& text) { drainBuilder(false); return afterLoad(prog_.loadText(text)) == 1; }

int Batch::setOption(unsigned option, bool on) {
    if (option & ~kOptAll) return -3;
    drainBuilder(false);
    prog_.options = on ? (prog_.options | option) : (prog_.options & ~option);
    lowDirty_ = true;
    return 0;
}

int Batch::afterLoad(bool ok) {
    (void)hipSetDevice(device_);
    // registers may have been created even when the load failed; keep host mirrors in step
    const size_t old = hostValue_.size();
    hostValue_.resize(prog_.regs.size());
    forcedLane_.resize(prog_.regs.size(), 0);
    laneWritten_.resize(prog_.regs.size(), 0);
    for (size_t r = old; r < prog_.regs.size(); ++r) hostValue_[r] = prog_.regs[r].value;
    lowDirty_ = true;
    // code generated for the program as it was is of no use any more (registers and instructions accumulate over loads)
    clearCodeCache();
    ++loadGen_;
    wantedClass_ = -1;
    otherClassBlocks_ = 0;
    for (Tuner& t : tune_) t = Tuner();
    // registers the program itself keeps per-instance (it writes them, reads them from a delay line or the PCM input): a
    // property of the program, valid until the next load
    intrinsicLane_.assign(prog_.regs.size(), 0);
    readByProgram_.assign(prog_.regs.size(), 0);
    declared_.assign(prog_.regs.size(), 0);
    for (const std::string& name : prog_.controls) {
        const int r = prog_.findRegister(name);
        if (r >= 0) declared_[(size_t)r] = 1;
    }
    hotControl_.assign(prog_.regs.size(), 0);   // (nothing has moved since THIS load)
    lastControlWrite_.assign(prog_.regs.size(), 0);
    coolAfter_.assign(prog_.regs.size(), kCoolSamples);
    cooledOnce_.assign(prog_.regs.size(), 0);
    leanActive_ = leanPending_ = false;
    leanStale_ = controlMode_;
    leanKey_.clear();
    leanFolded_.clear();
    leanWant_.clear();
    if (!prog_.instrs.empty()) {   // (also after a load that failed: its registers and instructions have been appended, as in the reference)
        const Lowered probe = lowerProgram(prog_, hostValue_, std::vector<uint8_t>(prog_.regs.size(), 0), 1, false, 1);
        for (size_t r = 0; r < probe.rowOfReg.size() && r < intrinsicLane_.size(); ++r) intrinsicLane_[r] = probe.rowOfReg[r] >= 0;
        for (const Instr& in : prog_.instrs)
            for (int o : {in.a, in.x, in.y})
                if (o >= 0 && (size_t)o < readByProgram_.size()) readByProgram_[(size_t)o] = 1;
    }
    // a load that failed after an earlier good one has still appended registers (literals and declarations are created
    // before the error, as in the reference): the state block must follow, or set_register of a new one would land in
    // the rows behind the registers (output latches, cursors, LFSR, counter)
    if (!ok) {
        if (dState_) (void)ensureState();
        return 0;
    }
    loaded_ = true;
    if (controlMode_) markControls();   // (a further load may have declared more controls)
    if (ensureState() != 0) return 0;
    return 1;
}

int Batch::fillRows(const std::vector<uint32_t>& rows, const std::vector<uint32_t>& values) {
    size_t done = 0;
    const size_t chunk = kScratchBytes / 8;
    while (done < rows.size()) {
        const size_t k = std::min(chunk, rows.size() - done);
        hipError_t e = hipMemcpyAsync(dScratch_, rows.data() + done, k * 4, hipMemcpyHostToDevice, stream_);
        if (e == hipSuccess) e = hipMemcpyAsync(dScratch_ + chunk, values.data() + done, k * 4, hipMemcpyHostToDevice, stream_);
        if (e == hipSuccess) e = launchFillRows(dState_, nPad_, dScratch_, dScratch_ + chunk, (int)k, stream_);
        if (e == hipSuccess) e = hipStreamSynchronize(stream_);  // host vectors are pageable and reused
        if (e != hipSuccess) return hipFail(e, "fillRows");
        done += k;
    }
    return 0;
}

// Allocate the state block, or grow it when a further loadFile() added registers
// (the reference accumulates registers across loads, source/FX8010.cpp:777 ff.).
int Batch::ensureState() {
    const StateLayout want = makeLayout((int)prog_.regs.size(), prog_.numChannels);
    if (dState_ && want.totalRows == stateRows_) return 0;
    uint32_t* fresh = nullptr;
    const size_t rowBytes = (size_t)nPad_ * 4;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&fresh), rowBytes * want.totalRows);
    if (e != hipSuccess) return hipFail(e, "hipMalloc state");
    std::vector<uint32_t> rows, values;
    const uint32_t* old = dState_;
    const StateLayout was = stateLayout_;
    dState_ = fresh;
    int firstNew = 0;
    if (old) {
        waitLastLaunch();
        auto copyRow = [&](int dst, int src) {
            return hipMemcpyAsync(fresh + (size_t)dst * nPad_, old + (size_t)src * nPad_, rowBytes, hipMemcpyDeviceToDevice, stream_);
        };
        for (int r = 0; r < was.nRegs && e == hipSuccess; ++r) e = copyRow(r, r);
        const int specials = was.totalRows - was.outBase;
        for (int k = 0; k < specials && e == hipSuccess; ++k) e = copyRow(want.outBase + k, was.outBase + k);
        if (e == hipSuccess) e = hipStreamSynchronize(stream_);
        (void)hipFree(const_cast<uint32_t*>(old));
        if (e != hipSuccess) return hipFail(e, "state grow");
        firstNew = was.nRegs;
    } else {
        // specials of a fresh batch: latches 0, cursors 0, reference LFSR seeds, flags 0, counter 0
        for (int k = want.outBase; k < want.totalRows; ++k) { rows.push_back(k); values.push_back(0); }
        values[want.noiseBase - want.outBase + 0] = 0x70f4f854u;  // g_x1, include/FX8010.h:290
        values[want.noiseBase - want.outBase + 1] = 0xe1e9f0a7u;  // g_x2, include/FX8010.h:291
    }
    for (int r = firstNew; r < want.nRegs; ++r) { rows.push_back(r); values.push_back(bitsOf(hostValue_[r])); }
    stateLayout_ = want;
    stateRows_ = want.totalRows;
    return fillRows(rows, values);
}

int Batch::ensureTram(const Lowered& low) {
    auto grow = [&](float*& buf, int& have, int want) -> int {
        if (want <= have) return 0;
        size_t waves = (size_t)((n_ + 64 * instPerLane_ - 1) / (64 * instPerLane_));
        const size_t pitch = 256 * (size_t)instPerLane_;  // bytes of one slot of one wavefront
        float* fresh = nullptr;
        const size_t bytes = waves * (size_t)want * pitch;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&fresh), bytes);
        if (e != hipSuccess) return hipFail(e, "hipMalloc TRAM");
        e = hipMemsetAsync(fresh, 0, bytes, stream_);  // the parity domain assumes zeroed delay memory
        if (e == hipSuccess && buf && have > 0) {
            waitLastLaunch();
            e = hipMemcpy2DAsync(fresh, (size_t)want * pitch, buf, (size_t)have * pitch, (size_t)have * pitch, waves, hipMemcpyDeviceToDevice, stream_);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream_);
        if (e != hipSuccess) { (void)hipFree(fresh); return hipFail(e, "TRAM init"); }
        (void)hipFree(buf);
        buf = fresh;
        have = want;
        return 0;
    };
    int rc = grow(dITram_, iSlotsAlloc_, low.iSlots);
    if (rc == 0) rc = grow(dXTram_, xSlotsAlloc_, low.xSlots);
    return rc;
}

// K = instances stepped by one lane.  More instances per lane amortise the scalar fetch/dispatch of
// a record over more work and widen every LDS access, but shrink the number of wavefronts; the LDS
// register file (rows * 256 * K bytes per wavefront) bounds how many wavefronts a CU can hold.
// Once TRAM has been allocated its [wave][slot][64][K] tiling pins K for the life of the batch.
int Batch::chooseInstPerLane() const {
    if (iSlotsAlloc_ > 0 || xSlotsAlloc_ > 0) return instPerLane_;
    if (knobs_.instPerLane) return knobs_.instPerLane;
    const Lowered probe = lowerProgram(prog_, hostValue_, laneForced(), 1);
    if (!probe.error.empty()) return 1;
    for (int k : {4, 2}) {
        const long long waves = (n_ + 64LL * k - 1) / (64LL * k);
        const long long perCu = kLdsBytesPerCU / ((long long)probe.nRows * 256 * k);
        if (waves >= 2048 && perCu >= 4) return k;  // >= 2 wavefronts per SIMD in flight and one per SIMD resident
    }
    return 1;
}

bool Batch::intrinsicLane(int reg) const { return reg >= 0 && (size_t)reg < intrinsicLane_.size() && intrinsicLane_[reg] != 0; }

bool Batch::laneResident(int reg) const {
    if ((reg < (int)forcedLane_.size() && (forcedLane_[reg] || laneWritten_[reg])) || tracked(reg)) return true;
    return !lowDirty_ ? c_.low.rowOfReg[reg] >= 0 : (reg < (int)c_.low.rowOfReg.size() && c_.low.rowOfReg[reg] >= 0);
}

// Small batches leave SIMDs empty (and a lone wavefront issues an instruction only every ~4 clocks): a program that can be cut
// runs as a pipeline of stages over the wavefronts of a workgroup (fx_xlate.hpp StageInfo).  Beyond two wavefronts of instances
// per SIMD the plain program has always been the faster one.  FX_STAGES pins the number asked for (1 = never).
bool Batch::stagingPossibleGiven(bool stagingOff) const {
    if (stagingOff) return false;
    if (knobs_.stages) return knobs_.stages >= 2;
    return (n_ + 63) / 64 < 2048;
}

// How many stages?  The planner's own costs decide (planStages: cost of every stage, pipeline overhead included, in units of
// ~1.4 per vector instruction), with a model of the machine calibrated on tools/stage_policy_probe.sh (profiles/r04_stage_policy*.txt:
// the filter chain, twelve parallel chains with 13-row packets, a delay line + SKIP + LOG / EXP program; 16 .. 2 048 wavefronts):
//   a wavefront alone:   L = 2.85 clocks x cost of the slowest stage + 165 (loop control, PCM) + 100 per LOG / EXP round trip
//                            + 5 per row its packets carry + the barrier: (100 + 15 K) clocks x 1 / 0.42 / 0.1 / 0 per sample for one
//                            every 1 / 2 / 4 / 8 samples (tools/stage_block_probe.py, profiles/r04_stage_block_probe.txt)
//   the CU's issue slots: G workgroups per CU x 2.4 clocks x the cost of ALL stages / (4 SIMDs x 0.8) - for K < 4 the wavefronts of
//                            the G workgroups can pile up on K SIMDs of the CU (as many as the VGPR build lets a SIMD hold)
//   a sample takes the larger of the two; a block also fills and drains the pipeline: 3 (K - 1) steps of `group` samples.
// The options come back cheapest first.  The model is good to ~ 20 % (how the dispatcher spreads workgroups over the CUs is not
// in it), so options within kTuneBand of the best are MEASURED on the caller's own blocks before one is kept (noteLaunchTime).
std::vector<Batch::StageOption> Batch::rankStages(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords,
                                                  const XlateProgram& xprog, int nRows, int blockClass, int wavesPerSimdCap, bool stagingOff) const {
    std::vector<StageOption> out;
    if (stagingOff) { out.push_back(StageOption()); return out; }
    if (knobs_.stages) {
        StageOption o;
        o.wanted = knobs_.stages;
        out.push_back(o);
        return out;
    }
    const double W = (double)((n_ + 63) / 64);
    const int64_t groupsPerCu = std::max<int64_t>(1, ((n_ + 63) / 64 + 255) / 256);
    const uint32_t ldsBudget = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(144 * 1024, 160 * 1024 / groupsPerCu - 256));
    const int maxGroup = blockClass == 0 ? 1 : (blockClass == 1 ? 2 : kStageGroupMax);
    const double blockSamples = blockClass == 0 ? 32.0 : (blockClass == 1 ? 128.0 : 2048.0);
    const double kClocksPerCost = 2.85, kIssuePerCost = 2.4, kFixed = 165.0, kLut = 100.0, kEta = 0.8;
    StageOption plain;
    bool havePlain = false;
    std::vector<int> seen;
    for (int wanted : {8, 4, 2}) {
        if (!stagingPossibleGiven(stagingOff)) break;
        const StagePlan plan = planStages(steadyRecords, lastRecords, xprog, nRows, wanted);
        if (!havePlain && plan.totalCost > 0) {
            plain.wanted = plain.stages = 1;
            const double lone = kClocksPerCost * plan.totalCost + kFixed + kLut * plan.totalLuts;
            const double shared = std::ceil(W / 1024.0) * kIssuePerCost * plan.totalCost / kEta;
            plain.predicted = std::max(lone, shared);
            havePlain = true;
        }
        if (plan.cuts.empty()) continue;
        const int k = (int)plan.cuts.size() + 1;
        if (std::find(seen.begin(), seen.end(), k) != seen.end()) continue;
        seen.push_back(k);
        StageLds lds;
        if (!stageLdsLayout(xprog, plan, ldsBudget, maxGroup, &lds, knobs_.stagesGroup)) continue;
        int worst = 0, luts = 0, sum = 0;
        for (size_t s = 0; s < plan.stageCost.size(); ++s) {
            worst = std::max(worst, plan.stageCost[s]);
            luts = std::max(luts, plan.stageLuts[s]);
            sum += plan.stageCost[s];
        }
        size_t rows = 0;
        for (size_t c2 = 0; c2 < plan.live.size(); ++c2) rows = std::max(rows, plan.live[c2].size() + (c2 + 1 < plan.live.size() ? plan.live[c2 + 1].size() : 0));
        const double barrier = (100.0 + 15.0 * k) * (lds.group >= 8 ? 0.0 : (lds.group == 4 ? 0.1 : (lds.group == 2 ? 0.42 : 1.0)));
        const double lone = kClocksPerCost * worst + kFixed + kLut * luts + 5.0 * (double)rows + barrier;
        const double G = std::ceil(W / 256.0);
        double shared = G * kIssuePerCost * sum / (4.0 * kEta);
        // (K < 4: measured between an even spread and a pile-up of the G workgroups' wavefronts on K SIMDs - the filter chain and the
        // parallel chains in two stages sit near the pile-up, the delay-line program near the even spread: the geometric mean)
        if (k < 4) {
            const double piled = std::min(G, (double)wavesPerSimdCap) * kIssuePerCost * worst / kEta;
            if (piled > std::max(shared, lone)) shared = std::sqrt(std::max(shared, lone) * piled);
        }
        StageOption o;
        o.wanted = wanted;
        o.stages = k;
        o.group = lds.group;
        o.predicted = std::max(lone, shared) * (1.0 + 3.0 * (k - 1) * lds.group / blockSamples);
        out.push_back(o);
    }
    if (havePlain || out.empty()) out.push_back(plain);
    std::stable_sort(out.begin(), out.end(), [](const StageOption& a, const StageOption& b2) { return a.predicted < b2.predicted; });
    return out;
}

// What the generated code is a function of.  Two calls with equal keys would build the same Code, so a finished one is reused
// (ensureLowered): the program (a load counter: registers and instructions only ever accumulate), the options, which registers
// have rows although no instruction writes them (per-instance values, moving controls, control tracks), the values of all the
// others (they are folded into the code as literals), the block-length class staged code is generated for, whether the
// translation is put off because compiled-in controls keep changing, and the diagnostic knobs of the environment.
std::string Batch::codeKey(int blockClass, bool defer) const { return codeKeyFor(laneForced(), blockClass, defer, pickFor(blockClass)); }

std::string Batch::codeKeyFor(const std::vector<uint8_t>& forced, int blockClass, bool defer, int pick) const {
    std::string k;
    auto word = [&](int64_t v) { k.append(reinterpret_cast<const char*>(&v), 8); };
    word(loadGen_); word((int64_t)prog_.options); word(blockClass); word(pick); word(defer ? 1 : 0);
    word(((iSlotsAlloc_ > 0 || xSlotsAlloc_ > 0) && instPerLane_ != 1) ? instPerLane_ : 0);   // delay lines tiled for K instances per lane pin the HIP C++ kernel
    // (the release knobs are fixed for the life of the handle - fx_knobs.hpp ReleaseKnobs - and so not part of the key)
    for (size_t r = 0; r < hostValue_.size(); ++r) {
        const bool f = r < forced.size() && forced[r];
        // (a register no instruction reads as an operand: its value lives in its state row and cannot reach the code)
        const uint32_t w = f ? 0x7fc0f0f0u : (readByProgram((int)r) ? bitsOf(hostValue_[r]) : 0x7fc0f0f1u);
        k.push_back(f ? 1 : 0);
        k.append(reinterpret_cast<const char*>(&w), 4);
    }
    for (int reg : trackRegs_) word(reg);   // (slot order is part of the code)
    return k;
}

void Batch::releaseCode(Code& c) {
    if (c.module) (void)hipModuleUnload(c.module);
    if (c.dStream) (void)hipFree(c.dStream);
    c.module = nullptr;
    c.fn = nullptr;
    c.dStream = nullptr;
    c.streamCap = 0;
}

void Batch::clearCodeCache() {
    (void)hipSetDevice(device_);
    waitLastLaunch();   // the most recent launch may still run one of them
    releaseCode(c_);
    c_ = Code();
    for (std::unique_ptr<Code>& e : cache_) releaseCode(*e);
    cache_.clear();
}

// c_ -> cache_.  The code that is being replaced may still be running: nothing of it is touched; only when the cache is full
// the least recently used entry goes, behind the most recent launch.
void Batch::stashCode() {
    if (c_.key.empty()) {   // nothing finished (a failed build): drop the pieces
        if (c_.module || c_.dStream) { waitLastLaunch(); releaseCode(c_); }
        c_ = Code();
        return;
    }
    c_.lastUse = ++useClock_;
    cache_.push_back(std::make_unique<Code>(std::move(c_)));
    c_ = Code();
    if (cache_.size() > kCodeCache) {
        const size_t lru = lruVictim();
        waitLastLaunch();
        releaseCode(*cache_[lru]);
        cache_.erase(cache_.begin() + (long)lru);
    }
}

bool Batch::cachedCode(const std::string& key) const {
    for (const std::unique_ptr<Code>& e : cache_)
        if (e->key == key) return true;
    return false;
}

bool Batch::adoptCode(const std::string& key) {
    for (size_t k = 0; k < cache_.size(); ++k)
        if (cache_[k]->key == key) {
            c_ = std::move(*cache_[k]);
            cache_.erase(cache_.begin() + (long)k);
            c_.lastUse = ++useClock_;
            return true;
        }
    return false;
}

// staged code is generated for a class of block lengths - when the batch is small enough to be staged at all
int Batch::keyClass() const {
    if (!stagingPossible()) return -1;
    return wantedClass_ >= 0 ? wantedClass_ : stageBlockClass(std::max(pendingSamples_, 1));
}

bool Batch::deferWanted() const {
    // controls that are compiled into the code keep changing (a set_register within the last few blocks): a translation costs
    // a module load (~1-2 ms), a re-encode for the interpreter ~0.05 ms - interpret until they have been quiet.  (A block of
    // more than ~half a millisecond of translated code pays for its translation at once.)
    const double blockMs = (double)n_ * (double)pendingSamples_ * (double)std::max<size_t>(prog_.instrs.size(), 1) / 1e10;
    return controlHeat_ > 0 && blockMs < 0.5 && !(prog_.options & kOptTramDane) && !knobs_.kernelStartsWith("xlate");
}

int Batch::ensureLowered() {
    if (!loaded_ || !prog_.ready) return fail(FX_E_NOTREADY, "no program loaded");
    if (!lowDirty_) return 0;
    (void)hipSetDevice(device_);
    collectBuilt();
    const int blockClass = keyClass();
    const bool defer = deferWanted();
    const std::string key = codeKey(blockClass, defer);
    if (!c_.key.empty() && c_.key == key) {   // (a register written with the value it had, a schedule armed again: nothing to do)
        lowDirty_ = false;
        return 0;
    }
    stashCode();
    bool have = adoptCode(key);
    if (!have && waitBuild(key)) {   // the builder thread is at it (the control variant, asked for at the first block): shorter than starting over
        collectBuilt();
        have = adoptCode(key);
    }
    if (have) {   // code for this shape exists: a pointer swap
        ++cacheHits_;
        lowDirty_ = false;
        adoptStageOptions();
        prebuildControlVariant();
        return 0;
    }
    std::string err;
    const int rc = buildCodeInto(c_, buildInputs(key, blockClass, defer), false, &err);
    if (rc != 0) return fail(rc, err);
    lowDirty_ = false;
    adoptStageOptions();
    prebuildControlVariant();
    return 0;
}

// The code in force came with the planner's ranking of the stage counts (Code::stageOptions).  The first code of a class of
// block lengths starts that class's tuner: the options the model cannot tell apart (within kTuneBand of the cheapest, three at
// most) are generated on the builder thread and then timed on the caller's own launches, kTuneRuns each (noteLaunchTime); the
// fastest is kept.  FX_STAGES_TUNE=0 (or no builder thread): the model's choice stands.
void Batch::adoptStageOptions() {
    const int cls = c_.blockClass;
    if (cls < 0 || cls >= 3 || !c_.useXlate || keyClass() != cls) return;
    Tuner& t = tune_[cls];
    if (t.init) return;
    t = Tuner();
    t.init = true;
    t.pick = c_.stagePick;
    // (built for "the cheapest": from now on the code goes by the stage count it was built for)
    c_.key = codeKeyFor(laneForced(), cls, c_.deferred, t.pick);
    const bool tuneOff = !knobs_.stagesTune;
    if (c_.stageOptions.empty() || knobs_.stages) { t.done = true; return; }
    const double best = c_.stageOptions.front().predicted;
    for (const StageOption& o : c_.stageOptions)
        if (t.options.size() < 3 && (t.options.empty() || o.predicted <= best * kTuneBand)) t.options.push_back(o);
    bool mine = false;
    for (const StageOption& o : t.options) mine = mine || o.wanted == t.pick;
    if (!mine || t.options.size() < 2 || tuneOff || !builderWanted()) { t.options.clear(); t.done = true; return; }
    t.bestNs.assign(t.options.size(), 0.0f);
    t.runs.assign(t.options.size(), 0);
    for (const StageOption& o : t.options) {
        if (o.wanted == t.pick) continue;
        BuildInputs in = buildInputs(codeKeyFor(laneForced(), cls, false, o.wanted), cls, false);
        in.stagePick = o.wanted;
        requestBuild(std::move(in));
    }
}

// Called at the head of a process call: what the previous launch took goes to the tuner of its class, and the tuner decides what
// the next launch runs - the same option again (kTuneRuns launches each), the next one whose code the builder has finished, or,
// when every option has been timed, the fastest for good.  All options compute the same words: a trial costs time, never bits.
void Batch::noteLaunchTime() {
    const int cls = lastLaunchClass_;
    if (cls < 0 || cls >= 3) return;
    Tuner& t = tune_[cls];
    if (!t.init || t.done) return;
    // (hipErrorNotReady is an answer, not a failure: it must not stay behind as the thread's "last error" for the launch
    // helpers that ask hipGetLastError() after their kernel)
    const hipError_t ready = (lastLaunchTimed_ && launched_) ? hipEventQuery(ev1_) : hipErrorNotReady;
    if (ready != hipSuccess) (void)hipGetLastError();
    if (ready == hipSuccess) {
        lastLaunchTimed_ = false;
        float ms = -1.0f;
        if (hipEventElapsedTime(&ms, ev0_, ev1_) == hipSuccess && ms > 0.0f && lastLaunchSamples_ >= kTuneMinSamples)
            for (size_t k = 0; k < t.options.size(); ++k)
                if (t.options[k].wanted == lastLaunchPick_) {
                    const float ns = ms * 1e6f / (float)lastLaunchSamples_;
                    t.bestNs[k] = t.runs[k] == 0 ? ns : std::min(t.bestNs[k], ns);
                    ++t.runs[k];
                    ++t.trials;
                }
    }
    if (lowDirty_ || keyClass() != cls) return;
    size_t cur = 0;
    while (cur < t.options.size() && t.options[cur].wanted != t.pick) ++cur;
    if (cur == t.options.size() || t.runs[cur] < kTuneRuns) return;
    collectBuilt();
    bool waiting = false;
    for (size_t k = 0; k < t.options.size(); ++k) {
        if (t.runs[k] >= kTuneRuns) continue;
        const std::string key = codeKeyFor(laneForced(), cls, false, t.options[k].wanted);
        if (cachedCode(key)) {   // its turn
            t.pick = t.options[k].wanted;
            lowDirty_ = true;
            return;
        }
        if (buildFailed(key)) continue;   // (the builder could not make it: out of the race)
        if (!buildPending(key)) {         // (e.g. the set of registers with rows has changed since the options were asked for)
            BuildInputs in = buildInputs(key, cls, false);
            in.stagePick = t.options[k].wanted;
            requestBuild(std::move(in));
        }
        waiting = waiting || buildPending(key);
    }
    if (waiting) return;
    size_t bestK = cur;
    for (size_t k = 0; k < t.options.size(); ++k)
        if (t.runs[k] >= kTuneRuns && t.bestNs[k] < t.bestNs[bestK]) bestK = k;
    t.done = true;
    if (t.options[bestK].wanted != t.pick) {
        t.pick = t.options[bestK].wanted;
        lowDirty_ = true;
    } else {
        prebuildControlVariant();
    }
}

// ---- the builder thread: code generated off the caller's thread -------------------------------------------------------------
// A translation and its module load take milliseconds; a real-time caller has 667 us per 32-sample block.  Two
// changes of code can be seen coming: the variant in which the declared controls have rows (wanted at the first touch of a
// slider - asked for right after the first build) and the code for another class of block lengths (asked for at the first
// block of that class, while the code in force - correct for every length, only slower - keeps running).  Both are built
// here and handed over through `finished`; the caller's thread picks them up at its next lowering (collectBuilt) as cache
// entries, so what it does then is a pointer swap.  FX_BUILDER=0: no thread, everything on the caller's (diagnostics).
struct Batch::Builder {
    std::thread thread;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<BuildInputs> jobs;
    std::string running;                          // key being built
    std::vector<std::unique_ptr<Code>> finished;
    std::deque<std::string> failed;               // keys the offline path could not build (left to the caller's thread); the most recent kMaxFailed
    static constexpr size_t kMaxFailed = 32;      // (a forgotten one is merely asked for again)
    hipStream_t upload = nullptr;                 // the thread's own copy stream (created and destroyed by it; read by it only)
    bool quit = false;
};

bool Batch::builderWanted() const {
    return knobs_.builder;
}

void Batch::requestBuild(BuildInputs&& in) {
    if (!builderWanted()) return;
    if (!builder_) {
        builder_.reset(new Builder);
        Builder* b = builder_.get();
        b->thread = std::thread([this, b] {
            (void)hipSetDevice(device_);
            if (hipStreamCreateWithFlags(&b->upload, hipStreamNonBlocking) != hipSuccess) { b->upload = nullptr; (void)hipGetLastError(); }
            std::unique_lock<std::mutex> lock(b->mu);
            for (;;) {
                b->cv.wait(lock, [b] { return b->quit || !b->jobs.empty(); });
                if (b->quit) {
                    if (b->upload) (void)hipStreamDestroy(b->upload);
                    b->upload = nullptr;
                    return;
                }
                BuildInputs job = std::move(b->jobs.front());
                b->jobs.pop_front();
                b->running = job.key;
                lock.unlock();
                std::unique_ptr<Code> c(new Code);
                std::string err;
                const int rc = buildCodeInto(*c, job, true, &err);
                if (rc != 0) releaseCode(*c);
                lock.lock();
                b->running.clear();
                if (rc == 0) b->finished.push_back(std::move(c));
                else {
                    b->failed.push_back(job.key);
                    if (b->failed.size() > Builder::kMaxFailed) b->failed.pop_front();
                }
                b->cv.notify_all();
            }
        });
    }
    std::lock_guard<std::mutex> lock(builder_->mu);
    if (builder_->running == in.key) return;
    for (const BuildInputs& j : builder_->jobs) if (j.key == in.key) return;
    for (const std::unique_ptr<Code>& c : builder_->finished) if (c->key == in.key) return;
    for (const std::string& k : builder_->failed) if (k == in.key) return;
    builder_->jobs.push_back(std::move(in));
    builder_->cv.notify_all();
}

void Batch::collectBuilt() {
    if (!builder_) return;
    std::vector<std::unique_ptr<Code>> got;
    {
        std::lock_guard<std::mutex> lock(builder_->mu);
        got.swap(builder_->finished);
    }
    for (std::unique_ptr<Code>& c : got) {
        if (cachedCode(c->key) || c_.key == c->key) { releaseCode(*c); continue; }
        c->lastUse = ++useClock_;
        cache_.push_back(std::move(c));
        if (cache_.size() > kCodeCache) {
            const size_t lru = lruVictim();
            waitLastLaunch();
            releaseCode(*cache_[lru]);
            cache_.erase(cache_.begin() + (long)lru);
        }
    }
}

bool Batch::buildPending(const std::string& key) {
    if (!builder_) return false;
    std::lock_guard<std::mutex> lock(builder_->mu);
    if (builder_->running == key) return true;
    for (const BuildInputs& j : builder_->jobs) if (j.key == key) return true;
    return false;
}

bool Batch::buildFailed(const std::string& key) {
    if (!builder_) return true;
    std::lock_guard<std::mutex> lock(builder_->mu);
    for (const std::string& k : builder_->failed) if (k == key) return true;
    return false;
}

// true: the builder has (or had) this key in hand and is done with it now
bool Batch::waitBuild(const std::string& key) {
    if (!builder_) return false;
    std::unique_lock<std::mutex> lock(builder_->mu);
    auto pending = [&] {
        if (builder_->running == key) return true;
        for (const BuildInputs& j : builder_->jobs) if (j.key == key) return true;
        return false;
    };
    if (!pending()) {
        for (const std::unique_ptr<Code>& c : builder_->finished) if (c->key == key) return true;
        return false;
    }
    builder_->cv.wait(lock, [&] { return !pending(); });
    return true;
}

// before anything a build reads changes (a load, an option) and at the end: no job running, none queued, nothing to pick up
void Batch::drainBuilder(bool stop) {
    if (!builder_) return;
    {
        std::unique_lock<std::mutex> lock(builder_->mu);
        builder_->jobs.clear();
        builder_->cv.wait(lock, [&] { return builder_->running.empty(); });
        for (std::unique_ptr<Code>& c : builder_->finished) releaseCode(*c);
        builder_->finished.clear();
        builder_->failed.clear();
        if (stop) {
            builder_->quit = true;
            builder_->cv.notify_all();
        }
    }
    if (stop) {
        builder_->thread.join();
        builder_.reset();
    }
}

void Batch::prebuildControlVariant() {
    if (controlMode_ || c_.key.empty() || !c_.useXlate || !builderWanted()) return;
    std::vector<uint8_t> forced = laneForced();
    bool any = false;
    for (const std::string& name : prog_.controls) {
        const int r = prog_.findRegister(name);
        if (r < 0 || forced[(size_t)r] || intrinsicLane(r) || !readByProgram(r) || !movableControl(r)) continue;
        forced[(size_t)r] = 1;
        any = true;
    }
    if (!any) return;
    const int blockClass = keyClass();
    // ... for the stage count in force and for every one still on trial: a slider may move while the trials run
    std::vector<int> picks{pickFor(blockClass)};
    if (blockClass >= 0 && blockClass < 3 && tune_[blockClass].init && !tune_[blockClass].done)
        for (const StageOption& o : tune_[blockClass].options)
            if (std::find(picks.begin(), picks.end(), o.wanted) == picks.end()) picks.push_back(o.wanted);
    for (int pick : picks) {
        BuildInputs in = buildInputs(codeKeyFor(forced, blockClass, false, pick), blockClass, false);
        if (cachedCode(in.key)) continue;
        in.forced = forced;
        in.stagePick = pick;
        requestBuild(std::move(in));
    }
}

Batch::BuildInputs Batch::buildInputs(const std::string& key, int blockClass, bool defer) const {
    BuildInputs in;
    in.key = key;
    in.blockClass = blockClass;
    in.defer = defer;
    in.hostValue = hostValue_;
    in.forced = laneForced();
    in.trackRegs = trackRegs_;
    in.stagePick = pickFor(blockClass);
    in.instPerLane = instPerLane_;
    in.iSlotsAlloc = iSlotsAlloc_;
    in.xSlotsAlloc = xSlotsAlloc_;
    in.stateRows = stateRows_;
    in.stagingOff = stagingOff_;
    return in;
}

// The lowering itself, into an empty Code: lower the program for the tier that takes it, translate it where it can be
// translated, load the code object, upload the tables.  offline: on the builder thread, while the batch keeps running other
// code - nothing of the batch's device state may change (no new state rows, no delay-line allocation) and only the translated
// tier qualifies; whatever else the program would need is left to the caller's thread (FX_E_NOTREADY).
int Batch::buildCodeInto(Code& c, const BuildInputs& in, bool offline, std::string* err) {
    auto fail = [&](int code, const std::string& what) { *err = what; return code; };
    auto hipFail = [&](hipError_t e, const char* where) {
        (void)hipGetLastError();   // (reported here: not again by the next launch helper that asks, Batch::hipFail)
        *err = std::string(where) + ": " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? FX_E_MEMORY : FX_E_NODEVICE;
    };
    const int blockClass = in.blockClass;
    const bool defer = in.defer;
    // Preferred: the hand-written gfx950 interpreter (one instance per lane, bookkeeping in VGPRs).
    // Programs it does not cover run on the HIP C++ kernel.  TRAM tiling pins K once allocated.
    Lowered fresh;
    bool asmOk = false;
    // (offline: the snapshot the request came with - the members belong to the caller's thread)
    const int instPerLane = offline ? in.instPerLane : instPerLane_;
    const bool stagingOff = offline ? in.stagingOff : stagingOff_;
    const bool tramPinned = offline ? (in.iSlotsAlloc > 0 || in.xSlotsAlloc > 0) : (iSlotsAlloc_ > 0 || xSlotsAlloc_ > 0);
    const char* forceHip = knobs_.kernel.empty() ? nullptr : knobs_.kernel.c_str();
    const bool wantAsm = !knobs_.kernelIs("hip") && !knobs_.instPerLaneSet;
    if (wantAsm && (!tramPinned || instPerLane == 1)) {
        // first choice: register file in VGPRs (row pitch 1 = plain indices), else in LDS
        const bool tryVgpr = !(forceHip && std::strcmp(forceHip, "asm_lds") == 0);
        if (tryVgpr) {
            fresh = lowerProgram(prog_, in.hostValue, in.forced, 1, false, 1);
            asmOk = fresh.error.empty() && asmEligible(fresh, &c.asmWhyNot);
            if (asmOk) {
                // smallest VGPR build that holds the register file = most wavefronts per SIMD
                int v = ASM_V64;
                while (v < ASM_V256 && fresh.nRows > kAsmVgprRows[v]) ++v;
                const int smallest = v;
                // ... and a larger one while that costs no residency this batch can use: the translator keeps the constants of
                // its LOG / EXP index guess and a small cache of products in VGPRs above the register file (fx_xlate.hpp).
                // (The interpreter tier has no use for spare registers, but runs the same build: it is the translator's fallback.)
                {
                    const int wavesPerSimd = (int)((((size_t)n_ + 63) / 64 + 1023) / 1024);  // 256 CUs x 4 SIMDs
                    auto usable = [&](int q) { return std::min(kAsmWavesPerSimd[q], std::max(wavesPerSimd, 1)); };
                    while (v < ASM_V256 && kAsmVgprRows[v] - fresh.nRows < kSpareVgprsWanted && usable(v + 1) >= usable(v)) ++v;
                }
                // a small batch is cut into stages (below): each stage wants spare registers for its packets and its input
                // bursts, and at most 4 wavefronts per SIMD will be resident anyway - the 128-register build costs nothing
                if (stagingPossibleGiven(stagingOff))
                    while (v < ASM_V128) ++v;
                // two or more wavefronts per SIMD: they take turns at the top priority (fx_xlate.hpp prioritySlices), which has
                // four levels - and a fifth resident wavefront adds nothing to a SIMD that four keep issuing (measured: config5
                // at 5 per SIMD on the 96-register build = at 4 per SIMD).  So at most four slots: the 128-register build or
                // larger, for batches of any number of rounds (1 048 576 instances, 4 slots with turns against 5-8 without:
                // config5 + 2.0 %, config4 + 2.4 %, the memory-bound probe and config3 unchanged).
                if (((size_t)n_ + 63) / 64 >= 2048)
                    while (v < ASM_V128) ++v;
                const char* pin = forceHip ? std::strstr(forceHip, "_v") : nullptr;
                if (pin && (std::strncmp(forceHip, "asm_v", 5) == 0 || std::strncmp(forceHip, "xlate_v", 7) == 0)) {
                    // diagnostics: pin a (large enough) build of the interpreter (asm_vNN) or of the translator (xlate_vNN)
                    static const char* const tags[ASM_VARIANTS] = {"", "_v64", "_v72", "_v80", "_v96", "_v128", "_v168", "_v256"};
                    for (int q = smallest; q < ASM_VARIANTS; ++q)
                        if (std::strcmp(pin, tags[q]) == 0) v = q;
                }
                c.variant = (AsmVariant)v;
            }
        }
        if (!asmOk) {
            fresh = lowerProgram(prog_, in.hostValue, in.forced, 1, false);
            asmOk = fresh.error.empty() && asmEligible(fresh, &c.asmWhyNot);
            c.variant = ASM_LDS;
        }
    } else {
        c.asmWhyNot = "disabled by FX_KERNEL / FX_INST_PER_LANE";
    }
    if (!asmOk && offline) return fail(FX_E_NOTREADY, "offline build: not a program for the assembly tiers");
    if (!asmOk) fresh = lowerProgram(prog_, in.hostValue, in.forced, chooseInstPerLane());
    if (!fresh.error.empty()) return fail(FX_E_PROGRAM, fresh.error);
    if (offline && (fresh.instPerLane != in.instPerLane || fresh.iSlots > in.iSlotsAlloc || fresh.xSlots > in.xSlotsAlloc || makeLayout((int)prog_.regs.size(), prog_.numChannels).totalRows != in.stateRows))
        return fail(FX_E_NOTREADY, "offline build: the batch's device state would have to change");
    int rc = 0;
    if (!offline) {
        instPerLane_ = fresh.instPerLane;
        rc = ensureState();
        if (rc != 0) { *err = lastError_; return rc; }
    }
    c.useAsm = asmOk;
    // (a register that turns per-instance needs no seeding: the state row of EVERY register holds its current value at all
    // times - ensureState fills new ones, setRegister writes through - and a per-instance write made before the first block
    // must survive the first lowering)
    c.low = std::move(fresh);
    if (!offline) {
        if ((rc = ensureTram(c.low)) != 0) { *err = lastError_; return rc; }
    }

    // upload: steady | last | row table
    c.useXlate = false;
    c.stages = 1;
    c.deferred = false;
    c.xlateWhyNot.clear();
    if (c.useAsm && forceHip && std::strncmp(forceHip, "asm", 3) == 0) c.xlateWhyNot = "the interpreter is pinned by FX_KERNEL";
    else if (c.useAsm && c.variant == ASM_LDS) c.xlateWhyNot = "register file in LDS (above 224 rows): no translation template";
    else if (c.useAsm && c.low.multipass) c.xlateWhyNot = "END can be skipped (multi-pass program): the interpreter runs the passes";
    if (c.useAsm && c.low.multipass) {
        // (generated code is one pass over the program; the interpreter's end-of-sample handler starts the next one)
    } else if (c.useAsm && c.variant != ASM_LDS && defer) {
        // controls are moving (a set_register within the last few blocks): a translation costs a module load
        // (~1-2 ms), a re-encode for the interpreter ~0.05 ms - interpret until the controls have been quiet
        c.deferred = true;
        c.xlateWhyNot = "deferred: control registers are changing";
    } else if (c.useAsm && c.variant != ASM_LDS && !(forceHip && std::strncmp(forceHip, "asm", 3) == 0)) {
        // first choice for a VGPR build: translate the program into gfx950 code (FX_KERNEL=asm* pins the interpreter)
        const std::vector<MicroOp> steadyRecords = encodeAsmStream(c.low.steady, nullptr, true), lastRecords = encodeAsmStream(c.low.last, nullptr, true);
        std::vector<int> trackRows;
        for (int reg : in.trackRegs) trackRows.push_back(c.low.rowOfReg[(size_t)reg]);
        XlateProgram xprog = xlateProgramOf(steadyRecords, lastRecords, prog_.iTramSize, prog_.xTramSize, c.low.nRows, c.low.inRow, c.low.latchRow, trackRows);
        // 256 bytes per wavefront and slot; the Infinity Cache holds 256 MiB
        const size_t slotsAlloc = offline ? (size_t)in.iSlotsAlloc + (size_t)in.xSlotsAlloc : (size_t)iSlotsAlloc_ + (size_t)xSlotsAlloc_;
        xprog.tramStreaming = slotsAlloc * (((size_t)n_ + 63) / 64) * 256 > ((size_t)512 << 20);
        {
            // wavefronts of a SIMD by turns at the top priority (fx_xlate.hpp prioritySlices): wherever a SIMD holds two or more
            // (the build chosen above has at most four slots then; FX_XLATE_PRIO=0 / 1 in the environment: never / whenever unstaged)
            const size_t waves = ((size_t)n_ + 63) / 64, simds = 1024;
            xprog.prioritySlices = knobs_.xlatePrio >= 0 ? knobs_.xlatePrio != 0 : (waves >= 2 * simds && kAsmWavesPerSimd[c.variant] <= 4);
            c.prioritySlices = xprog.prioritySlices;
        }
        XlateImage image;
        const XlateTemplate* tmpl = nullptr;
        bool built = false;
        tmpl = xlateTemplate(c.variant, &c.xlateWhyNot);
        // Small batches leave SIMDs empty (and a lone wavefront issues an instruction every ~4.5 clocks): cut the program
        // into stages run by the wavefronts of one workgroup (fx_xlate.hpp StageInfo) until ~4 wavefronts per SIMD are in
        // flight.  FX_STAGES pins the number asked for (1 = never).
        // how many stages: the caller's pick (a measured one, or an option on trial), else the cheapest by the planner's costs
        c.stageOptions = tmpl ? rankStages(steadyRecords, lastRecords, xprog, c.low.nRows, blockClass, kAsmWavesPerSimd[c.variant], stagingOff) : std::vector<StageOption>();
        int wantStages = in.stagePick > 0 ? in.stagePick : (c.stageOptions.empty() ? 1 : c.stageOptions.front().wanted);
        c.stagePick = wantStages;
        // (the wavefronts of a workgroup must be resident together: a CU holds 4 SIMDs x the build's wavefronts per SIMD - a pinned
        // FX_STAGES=16 in the 256-register build would be a launch that cannot start)
        wantStages = std::min(wantStages, 4 * kAsmWavesPerSimd[c.variant]);
        // Measured with config2 at 4 096 instances (profiles/r03b_stage_blocks.txt): a block of 32 samples takes 27 us unstaged, 32 us
        // in 8 stages with a barrier every 8 samples (3 x 7 steps of 8 samples to fill and drain) and 21 us in 4 stages with a
        // barrier per sample; 128 samples 69 / 48 / 40 us (8 stages, every 2 samples); from 256 samples on the long steps win
        // (rankStages charges a block of the class's typical length with the 3 (K - 1) steps of filling and draining)
        const int maxGroup = blockClass == 0 ? 1 : (blockClass == 1 ? 2 : kStageGroupMax);
        c.blockClass = blockClass;
        c.stagesWhyNot.clear();
        if (tmpl && wantStages >= 2) {
            const StagePlan plan = planStages(steadyRecords, lastRecords, xprog, c.low.nRows, wantStages);
            c.stagesWhyNot = plan.why;
            std::string why;
            c.classMatters = !plan.cuts.empty();
            if (!plan.cuts.empty()) {
                // (several workgroups per CU must fit its 160 KiB of LDS together)
                const int64_t groupsPerCu = std::max<int64_t>(1, ((n_ + 63) / 64 + 255) / 256);
                const uint32_t ldsBudget = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(144 * 1024, 160 * 1024 / groupsPerCu - 256));
                built = buildStagedImage(steadyRecords, lastRecords, *tmpl, xprog, plan, &image, nullptr, nullptr, &why, ldsBudget, maxGroup, knobs_.stagesGroup);
                if (!built) { c.stagesWhyNot = why; image = XlateImage(); }
            }
        }
        if (!built) built = tmpl && buildXlateImage(steadyRecords, lastRecords, *tmpl, xprog, &image, &c.xlateWhyNot);
        if (built) {
            hipError_t me = hipModuleLoadData(&c.module, image.elf.data());
            if (me == hipSuccess) me = hipModuleGetFunction(&c.fn, c.module, tmpl->kernelName.c_str());
            if (me != hipSuccess) return hipFail(me, "loading the translated program");
            c.steady = (uint64_t)image.steadyFastOff | ((uint64_t)image.steadyOff << 32);
            c.last = (uint64_t)image.lastFastOff | ((uint64_t)image.lastOff << 32);
            c.codeBytes = image.codeBytes;
            c.codeHash = imageHash(image);
            c.initOff = image.initOff;
            c.ldsBytes = image.ldsBytes;
            c.wildRow = image.wildRow;
            c.unsaturated = image.steady.unsaturated;
            c.inlined = image.steady.inlined;
            c.called = image.steady.called;
            c.valu = image.steady.valu;
            c.valuSlow = image.steady.valuSlow;
            c.valuClocks = image.steady.valuClocks;
            c.vgprConstants = image.vgprConstants;
            if (offline) ++backgroundBuilds_; else ++xlateBuilds_;
            c.stages = image.stages;
            c.stageDesc = image.stageDesc;
            c.stageStoreRows = image.stageStoreRows;
            c.useXlate = true;
        }
    }
    if (offline && !c.useXlate) return fail(FX_E_NOTREADY, "offline build: the translation failed (" + c.xlateWhyNot + ")");
    if (c.useAsm && !c.useXlate) {
        hipError_t pe = hipSuccess;
        const uint64_t* handlers = asmHandlerTable(c.variant, device_, &pe);
        if (!handlers) return hipFail(pe, "probe of the assembly interpreter");
        const bool fold = c.variant != ASM_LDS;
        c.low.steady = encodeAsmStream(c.low.steady, handlers, fold);
        c.low.last = encodeAsmStream(c.low.last, handlers, fold);
    }
    const size_t nOps = c.low.steady.size();
    const bool staged = c.useXlate && c.stages > 1;
    const size_t words = nOps * 8 * 2 + c.low.loadRows.size() + c.low.storeRows.size() + c.low.zeroRows.size() + (staged ? (size_t)c.stages * 8 : 0);
    if (words > c.streamCap) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&c.dStream), words * 4 + 256);
        if (e != hipSuccess) return hipFail(e, "hipMalloc stream");
        c.streamCap = words;
    }
    std::vector<uint32_t> host(words);
    std::memcpy(host.data(), c.low.steady.data(), nOps * 32);
    std::memcpy(host.data() + nOps * 8, c.low.last.data(), nOps * 32);
    size_t p = nOps * 16;
    for (const RowCopy& rcp : c.low.loadRows) {
        // translated programs: bit 15 marks a row of the BOUNDED class (its state value is checked against 1.0)
        const bool bounded = c.useXlate && rcp.ldsRow < c.wildRow.size() && !c.wildRow[rcp.ldsRow];
        host[p++] = rcp.ldsRow | (bounded ? 0x8000u : 0u) | ((uint32_t)rcp.stateRow << 16);
    }
    if (staged) {
        // the store rows grouped by the stage that owns them; each stage's descriptor names its slice
        for (int k = 0; k < c.stages; ++k) {
            c.stageDesc[(size_t)k].storeFirst = (uint32_t)(p - (nOps * 16 + c.low.loadRows.size()));
            uint32_t count = 0;
            for (const RowCopy& rcp : c.low.storeRows) {
                const std::vector<int>& mine = c.stageStoreRows[(size_t)k];
                if (std::find(mine.begin(), mine.end(), (int)rcp.ldsRow) == mine.end()) continue;
                host[p++] = rcp.ldsRow | ((uint32_t)rcp.stateRow << 16);
                ++count;
            }
            c.stageDesc[(size_t)k].storeCount = count;
        }
        if (p != nOps * 16 + c.low.loadRows.size() + c.low.storeRows.size()) return fail(FX_E_PROGRAM, "internal: a store row without a stage");
    } else {
        for (const RowCopy& rcp : c.low.storeRows) host[p++] = rcp.ldsRow | ((uint32_t)rcp.stateRow << 16);
    }
    for (int zr : c.low.zeroRows) host[p++] = (uint32_t)zr;
    if (staged) {
        static_assert(sizeof(StageDescriptor) == 32, "StageDescriptor layout");
        std::memcpy(host.data() + p, c.stageDesc.data(), (size_t)c.stages * 32);
        p += (size_t)c.stages * 8;
    }
    // (a buffer of its own: no launch reads it yet.  The builder thread copies through a non-blocking stream of its own: a plain
    // hipMemcpy goes through the null stream, which would wait for - and hold up - a caller that launches on a blocking stream)
    hipError_t e;
    if (offline && builder_ && builder_->upload) {
        e = hipMemcpyAsync(c.dStream, host.data(), words * 4, hipMemcpyHostToDevice, builder_->upload);
        if (e == hipSuccess) e = hipStreamSynchronize(builder_->upload);
    } else {
        e = hipMemcpy(c.dStream, host.data(), words * 4, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) return hipFail(e, "stream upload");
    c.key = in.key;
    return 0;
}

// A register whose value is compiled into the instruction stream (a literal of the generated code, an immediate of a record)
// and that the caller changes AFTER the program has run is a moving control - the reference's setRegisterValue is a store
// (source/FX8010.cpp:236-253), called every 8 samples by its harness (source/main.cpp:107-114).  Such a register is given a
// row of the register file, once: the code then reads it as a VGPR operand, and every later change is a fill of that row -
// no re-lowering, no re-translation.  Not for the few operand positions that shape the code itself: a LOG / EXP table
// number, a SKIP's condition or count, a delay-line offset (those keep being compiled in).
bool Batch::movableControl(int r) const {
    for (const Instr& in : prog_.instrs) {
        if ((in.op == LOG || in.op == EXP) && in.x == r) return false;
        if (in.op == SKIP && (in.x == r || in.y == r)) return false;
        if ((in.op == IDELAY || in.op == XDELAY) && in.y == r) return false;
    }
    return true;
}

// Every declared control (`control name = v`, the reference's control list, source/FX8010.cpp:408-411) that an instruction reads
// as a plain operand gets its row - all of them at the first touch of any: a host that moves one slider moves others (a preset
// recall writes the whole panel), and one change of code for the panel is one stall at most - none when the variant was built
// ahead (prebuildControlVariant).  Registers no instruction reads never get a row: their value cannot reach the code.
void Batch::markControls() {
    for (const std::string& name : prog_.controls) {
        const int r = prog_.findRegister(name);
        if (r < 0 || forcedLane_[(size_t)r] || intrinsicLane(r) || !readByProgram(r) || !movableControl(r)) continue;
        forcedLane_[(size_t)r] = 1;
        lowDirty_ = true;
    }
}

bool Batch::readByProgram(int reg) const { return reg >= 0 && (size_t)reg < readByProgram_.size() && readByProgram_[(size_t)reg] != 0; }

bool Batch::declaredControl(int reg) const { return reg >= 0 && (size_t)reg < declared_.size() && declared_[(size_t)reg] != 0; }

int Batch::setRegister(const std::string& key, float v) {
    (void)hipSetDevice(device_);
    const int r = prog_.findRegister(key);
    if (r < 0) return 1;
    hostValue_[r] = v;
    if (tracked(r) || intrinsicLane(r)) {
        // the register lives in a row whatever the host does (a schedule, or the program writes it): the fill below is all there is to do
    } else if (forcedLane_[r]) {
        if (controlMode_ && declaredControl(r)) controlWritten(r);   // (a control that had its row for company starts moving itself: lean code with its value folded in goes)
        // a row from an earlier write.  One that only per-instance writes asked for is given back now that every instance holds
        // the same value again (the register file does not grow with every register a host has ever touched) - unless it is a
        // moving control, whose next change should stay a fill
        if (laneWritten_[r] && !(controlMode_ && declaredControl(r)) ) {
            forcedLane_[r] = 0;
            lowDirty_ = true;
        }
    } else if (!readByProgram(r)) {
        // no instruction reads it: the value lives in the state row (get_register) and nowhere in the code
    } else if (loaded_ && everLowered_ && movableControl(r)) {
        // a moving control: a row from now on - and with it the other declared controls (one change of code for the panel)
        if (declaredControl(r)) {
            controlMode_ = true;
            markControls();
            coldSetChanged();
            controlWritten(r);
        }
        forcedLane_[r] = 1;
        lowDirty_ = true;
    } else {
        lowDirty_ = true;    // immediates (and possibly the classification) change
        if (loaded_ && everLowered_) controlHeat_ = kHeatPerChange;
    }
    laneWritten_[r] = 0;
    if (dState_) {
        // Invariant: the state row of EVERY register holds its current value for every instance, also while the
        // register is uniform (folded into the code) - so that a later per-instance write only has to force the
        // register per-lane, whatever the lowering in force says about it.
        waitLastLaunch();
        int rc = fillRows({(uint32_t)r}, {bitsOf(v)});
        if (rc != 0) return rc;
    }
    return 0;
}

int Batch::setRegisterAt(const std::string& key, int64_t inst, float v) {
    (void)hipSetDevice(device_);
    const int r = prog_.findRegister(key);
    if (r < 0) return 1;
    if (inst < 0 || inst >= n_) return fail(FX_E_ARG, "instance out of range");
    if (!dState_) return fail(FX_E_NOTREADY, "no program loaded");
    waitLastLaunch();
    if (coldControl(r)) coldSetChanged();
    if (!forcedLane_[r] && !intrinsicLane(r) && readByProgram(r)) {  // (the row is valid, see setRegister; the next lowering keeps the register per-lane)
        forcedLane_[r] = 1;
        lowDirty_ = true;
    }
    laneWritten_[r] = 1;
    hipError_t e = hipMemcpy(dState_ + (size_t)r * nPad_ + inst, &v, 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) return hipFail(e, "setRegisterAt");
    return 0;
}

int Batch::setRegisterArray(const std::string& key, const float* values) {
    (void)hipSetDevice(device_);
    const int r = prog_.findRegister(key);
    if (r < 0) return 1;
    if (!values) return fail(FX_E_ARG, "null buffer");
    if (!dState_) return fail(FX_E_NOTREADY, "no program loaded");
    waitLastLaunch();
    if (coldControl(r)) coldSetChanged();
    if (!forcedLane_[r] && !intrinsicLane(r) && readByProgram(r)) {  // from now on a per-instance row (every lane is overwritten below)
        forcedLane_[r] = 1;
        lowDirty_ = true;
    }
    laneWritten_[r] = 1;
    hipError_t e = hipMemcpy(dState_ + (size_t)r * nPad_, values, sizeof(float) * (size_t)n_, hipMemcpyHostToDevice);
    if (e != hipSuccess) return hipFail(e, "setRegisterArray");
    return 0;
}

int Batch::getRegisterArray(const std::string& key, float* values) {
    (void)hipSetDevice(device_);
    const int r = prog_.findRegister(key);
    if (r < 0) return 1;
    if (!values) return fail(FX_E_ARG, "null buffer");
    if (!dState_ || !laneResident(r)) {
        for (int64_t i = 0; i < n_; ++i) values[i] = hostValue_[r];
        return 0;
    }
    waitLastLaunch();
    hipError_t e = hipMemcpy(values, dState_ + (size_t)r * nPad_, sizeof(float) * (size_t)n_, hipMemcpyDeviceToHost);
    return e == hipSuccess ? 0 : hipFail(e, "getRegisterArray");
}

float Batch::getRegisterAt(const std::string& key, int64_t inst) {
    (void)hipSetDevice(device_);
    const int r = prog_.findRegister(key);
    if (r < 0) return 1.0f;  // reference getRegisterValue default, source/FX8010.cpp:265
    if (inst < 0 || inst >= n_ || !dState_ || !laneResident(r)) return hostValue_[r];
    waitLastLaunch();
    float v = 0.0f;
    if (hipMemcpy(&v, dState_ + (size_t)r * nPad_ + inst, 4, hipMemcpyDeviceToHost) != hipSuccess) return hostValue_[r];
    return v;
}

int Batch::seedNoiseAt(int64_t inst, int32_t x1, int32_t x2) {
    (void)hipSetDevice(device_);
    if (inst < 0 || inst >= n_) return fail(FX_E_ARG, "instance out of range");
    if (!dState_) return fail(FX_E_NOTREADY, "no program loaded");
    waitLastLaunch();
    hipError_t e = hipMemcpy(dState_ + (size_t)(stateLayout_.noiseBase + 0) * nPad_ + inst, &x1, 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dState_ + (size_t)(stateLayout_.noiseBase + 1) * nPad_ + inst, &x2, 4, hipMemcpyHostToDevice);
    return e == hipSuccess ? 0 : hipFail(e, "seedNoiseAt");
}

bool Batch::tracked(int reg) const { return std::find(trackRegs_.begin(), trackRegs_.end(), reg) != trackRegs_.end(); }

std::vector<uint8_t> Batch::laneForcedFull() const {
    std::vector<uint8_t> f = forcedLane_;
    for (int reg : trackRegs_)
        if ((size_t)reg < f.size()) f[(size_t)reg] = 1;
    return f;
}

// a declared control that has a row only because ANOTHER control moved (markControls): every instance holds hostValue_ of it,
// and nothing but setRegister / setRegisterAt / setRegisterArray / a schedule / a state image can change that - each of which
// calls coldSetChanged() first
bool Batch::coldControl(int reg) const {
    const size_t r = (size_t)reg;
    return controlMode_ && reg >= 0 && r < forcedLane_.size() && r < declared_.size() && r < hotControl_.size() && declared_[r] && forcedLane_[r] && !hotControl_[r] &&
           !laneWritten_[r] && !tracked(reg) && !intrinsicLane(reg);
}

// the controls that could be folded into the code right now
std::vector<uint8_t> Batch::coldControls() const {
    std::vector<uint8_t> cold(forcedLane_.size(), 0);
    for (size_t r = 0; r < cold.size(); ++r) cold[r] = coldControl((int)r);
    return cold;
}

// laneForcedFull() minus a set of folded controls.  (The set is remembered, not the result: rows that other registers get
// meanwhile - a per-instance write, a schedule - must show up in the key of the next lowering; found by the API fuzzer's control
// panel, seed 2600545.)
std::vector<uint8_t> Batch::forcedWithout(const std::vector<uint8_t>& folded) const {
    std::vector<uint8_t> f = laneForcedFull();
    for (size_t r = 0; r < f.size() && r < folded.size(); ++r)
        if (folded[r]) f[r] = 0;
    return f;
}

std::vector<uint8_t> Batch::laneForced() const { return leanActive_ ? forcedWithout(leanFolded_) : laneForcedFull(); }

// Something is about to change that the lean code in force (or on order) may have folded in: back to the full variant (in the
// cache, never evicted while controls have rows: lruVictim) for the next block; a new lean one is asked for then.
void Batch::coldSetChanged() {
    if (!controlMode_) return;
    if (leanActive_) {
        leanActive_ = false;
        lowDirty_ = true;
    }
    leanPending_ = false;
    leanStale_ = true;
}

// a declared control is being written (broadcast): it is hot from now on; lean code that has its old value folded in goes
void Batch::controlWritten(int reg) {
    const size_t r = (size_t)reg;
    if (!controlMode_ || r >= hotControl_.size()) return;
    // a control that starts moving again after it had cooled down rests twice as long before it is folded in the next time (a
    // slider that moves every few hundred milliseconds would otherwise have code built for it, in the background but beside a
    // real-time stream, again and again: tools/realtime_capacity.py with FX_RT_SLIDER_EVERY=300, profiles/r05_rt_slow_slider.txt)
    if (!hotControl_[r] && cooledOnce_[r]) coolAfter_[r] = std::min<int64_t>(coolAfter_[r] * 2, kCoolSamplesMost);
    hotControl_[r] = 1;
    lastControlWrite_[r] = sampleClock_;
    if (leanActive_ && r < leanFolded_.size() && leanFolded_[r]) coldSetChanged();
    else leanStale_ = true;
}

// Head of a block, code in force and clean.  Controls that have not been written for kCoolSamples sample periods cool down (a
// slider is at rest most of the time; a preset recall writes the whole panel once); when the set of controls that could be
// folded differs from what the code in force has folded, the variant for it is asked of the builder thread, and adopted - a
// pointer swap in the lowering that follows - once it has arrived and is still what is wanted.
void Batch::leanStep() {
    if (!controlMode_ || lowDirty_) return;
    if (!builderWanted() || c_.key.empty() || !c_.useXlate || c_.deferred || tracksArmed()) return;
    if (sampleClock_ - lastCoolCheck_ >= kCoolSamples / 8) {
        lastCoolCheck_ = sampleClock_;
        for (size_t r = 0; r < hotControl_.size(); ++r)
            if (hotControl_[r] && sampleClock_ - lastControlWrite_[r] >= coolAfter_[r]) {
                hotControl_[r] = 0;
                cooledOnce_[r] = 1;
                leanStale_ = true;
            }
    }
    if (!leanStale_ && !leanPending_) return;
    const int cls = keyClass();
    const int pick = pickFor(cls);
    if (leanPending_) {
        if (codeKeyFor(forcedWithout(leanWant_), cls, false, pick) != leanKey_) {   // (a folded value, another register's row, the class or the stage count has changed meanwhile)
            leanPending_ = false;
            leanStale_ = true;
        } else {
            collectBuilt();
            if (cachedCode(leanKey_)) {
                leanPending_ = false;
                leanFolded_ = leanWant_;
                leanActive_ = true;
                lowDirty_ = true;
                return;
            }
            if (buildPending(leanKey_)) return;
            leanPending_ = false;   // (the builder could not make it: what runs stays)
        }
    }
    if (!leanStale_) return;
    leanStale_ = false;
    const std::vector<uint8_t> want = coldControls();
    const bool none = std::find(want.begin(), want.end(), (uint8_t)1) == want.end();
    if (leanActive_ ? want == leanFolded_ : none) return;
    if (none) {   // every control with a row is hot again: the full variant is the lean one
        leanActive_ = false;
        lowDirty_ = true;
        return;
    }
    const std::vector<uint8_t> forced = forcedWithout(want);
    const std::string key = codeKeyFor(forced, cls, false, pick);
    if (cachedCode(key)) {
        leanFolded_ = want;
        leanActive_ = true;
        lowDirty_ = true;
        return;
    }
    if (builder_ && buildFailed(key)) return;
    BuildInputs in = buildInputs(key, cls, false);
    in.forced = forced;
    in.stagePick = pick;
    requestBuild(std::move(in));
    leanWant_ = want;
    leanKey_ = key;
    leanPending_ = true;
}

// the cache entry to give up when it is full: the least recently used - but never the full control variant while controls have
// rows (the code every first touch, per-instance write and state image falls back to without a translation)
size_t Batch::lruVictim() const {
    std::string keep;
    if (controlMode_) {
        const int cls = keyClass();
        keep = codeKeyFor(laneForcedFull(), cls, false, pickFor(cls));
    }
    size_t lru = cache_.size();
    for (size_t k = 0; k < cache_.size(); ++k) {
        if (!keep.empty() && cache_[k]->key == keep) continue;
        if (lru == cache_.size() || cache_[k]->lastUse < cache_[lru]->lastUse) lru = k;
    }
    return lru == cache_.size() ? 0 : lru;
}

int Batch::setRegisterTrack(const std::string& key, const float* values, int nSteps, int period, bool perInstance, int64_t pitch) {
    (void)hipSetDevice(device_);
    const int r = prog_.findRegister(key);
    if (r < 0) return 1;
    if (!values || nSteps < 1 || period < 1) return fail(FX_E_ARG, "track: values, n_steps >= 1 and period >= 1 are required");
    if (!dState_) return fail(FX_E_NOTREADY, "no program loaded");
    if (pitch <= 0) pitch = n_;
    if (perInstance && pitch < n_) return fail(FX_E_ARG, "track: pitch below the instance count");
    size_t slot = 0;
    while (slot < trackRegs_.size() && trackRegs_[slot] != r) ++slot;
    if (slot == trackRegs_.size()) {
        if (trackRegs_.size() >= (size_t)kMaxTracks) return fail(FX_E_ARG, "track: at most " + std::to_string(kMaxTracks) + " registers can have schedules");
        coldSetChanged();
        trackRegs_.push_back(r);
        pendingTracks_.resize(trackRegs_.size());
        lowDirty_ = true;  // the register gets a row of its own and the generated loop the code to re-load it
    }
    PendingTrack& t = pendingTracks_[slot];
    t.period = period;
    t.steps = nSteps;
    t.perInstance = perInstance;
    if (perInstance) {
        t.values.resize((size_t)nSteps * (size_t)n_);
        for (int k = 0; k < nSteps; ++k) std::memcpy(&t.values[(size_t)k * (size_t)n_], values + (size_t)k * (size_t)pitch, (size_t)n_ * 4);
    } else {
        t.values.assign(values, values + nSteps);
    }
    return 0;
}

// the events of the armed schedules, sorted by sample, and their values -> dTracks_ (on the launch stream, ahead of the kernel);
// the schedules are one-shot
int Batch::uploadTracks(int nSamples, hipStream_t s) {
    if (!tracksArmed() && tracksClear_ && dTracks_) return 0;  // nothing armed and the device list already says so
    struct Due { uint32_t sample, slot, step; };
    std::vector<Due> due;
    std::vector<int> used(pendingTracks_.size(), 0);
    size_t valueWords = 0;
    for (size_t k = 0; k < pendingTracks_.size(); ++k) {
        const PendingTrack& t = pendingTracks_[k];
        if (t.steps <= 0) continue;
        used[k] = std::min(t.steps, (nSamples + t.period - 1) / t.period);  // changes that fall inside this block
        for (int q = 0; q < used[k]; ++q) due.push_back({(uint32_t)q * (uint32_t)t.period, (uint32_t)k, (uint32_t)q});
        valueWords += t.perInstance ? (size_t)used[k] * (size_t)nPad_ : (size_t)used[k];
    }
    std::stable_sort(due.begin(), due.end(), [](const Due& a, const Due& b) { return a.sample < b.sample; });  // (same sample: by slot, as armed)
    const size_t listWords = (due.size() + 1) * 4;
    const size_t words = listWords + valueWords;
    if (words * 4 > tracksCap_) {
        waitLastLaunch();
        (void)hipFree(dTracks_);
        dTracks_ = nullptr;
        tracksCap_ = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&dTracks_), words * 4 + 256);
        if (e != hipSuccess) return hipFail(e, "hipMalloc tracks");
        tracksCap_ = words * 4 + 256;
    }
    waitLastLaunch();  // trackStage_ / dTracks_ of the previous block are free again
    trackStage_.assign(words, 0);
    std::vector<size_t> valuesAt(pendingTracks_.size(), 0);
    size_t at = listWords;
    for (size_t k = 0; k < pendingTracks_.size(); ++k) {
        PendingTrack& t = pendingTracks_[k];
        if (t.steps <= 0) continue;
        valuesAt[k] = at;
        const int reg = trackRegs_[k];
        if (t.perInstance) {
            for (int q = 0; q < used[k]; ++q) std::memcpy(&trackStage_[at + (size_t)q * (size_t)nPad_], &t.values[(size_t)q * (size_t)n_], (size_t)n_ * 4);
            at += (size_t)used[k] * (size_t)nPad_;
            forcedLane_[(size_t)reg] = 1;
            laneWritten_[(size_t)reg] = 1;
        } else {
            std::memcpy(&trackStage_[at], t.values.data(), (size_t)used[k] * 4);
            at += (size_t)used[k];
            hostValue_[(size_t)reg] = t.values[(size_t)used[k] - 1];  // what every instance holds after the block
            forcedLane_[(size_t)reg] = 0;
            laneWritten_[(size_t)reg] = 0;
        }
    }
    for (size_t i = 0; i < due.size(); ++i) {
        const Due& d = due[i];
        const PendingTrack& t = pendingTracks_[d.slot];
        const size_t value = valuesAt[d.slot] + (t.perInstance ? (size_t)d.step * (size_t)nPad_ : (size_t)d.step);
        const TrackEvent ev{d.sample, d.slot, (uint32_t)(value * 4), t.perInstance ? (uint32_t)(nPad_ * 4) : 4u};
        std::memcpy(&trackStage_[i * 4], &ev, sizeof(ev));
    }
    const TrackEvent closing{0xffffffffu, 0, 0, 4};
    std::memcpy(&trackStage_[due.size() * 4], &closing, sizeof(closing));
    for (PendingTrack& t : pendingTracks_) {
        t.steps = 0;
        t.values.clear();
    }
    tracksClear_ = due.empty();
    hipError_t e = hipMemcpyAsync(dTracks_, trackStage_.data(), words * 4, hipMemcpyHostToDevice, s);
    return e == hipSuccess ? 0 : hipFail(e, "tracks upload");
}

// Tiers without in-kernel tracks (interpreter, HIP C++ kernel): the same schedule by cutting the block at its change
// points and writing the registers in between - what the caller would have had to do.
int Batch::processWithTrackFallback(const float* dIn, float* dOut, int nSamples, hipStream_t stream) {
    std::vector<PendingTrack> tracks;
    tracks.swap(pendingTracks_);
    pendingTracks_.resize(trackRegs_.size());
    std::vector<int> cuts{0, nSamples};
    for (const PendingTrack& t : tracks)
        for (int k = 0; k < t.steps && (int64_t)k * t.period < nSamples; ++k) cuts.push_back(k * t.period);
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    const size_t rowFloats = (size_t)prog_.numChannels * (size_t)n_;
    for (size_t c = 0; c + 1 < cuts.size(); ++c) {
        const int lo = cuts[c], hi = cuts[c + 1];
        for (size_t k = 0; k < tracks.size(); ++k) {
            const PendingTrack& t = tracks[k];
            if (t.steps <= 0 || lo % t.period != 0 || lo / t.period >= t.steps) continue;
            const std::string& name = prog_.regs[(size_t)trackRegs_[k]].name;
            const int rc = t.perInstance ? setRegisterArray(name, &t.values[(size_t)(lo / t.period) * (size_t)n_]) : setRegister(name, t.values[(size_t)(lo / t.period)]);
            if (rc != 0) return rc < 0 ? rc : fail(FX_E_ARG, "track: register vanished");
        }
        controlHeat_ = 0;  // these writes are the schedule, not a moving slider
        const int rc = processDevice(dIn + (size_t)lo * rowFloats, dOut + (size_t)lo * rowFloats, hi - lo, stream);
        if (rc != 0) return rc;
    }
    return 0;
}

// Staged code is generated for a class of block lengths (a pipeline fills and drains in 3 (K - 1) steps: short blocks want
// short steps and fewer stages).  The class wanted follows the caller: at once when code for the new class exists already (a
// pointer swap in ensureLowered), after four blocks in a row otherwise - a stray block of another length is not worth a
// translation.  Programs that cannot be cut have one code for every length.
void Batch::noteBlockLength(int nSamples) {
    if (nSamples <= 0) return;
    const int cls = stageBlockClass(nSamples);
    if (wantedClass_ < 0 || c_.key.empty()) {   // the first block after a load (or after a failed build)
        wantedClass_ = cls;
        otherClassBlocks_ = 0;
        // (code that an fxb_info call had generated before the first block is for the shortest class)
        if (!c_.key.empty() && c_.classMatters && c_.blockClass != cls) lowDirty_ = true;
        return;
    }
    if (cls == wantedClass_) { otherClassBlocks_ = 0; return; }
    if (!c_.classMatters) return;   // one code for every block length
    ++otherClassBlocks_;
    const int was = wantedClass_;
    wantedClass_ = cls;
    collectBuilt();
    const int blockClass = keyClass();
    const bool defer = deferWanted();
    const std::string key = codeKey(blockClass, defer);
    if (cachedCode(key)) {
        otherClassBlocks_ = 0;
        lowDirty_ = true;
        return;
    }
    // not there: the builder thread makes it while the code in force (right for every length, only slower) keeps running; a
    // caller without that thread, or whose build cannot be done offline, gets it on its own thread once it has stayed
    if (!lowDirty_) requestBuild(buildInputs(key, blockClass, defer));
    if (otherClassBlocks_ >= 4 && !buildPending(key)) {
        otherClassBlocks_ = 0;
        lowDirty_ = true;
        return;
    }
    wantedClass_ = was;
}

int Batch::processDevice(const float* dIn, float* dOut, int nSamples, hipStream_t stream) {
    (void)hipSetDevice(device_);
    if (nSamples < 0) return fail(FX_E_ARG, "n_samples < 0");
    if (!piecewise_) {   // (a piece of a pipelined host block: done once for the whole block)
        pendingSamples_ = nSamples;
        if (controlHeat_ > 0 && --controlHeat_ == 0 && c_.deferred) lowDirty_ = true;  // quiet again: translate
        noteLaunchTime();
        noteBlockLength(nSamples);
        leanStep();
        sampleClock_ += nSamples;
    }
    int rc = ensureLowered();
    if (rc != 0) return rc;
    everLowered_ = true;
    if (nSamples == 0) return 0;
    if (!dIn || !dOut) return fail(FX_E_ARG, "null buffer");
    if (tracksArmed() && !c_.useXlate) return processWithTrackFallback(dIn, dOut, nSamples, stream);
    hipStream_t s = pick(stream);
    if (c_.useXlate && !trackRegs_.empty() && (rc = uploadTracks(nSamples, s)) != 0) return rc;
    KernelArgs a{};
    const size_t nOps = c_.low.steady.size();
    a.steady = c_.dStream;
    a.last = c_.dStream + nOps * 8;
    a.rowTable = c_.dStream + nOps * 16;
    a.state = dState_;
    a.in = dIn;
    a.out = dOut;
    a.itram = dITram_;
    a.xtram = dXTram_;
    a.lut = dLut_;
    a.n = n_;
    a.nPad = nPad_;
    a.nOps = (int)nOps;
    a.nLoad = (int)c_.low.loadRows.size();
    a.nStore = (int)c_.low.storeRows.size();
    a.nSamples = nSamples;
    a.channels = prog_.numChannels;
    for (int c = 0; c < kMaxChannels; ++c) {
        a.inRow[c] = c < prog_.numChannels ? c_.low.inRow[c] : -1;
        a.latchRow[c] = c < prog_.numChannels ? c_.low.latchRow[c] : 0;
    }
    a.iSlots = iSlotsAlloc_;
    a.xSlots = xSlotsAlloc_;
    a.iSize = prog_.iTramSize;
    a.xSize = prog_.xTramSize;
    a.nZero = (int)c_.low.zeroRows.size();
    const uint32_t rowBytes = 256u * (uint32_t)instPerLane_;
    a.skipOff = c_.low.skipRow >= 0 ? (uint32_t)c_.low.skipRow * rowBytes : 0;
    a.cursorOff = c_.low.cursorRow >= 0 ? (uint32_t)c_.low.cursorRow * rowBytes : 0;
    a.noiseOff = c_.low.noiseRow >= 0 ? (uint32_t)c_.low.noiseRow * rowBytes : 0;
    a.oodOff = c_.low.oodRow >= 0 ? (uint32_t)c_.low.oodRow * rowBytes : 0;
    a.aliveOff = c_.low.aliveRow >= 0 ? (uint32_t)c_.low.aliveRow * rowBytes : 0;
    a.hasShadow = c_.low.skipRow >= 0 ? 1 : 0;
    a.instPerLane = instPerLane_;
    a.tramDane = (c_.low.tramDane && c_.low.cursorRow >= 0) ? 1 : 0;
    a.oodRow = stateLayout_.oodRow;
    a.countLo = stateLayout_.countLo;
    a.countHi = stateLayout_.countHi;
    a.staticCount = c_.low.staticCount;
    a.nRows = c_.low.nRows;
    hipError_t e = untimed_ ? hipSuccess : hipEventRecord(ev0_, s);
    if (e == hipSuccess) {
        if (c_.useAsm) {
            AsmArgs g{};
            g.steady = a.steady; g.last = a.last; g.rowTable = a.rowTable; g.state = a.state;
            g.in = a.in; g.out = a.out; g.itram = a.itram; g.xtram = a.xtram; g.lut = a.lut;
            g.n = a.n; g.nPad = a.nPad; g.nLoad = a.nLoad; g.nStore = a.nStore;
            g.nSamples = a.nSamples; g.channels = a.channels;
            for (int c = 0; c < kMaxChannels; ++c) {
                g.inOff[c] = a.inRow[c] >= 0 ? a.inRow[c] * (int)c_.low.rowPitch : -1;
                g.latchOff[c] = a.latchRow[c] * (int)c_.low.rowPitch;
            }
            g.iSlots = a.iSlots; g.xSlots = a.xSlots; g.iSize = a.iSize; g.xSize = a.xSize;
            g.cursorRow = stateLayout_.cursorBase; g.noiseRow = stateLayout_.noiseBase;
            g.oodRow = a.oodRow; g.countLo = a.countLo; g.countHi = a.countHi; g.staticCount = a.staticCount;
            g.lutX1Off = kLutX1Off * 8;
            g.tramDane = (c_.low.tramDane && (c_.low.usesITram || c_.low.usesXTram)) ? 1 : 0;   // (like the other tiers: counters step where the program has taps)
            if (c_.low.multipass) g.tramDane |= 2;
            if (!c_.useXlate && c_.variant != ASM_LDS) {
                // the interpreter's wavefronts take turns at the top priority like generated code's (fx_xlate.hpp prioritySlices):
                // wherever a SIMD holds two or more (a build of at most four slots); a turn = 1/24 of the block at about 20 us
                // per sample and four wavefronts, between 0.66 and 10 ms
                const size_t waves = ((size_t)n_ + 63) / 64;
                if (knobs_.xlatePrio >= 0 ? knobs_.xlatePrio != 0 : (waves >= 2048 && kAsmWavesPerSimd[c_.variant] <= 4)) {
                    int shift = 6;
                    while ((1 << (shift - 6 + 1)) <= a.nSamples) ++shift;   // floor(log2(samples of the launch)) + 6
                    g.tramDane |= 4 | (std::min(std::max(shift, 16), 20) << 8);
                }
            }
            if (c_.useXlate) {
                // code streams are named by their byte offset from the kernel entry: {fast, exact} per argument
                g.steady = reinterpret_cast<const uint32_t*>((uintptr_t)c_.steady);
                g.last = reinterpret_cast<const uint32_t*>((uintptr_t)c_.last);
                g.initOff = (int)c_.initOff;
                g.tracks = trackRegs_.empty() ? nullptr : dTracks_;
                if (c_.stages > 1) {
                    g.stages = c_.dStream + nOps * 16 + c_.low.loadRows.size() + c_.low.storeRows.size() + c_.low.zeroRows.size();
                    g.nStages = c_.stages;
                }
#ifdef FX_DIAGNOSTICS
                if (c_.stages <= 1 && FX_DIAG_KNOB("FX_XLATE_ENDSTAMP")) {
                    // the end stamps' own buffer rides in the kernarg slot of the stage descriptors, which an unstaged launch leaves
                    // unused (nStages stays 0: the template never looks at the pointer)
                    const size_t words = ((size_t)n_ + 63) / 64;
                    if (words > stampWords_) {
                        waitLastLaunch();
                        (void)hipFree(dStamps_);
                        dStamps_ = nullptr;
                        stampWords_ = 0;
                        if (hipMalloc(reinterpret_cast<void**>(&dStamps_), words * 4) != hipSuccess) return fail(FX_E_MEMORY, "end stamps");
                        stampWords_ = words;
                    }
                    g.stages = dStamps_;
                }
#endif
                e = launchAsmFunction(c_.fn, g, (unsigned)((n_ + 63) / 64), c_.ldsBytes, s, (unsigned)c_.stages);
            } else {
                e = launchAsmInterp(g, c_.variant, c_.variant == ASM_LDS ? (size_t)a.nRows * 256 : 0, device_, s);
            }
        } else {
            e = launchStepBlock(a, c_.low.multipass, s);
        }
    }
    const hipError_t launchError = e;   // (of the event record in front of the launch or of the launch itself)
    if (e == hipSuccess && !untimed_) e = hipEventRecord(ev1_, s);
    // A workgroup of several wavefronts that the device will not start (registers x wavefronts beyond a CU, LDS): the plain
    // program runs everywhere - no stages for this handle from now on, and this block again.  Only for what a launch
    // CONFIGURATION can cause: any other error (a fault of an earlier kernel that this call merely inherits, a lost device) is
    // reported as it is and leaves the handle's choice of code alone.  Nothing has been consumed at this point that the second
    // attempt needs: a staged program has no control tracks (planStages refuses them), so uploadTracks has not run.
    const bool configError = launchError == hipErrorInvalidValue || launchError == hipErrorInvalidConfiguration || launchError == hipErrorLaunchOutOfResources;
    if (configError && c_.useXlate && c_.stages > 1 && !stagingOff_ && trackRegs_.empty()) {
        (void)hipGetLastError();   // (the launch's own error, just read: not a sticky one)
        stagingOff_ = true;
        lowDirty_ = true;
        return processDevice(dIn, dOut, nSamples, stream);
    }
    if (e != hipSuccess) return hipFail(e, "launch fx_step_block");
    launched_ = !untimed_;  // (an untimed launch is synchronised by its caller before anything else happens)
    timed_ = !untimed_;
    lastLaunchTimed_ = !untimed_ && !piecewise_;
    lastLaunchPick_ = c_.stagePick;
    lastLaunchSamples_ = nSamples;
    lastLaunchClass_ = c_.useXlate ? c_.blockClass : -1;
    lastGrid_ = (unsigned)((n_ + 64 * instPerLane_ - 1) / (64 * instPerLane_));
    return 0;
}

// pitch: instances per PCM row of the HOST buffers (>= n_); a shard of a larger batch reads / writes its columns of the
// caller's [sample][channel][all instances] arrays in place (fx_shard.cpp)
namespace {
constexpr size_t kPinnedFloats = 512;
// A host block of 32 MiB and more is cut into eight pieces (consecutive sample ranges) whose copy-in, kernel and copy-out overlap
// on three streams (round 2: 268 MB each way in 6.5 instead of 12.7 ms from pinned buffers).  Smaller blocks were tried in pieces in
// round 5 (by size, 2-8 of them) for real-time callers and gained little - 32 samples x 131 072 instances: 806 -> 622 us - because
// the runtime's copy-OUT is a shader copy that slows a kernel running beside it threefold (profiles/r05_rt_timeline_131072.txt),
// and pieces are not timed by the stage tuner; what serves those callers is the in-place path above (pinned buffers: 486 us).
constexpr size_t kPipelinedBytes = (size_t)32 << 20;
inline int hostPieces(size_t bytes, int nSamples, int most) { return (bytes >= kPipelinedBytes && nSamples >= 2 * most) ? most : 1; }

// Is this host buffer memory the device can address (pinned by hipHostMalloc / hipHostRegister - e.g. a torch pinned tensor)?  Then
// the kernel reads and writes it in place.  Pageable memory: the runtime says "invalid value" (which must not stay behind as the
// thread's last error).
inline bool deviceVisibleHost(const void* host, const void** device) {
    hipPointerAttribute_t attr;
    std::memset(&attr, 0, sizeof(attr));
    if (hipPointerGetAttributes(&attr, host) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (attr.type != hipMemoryTypeHost || !attr.devicePointer) return false;
    *device = attr.devicePointer;
    return true;
}

// ... the whole of [host, host + bytes) inside ONE pinned mapping: the runtime's own record of the allocation the address belongs
// to; where it keeps none for registered memory, both ends pinned with one address offset between them.  A caller that registered
// part of a buffer takes the staged copies.
inline bool deviceVisibleRange(const void* host, size_t bytes, const void** device) {
    if (!deviceVisibleHost(host, device)) return false;
    if (bytes <= 1) return true;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, const_cast<void*>(*device)) == hipSuccess && base && size) {
        const char *lo = static_cast<const char*>(base), *at = static_cast<const char*>(*device);
        return at >= lo && bytes <= size && static_cast<size_t>(at - lo) <= size - bytes;
    }
    (void)hipGetLastError();
    const void* last = nullptr;
    if (!deviceVisibleHost(static_cast<const char*>(host) + (bytes - 1), &last)) return false;
    return static_cast<const char*>(last) - static_cast<const char*>(*device) == static_cast<std::ptrdiff_t>(bytes - 1);
}

// in == out is fine in place (an instance reads its sample before it writes it, and no other instance touches that word); ranges
// that overlap in any other way need the whole input read before the first output is written: the staged copies do that.
inline bool overlapButNotEqual(const void* a, const void* b, size_t bytes) {
    const char *x = static_cast<const char*>(a), *y = static_cast<const char*>(b);
    return x != y && x < y + bytes && y < x + bytes;
}
}  // namespace

int Batch::processHost(const float* in, float* out, int nSamples, int64_t pitch) {
    (void)hipSetDevice(device_);
    if (pitch <= 0) pitch = n_;
    if (pitch < n_) return fail(FX_E_ARG, "host row pitch below the instance count");
    if (nSamples < 0) return fail(FX_E_ARG, "n_samples < 0");
    if (nSamples == 0) return ensureLowered();
    if (!in || !out) return fail(FX_E_ARG, "null buffer");
    const size_t count = (size_t)nSamples * prog_.numChannels * (size_t)n_;
    // A few KB of PCM (per-sample calls on a handful of instances): two staged copies cost more than the launch.  The kernel
    // reads and writes pinned host memory instead - one launch, one synchronisation.
    if (count <= kPinnedFloats && pitch == n_) {
        if (!pinTried_) {
            pinTried_ = true;
            if (hipHostMalloc(reinterpret_cast<void**>(&hPinIn_), kPinnedFloats * 4, hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc(reinterpret_cast<void**>(&hPinOut_), kPinnedFloats * 4, hipHostMallocDefault) != hipSuccess) {
                if (hPinIn_) (void)hipHostFree(hPinIn_);
                hPinIn_ = hPinOut_ = nullptr;
                (void)hipGetLastError();
            }
        }
        if (hPinIn_ && hPinOut_) {
            std::memcpy(hPinIn_, in, count * 4);
            waitLastLaunch();
            // no event pair around a launch that is waited for right here (last_kernel_ms: -1) - unless schedules are armed:
            // the tiers that cut the block at every step launch several times and wait for each launch through its event
            untimed_ = !tracksArmed();
            int rc = processDevice(hPinIn_, hPinOut_, nSamples, stream_);
            untimed_ = false;
            if (rc != 0) return rc;
            hipError_t se = hipStreamSynchronize(stream_);
            if (se != hipSuccess) return hipFail(se, "synchronising a small block");
            std::memcpy(out, hPinOut_, count * 4);
            return 0;
        }
    }
    // The caller's buffers are pinned host memory (a real-time host keeps its PCM in such buffers): NO copies at all - the kernel
    // reads its input from and stores its output to the caller's memory over PCIe, in both directions at once, while it
    // computes.  One launch, one wait.  Measured (tools/realtime_capacity.py, 32-sample blocks of config5): the staged path's
    // copy-out is a shader copy (__amd_rocclr_copyBuffer) that slows a kernel running beside it threefold
    // (profiles/r05_rt_timeline_131072.txt); in place, a block of 131 072 instances takes about what its 16.8 MB each way take the
    // link.  FX_HOST_PIPELINE=0 keeps the staged copies.
    if (knobs_.hostPipeline && pitch == n_) {
        const void *dIn = nullptr, *dOut = nullptr;
        if (!overlapButNotEqual(in, out, count * 4) && deviceVisibleRange(in, count * 4, &dIn) &&
            (static_cast<const void*>(out) == in ? (dOut = dIn, true) : deviceVisibleRange(out, count * 4, &dOut))) {
            const int rc = processDevice(static_cast<const float*>(dIn), static_cast<float*>(const_cast<void*>(dOut)), nSamples, stream_);
            const hipError_t se = hipStreamSynchronize(stream_);   // (also when the call failed: nothing of it may still touch the caller's memory)
            if (rc != 0) return rc;
            return se == hipSuccess ? 0 : hipFail(se, "synchronising a block on pinned host buffers");
        }
    }
    if (count > ioCap_) {
        (void)hipStreamSynchronize(stream_);
        (void)hipFree(dIn_);
        (void)hipFree(dOut_);
        dIn_ = dOut_ = nullptr;
        ioCap_ = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&dIn_), count * 4);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dOut_), count * 4);
        if (e != hipSuccess) return hipFail(e, "hipMalloc io");
        ioCap_ = count;
    }
    // Large blocks: copy-in, kernel and copy-out of consecutive pieces overlap.  268 MB each way (tools/host_block_rate.py):
    // pinned caller buffers 6.5 ms instead of 12.7 (both DMA directions at once), pageable ones 9.7 instead of 12.9 (the driver
    // pins them on the fly; a freshly allocated, untouched output buffer costs 2-3 x that in page faults either way).
    const int pieces = hostPieces(count * 4, nSamples, kHostPieces);
    if (pieces >= 2 && !tracksArmed() && knobs_.hostPipeline)
        return processHostPipelined(in, out, nSamples, pitch, pieces);
    const size_t rows = (size_t)nSamples * prog_.numChannels, width = (size_t)n_ * 4;
    hipError_t e = pitch == n_ ? hipMemcpyAsync(dIn_, in, count * 4, hipMemcpyHostToDevice, stream_)
                               : hipMemcpy2DAsync(dIn_, width, in, (size_t)pitch * 4, width, rows, hipMemcpyHostToDevice, stream_);
    if (e != hipSuccess) return hipFail(e, "H2D");
    int rc = processDevice(dIn_, dOut_, nSamples, stream_);
    if (rc != 0) {
        (void)hipStreamSynchronize(stream_);   // (whatever went wrong: no copy may still read the caller's buffer when this returns)
        return rc;
    }
    e = pitch == n_ ? hipMemcpyAsync(out, dOut_, count * 4, hipMemcpyDeviceToHost, stream_)
                    : hipMemcpy2DAsync(out, (size_t)pitch * 4, dOut_, width, width, rows, hipMemcpyDeviceToHost, stream_);
    if (e == hipSuccess) e = hipStreamSynchronize(stream_);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(stream_);
        return hipFail(e, "D2H");
    }
    return 0;
}

// A large host block in pieces (dIn_ / dOut_ hold the whole block): while the kernel works on piece p, piece p + 1 is on its way
// in and piece p - 1 on its way out - two copy streams beside the compute stream, ordered by events.  The pieces are
// consecutive blocks to the kernel: state carries over exactly as between two calls.
int Batch::processHostPipelined(const float* in, float* out, int nSamples, int64_t pitch, int pieces) {
    if (!copyIn_) {
        hipError_t c = hipStreamCreateWithFlags(&copyIn_, hipStreamNonBlocking);
        if (c == hipSuccess) c = hipStreamCreateWithFlags(&copyOut_, hipStreamNonBlocking);
        for (int k = 0; k < kHostPieces && c == hipSuccess; ++k) {
            c = hipEventCreateWithFlags(&evIn_[k], hipEventDisableTiming);
            if (c == hipSuccess) c = hipEventCreateWithFlags(&evDone_[k], hipEventDisableTiming);
        }
        if (c != hipSuccess) return hipFail(c, "streams for the pipelined host block");
    }
    const size_t ch = (size_t)prog_.numChannels, width = (size_t)n_ * 4;
    auto lo = [&](int p) { return (int)((int64_t)nSamples * p / pieces); };
    auto copyIn = [&](int p) {
        const size_t first = (size_t)lo(p) * ch, rows = (size_t)(lo(p + 1) - lo(p)) * ch;
        hipError_t e = pitch == n_ ? hipMemcpyAsync(dIn_ + first * (size_t)n_, in + first * (size_t)n_, rows * width, hipMemcpyHostToDevice, copyIn_)
                                   : hipMemcpy2DAsync(dIn_ + first * (size_t)n_, width, in + first * (size_t)pitch, (size_t)pitch * 4, width, rows, hipMemcpyHostToDevice, copyIn_);
        if (e == hipSuccess) e = hipEventRecord(evIn_[p], copyIn_);
        return e;
    };
    auto launch = [&](int p) -> int {
        hipError_t e = hipStreamWaitEvent(stream_, evIn_[p], 0);
        if (e != hipSuccess) return hipFail(e, "pipelined host block");
        const size_t first = (size_t)lo(p) * ch * (size_t)n_;
        const int rc = processDevice(dIn_ + first, dOut_ + first, lo(p + 1) - lo(p), stream_);
        if (rc != 0) return rc;
        e = hipEventRecord(evDone_[p], stream_);
        return e == hipSuccess ? 0 : hipFail(e, "pipelined host block");
    };
    auto copyOut = [&](int p) {
        const size_t first = (size_t)lo(p) * ch, rows = (size_t)(lo(p + 1) - lo(p)) * ch;
        hipError_t e = hipStreamWaitEvent(copyOut_, evDone_[p], 0);
        if (e != hipSuccess) return e;
        return pitch == n_ ? hipMemcpyAsync(out + first * (size_t)n_, dOut_ + first * (size_t)n_, rows * width, hipMemcpyDeviceToHost, copyOut_)
                           : hipMemcpy2DAsync(out + first * (size_t)pitch, (size_t)pitch * 4, dOut_ + first * (size_t)n_, width, width, rows, hipMemcpyDeviceToHost, copyOut_);
    };
    // (whatever goes wrong: no copy may still touch the caller's buffers when this returns)
    auto drain = [&]() {
        (void)hipStreamSynchronize(copyIn_);
        (void)hipStreamSynchronize(stream_);
        (void)hipStreamSynchronize(copyOut_);
    };
    waitLastLaunch();
    // the block is ONE call to the bookkeeping of control changes and to the lowering (with its real length), not kHostPieces:
    // a translation must not fire between two pieces
    pendingSamples_ = nSamples / pieces;   // what the kernel is launched with: the class of block lengths is the piece's
    if (controlHeat_ > 0 && --controlHeat_ == 0 && c_.deferred) lowDirty_ = true;
    noteBlockLength(pendingSamples_);
    leanStep();
    sampleClock_ += nSamples;
    int rc = ensureLowered();
    if (rc != 0) return rc;
    piecewise_ = true;
    struct Reset { bool& f; ~Reset() { f = false; } } reset{piecewise_};
    hipError_t e = copyIn(0);
    if (e != hipSuccess) { drain(); return hipFail(e, "H2D"); }
    rc = launch(0);
    if (rc != 0) { drain(); return rc; }
    for (int p = 0; p < pieces; ++p) {
        if (p + 1 < pieces) {
            if ((e = copyIn(p + 1)) != hipSuccess) { drain(); return hipFail(e, "H2D"); }
            if ((rc = launch(p + 1)) != 0) { drain(); return rc; }
        }
        if ((e = copyOut(p)) != hipSuccess) { drain(); return hipFail(e, "D2H"); }
    }
    e = hipStreamSynchronize(copyOut_);
    if (e == hipSuccess) e = hipStreamSynchronize(stream_);
    return e == hipSuccess ? 0 : hipFail(e, "pipelined host block");
}

// Generate the code a stream of `nSamples`-sample blocks will run, now - a real-time caller does this after loading, before the
// stream starts, instead of paying for the translation in its first block.  wait: also until the builder thread has finished
// what it was asked for (the variant with the controls in rows, other stage counts on trial).
int Batch::prepare(int nSamples, bool wait) {
    (void)hipSetDevice(device_);
    if (nSamples < 1) return fail(FX_E_ARG, "prepare: n_samples >= 1");
    pendingSamples_ = nSamples;
    noteBlockLength(nSamples);
    int rc = ensureLowered();
    if (rc != 0) return rc;
    everLowered_ = true;
    leanStep();   // (controls have rows and some of them rest: their variant is asked for now)
    if (wait && builder_) {
        std::unique_lock<std::mutex> lock(builder_->mu);
        builder_->cv.wait(lock, [&] { return builder_->running.empty() && builder_->jobs.empty(); });
        lock.unlock();
        collectBuilt();
        leanStep();   // ... and in force when this returns
        rc = ensureLowered();
        if (rc != 0) return rc;
    }
    return 0;
}

int Batch::sync() {
    (void)hipSetDevice(device_);
    hipError_t e = hipStreamSynchronize(stream_);
    if (e == hipSuccess && launched_) e = hipEventSynchronize(ev1_);
    return e == hipSuccess ? 0 : hipFail(e, "sync");
}

// ---- state snapshot ------------------------------------------------------------------------------------------------------------
int Batch::snapshotShape(SnapshotHeader* hdr) {
    (void)hipSetDevice(device_);
    int rc = ensureLowered();
    if (rc != 0) return rc;
    *hdr = SnapshotHeader();
    hdr->n = n_;
    hdr->channels = prog_.numChannels;
    hdr->nRegs = (int32_t)prog_.regs.size();
    hdr->stateRows = stateRows_;
    hdr->iSlots = iSlotsAlloc_;
    hdr->xSlots = xSlotsAlloc_;
    return 0;
}

namespace {
// delay memory on the device: [wave][slot][64 * K] (K instances per lane: the HIP C++ tier), an instance i at
// wave i / (64 K), column i % (64 K); in an image: [instance][slot]
constexpr size_t kTramChunkBytes = (size_t)64 << 20;
}

int Batch::saveStateColumns(uint8_t* image, const SnapshotHeader& hdr, int64_t first) {
    (void)hipSetDevice(device_);
    if (!image || first < 0 || first + n_ > hdr.n) return fail(FX_E_ARG, "snapshot: bad image");
    SnapshotHeader mine;
    int rc = snapshotShape(&mine);
    if (rc != 0) return rc;
    if (mine.channels != hdr.channels || mine.nRegs != hdr.nRegs || mine.stateRows != hdr.stateRows || mine.iSlots != hdr.iSlots || mine.xSlots != hdr.xSlots)
        return fail(FX_E_ARG, "snapshot: the image was laid out for another program");
    waitLastLaunch();
    hipError_t e = hipStreamSynchronize(stream_);
    uint8_t* rows = image + sizeof(SnapshotHeader);
    if (e == hipSuccess)
        e = hipMemcpy2D(rows + (size_t)first * 4, (size_t)hdr.n * 4, dState_, (size_t)nPad_ * 4, (size_t)n_ * 4, (size_t)stateRows_, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hipFail(e, "snapshot: state rows");
    uint8_t* sec = rows + (size_t)hdr.stateRows * (size_t)hdr.n * 4;
    for (int which = 0; which < 2; ++which) {
        const float* dev = which ? dXTram_ : dITram_;
        const int slots = which ? xSlotsAlloc_ : iSlotsAlloc_;
        float* out = reinterpret_cast<float*>(sec);
        sec += (size_t)hdr.n * (size_t)slots * 4;
        if (slots == 0) continue;
        const size_t cols = 64 * (size_t)instPerLane_, waveFloats = (size_t)slots * cols;
        const size_t waves = ((size_t)n_ + cols - 1) / cols, chunk = std::max<size_t>(1, kTramChunkBytes / (waveFloats * 4));
        std::vector<float> tmp(std::min(waves, chunk) * waveFloats);
        for (size_t w0 = 0; w0 < waves; w0 += chunk) {
            const size_t nw = std::min(chunk, waves - w0);
            e = hipMemcpy(tmp.data(), dev + w0 * waveFloats, nw * waveFloats * 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return hipFail(e, "snapshot: delay memory");
            for (size_t w = 0; w < nw; ++w)
                for (size_t j = 0; j < cols; ++j) {
                    const size_t inst = (w0 + w) * cols + j;
                    if (inst >= (size_t)n_) break;
                    float* line = out + ((size_t)first + inst) * (size_t)slots;
                    const float* src = tmp.data() + w * waveFloats + j;
                    for (int s = 0; s < slots; ++s) line[s] = src[(size_t)s * cols];
                }
        }
    }
    return 0;
}

int Batch::loadStateColumns(const uint8_t* image, const SnapshotHeader& hdr, int64_t first) {
    (void)hipSetDevice(device_);
    if (!image || first < 0 || first + n_ > hdr.n) return fail(FX_E_ARG, "snapshot: bad image");
    SnapshotHeader mine;
    int rc = snapshotShape(&mine);
    if (rc != 0) return rc;
    if (hdr.magic != mine.magic || hdr.version != mine.version) return fail(FX_E_ARG, "snapshot: not a state image of this library version");
    // (every field is checked before any of them enters an address: the image may be a damaged file)
    if (hdr.iSlots < 0 || hdr.xSlots < 0 || hdr.n < n_ || mine.channels != hdr.channels || mine.nRegs != hdr.nRegs || mine.stateRows != hdr.stateRows ||
        hdr.iSlots > mine.iSlots || hdr.xSlots > mine.xSlots)
        return fail(FX_E_ARG, "snapshot: the image is of another program (registers, channels or delay lines differ)");
    waitLastLaunch();
    hipError_t e = hipStreamSynchronize(stream_);
    const uint8_t* rows = image + sizeof(SnapshotHeader);
    if (e == hipSuccess)
        e = hipMemcpy2D(dState_, (size_t)nPad_ * 4, rows + (size_t)first * 4, (size_t)hdr.n * 4, (size_t)n_ * 4, (size_t)stateRows_, hipMemcpyHostToDevice);
    if (e != hipSuccess) return hipFail(e, "snapshot: state rows");
    // what the host knows about the registers follows the image: a register every instance holds the same value of is a
    // broadcast write of that value, any other one a per-instance write (as if the caller had made them)
    coldSetChanged();
    for (int r = 0; r < hdr.nRegs; ++r) {
        if (tracked(r) || intrinsicLane(r)) continue;
        const uint32_t* row = reinterpret_cast<const uint32_t*>(rows + ((size_t)r * (size_t)hdr.n + (size_t)first) * 4);
        bool same = true;
        for (int64_t i = 1; i < n_ && same; ++i) same = row[i] == row[0];
        float v;
        std::memcpy(&v, &row[0], 4);
        if (same) {
            const bool moved = bitsOf(hostValue_[(size_t)r]) != row[0];
            hostValue_[(size_t)r] = v;
            if (forcedLane_[(size_t)r]) {
                if (laneWritten_[(size_t)r] && !(controlMode_ && declaredControl(r))) { forcedLane_[(size_t)r] = 0; lowDirty_ = true; }
            } else if (moved && readByProgram(r)) {
                lowDirty_ = true;   // (compiled in: the next block is generated with the image's value - loading a snapshot is not the audio path)
            }
            laneWritten_[(size_t)r] = 0;
        } else {
            if (!forcedLane_[(size_t)r] && readByProgram(r)) { forcedLane_[(size_t)r] = 1; lowDirty_ = true; }
            laneWritten_[(size_t)r] = 1;
        }
    }
    const uint8_t* sec = rows + (size_t)hdr.stateRows * (size_t)hdr.n * 4;
    for (int which = 0; which < 2; ++which) {
        float* dev = which ? dXTram_ : dITram_;
        const int alloc = which ? xSlotsAlloc_ : iSlotsAlloc_, slots = which ? hdr.xSlots : hdr.iSlots;
        const float* in = reinterpret_cast<const float*>(sec);
        sec += (size_t)hdr.n * (size_t)slots * 4;
        if (alloc == 0) continue;
        const size_t cols = 64 * (size_t)instPerLane_, waveFloats = (size_t)alloc * cols;
        const size_t waves = ((size_t)n_ + cols - 1) / cols, chunk = std::max<size_t>(1, kTramChunkBytes / (waveFloats * 4));
        std::vector<float> tmp(std::min(waves, chunk) * waveFloats);
        for (size_t w0 = 0; w0 < waves; w0 += chunk) {
            const size_t nw = std::min(chunk, waves - w0);
            std::fill(tmp.begin(), tmp.begin() + (long)(nw * waveFloats), 0.0f);
            for (size_t w = 0; w < nw; ++w)
                for (size_t j = 0; j < cols; ++j) {
                    const size_t inst = (w0 + w) * cols + j;
                    if (inst >= (size_t)n_) break;
                    const float* line = in + ((size_t)first + inst) * (size_t)slots;
                    float* dst = tmp.data() + w * waveFloats + j;
                    for (int s = 0; s < slots; ++s) dst[(size_t)s * cols] = line[s];
                }
            e = hipMemcpy(dev + w0 * waveFloats, tmp.data(), nw * waveFloats * 4, hipMemcpyHostToDevice);
            if (e != hipSuccess) return hipFail(e, "snapshot: delay memory");
        }
    }
    return 0;
}

int Batch::getTramAt(int which, int64_t inst, float* out, int nSlots) {
    (void)hipSetDevice(device_);
    if (inst < 0 || inst >= n_ || !out || nSlots < 0 || which < 0 || which > 1) return fail(FX_E_ARG, "get_tram: bad argument");
    int rc = ensureLowered();
    if (rc != 0) return rc;
    waitLastLaunch();
    const float* dev = which ? dXTram_ : dITram_;
    const int alloc = which ? xSlotsAlloc_ : iSlotsAlloc_;
    for (int s = 0; s < nSlots; ++s) out[s] = 0.0f;   // (slots the program cannot reach are not allocated: zero, as in a fresh reference object)
    const int take = std::min(nSlots, alloc);
    if (take <= 0) return 0;
    const size_t cols = 64 * (size_t)instPerLane_;
    const float* src = dev + ((size_t)inst / cols) * (size_t)alloc * cols + (size_t)inst % cols;
    hipError_t e = hipMemcpy2D(out, 4, src, cols * 4, 4, (size_t)take, hipMemcpyDeviceToHost);
    return e == hipSuccess ? 0 : hipFail(e, "get_tram");
}

int Batch::getCursorsAt(int64_t inst, int32_t out4[4]) {
    (void)hipSetDevice(device_);
    if (inst < 0 || inst >= n_ || !out4 || !dState_) return fail(FX_E_ARG, "get_cursors: bad argument");
    waitLastLaunch();
    hipError_t e = hipMemcpy2D(out4, 4, dState_ + (size_t)stateLayout_.cursorBase * nPad_ + inst, (size_t)nPad_ * 4, 4, 4, hipMemcpyDeviceToHost);
    return e == hipSuccess ? 0 : hipFail(e, "get_cursors");
}

int64_t Batch::instructionCounter() {
    (void)hipSetDevice(device_);
    if (!dState_) return 0;
    waitLastLaunch();
    unsigned long long zero[2] = {0, 0};
    unsigned long long* dSum = reinterpret_cast<unsigned long long*>(dScratch_);
    uint32_t* dOr = dScratch_ + 2;
    if (hipMemcpy(dSum, zero, 16, hipMemcpyHostToDevice) != hipSuccess) return -1;
    if (launchReduceRow(dState_, nPad_, n_, stateLayout_.countLo, stateLayout_.countHi, stateLayout_.oodRow, dSum, dOr, stream_) != hipSuccess) return -1;
    if (hipStreamSynchronize(stream_) != hipSuccess) return -1;
    unsigned long long res[2] = {0, 0};
    if (hipMemcpy(res, dSum, 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)res[0];
}

uint32_t Batch::oodFlags() {
    (void)hipSetDevice(device_);
    if (!dState_) return 0;
    waitLastLaunch();
    unsigned long long zero[2] = {0, 0};
    unsigned long long* dSum = reinterpret_cast<unsigned long long*>(dScratch_);
    uint32_t* dOr = dScratch_ + 2;
    if (hipMemcpy(dSum, zero, 16, hipMemcpyHostToDevice) != hipSuccess) return ~0u;
    if (launchReduceRow(dState_, nPad_, n_, stateLayout_.countLo, stateLayout_.countHi, stateLayout_.oodRow, dSum, dOr, stream_) != hipSuccess) return ~0u;
    if (hipStreamSynchronize(stream_) != hipSuccess) return ~0u;
    uint32_t res[4] = {0, 0, 0, 0};
    if (hipMemcpy(res, dSum, 16, hipMemcpyDeviceToHost) != hipSuccess) return ~0u;
    return res[2];
}

int64_t Batch::instructionCounterAt(int64_t inst) {
    (void)hipSetDevice(device_);
    if (!dState_ || inst < 0 || inst >= n_) return 0;
    waitLastLaunch();
    uint32_t lo = 0, hi = 0;
    (void)hipMemcpy(&lo, dState_ + (size_t)stateLayout_.countLo * nPad_ + inst, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&hi, dState_ + (size_t)stateLayout_.countHi * nPad_ + inst, 4, hipMemcpyDeviceToHost);
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}

float Batch::lastKernelMs() {
    (void)hipSetDevice(device_);
    if (!timed_) return -1.0f;
    if (hipEventSynchronize(ev1_) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, ev0_, ev1_) != hipSuccess) return -1.0f;
    return ms;
}

#ifdef FX_DIAGNOSTICS
int Batch::readEndStamps(uint32_t* out, int64_t nWords) {
    (void)hipSetDevice(device_);
    if (!out || nWords < 0) return fail(FX_E_ARG, "end stamps: bad argument");
    if (!dStamps_) return fail(FX_E_NOTREADY, "end stamps: no unstaged translated launch with FX_XLATE_ENDSTAMP set has run");
    waitLastLaunch();
    const size_t take = std::min((size_t)nWords, stampWords_);
    const hipError_t e = hipMemcpy(out, dStamps_, take * 4, hipMemcpyDeviceToHost);
    return e == hipSuccess ? (int)std::min<size_t>(take, 0x7fffffff) : hipFail(e, "end stamps");
}
#endif

// "translated to gfx950 code (fx_xlate_v128, 8 stages)" / "interpreter (fx_interp_v96): <why the translation failed>" / "HIP C++
// kernel: <why not an assembly tier>" - of the code in force (before the first block: nothing has been lowered yet)
std::string Batch::tierNote() const {
    // (never in the release library: fx_knobs.hpp)
    return kDiagnosticsBuild ? "DIAGNOSTICS BUILD (environment knobs may change or corrupt results): " + tierNotePlain() : tierNotePlain();
}

std::string Batch::tierNotePlain() const {
    static const char* const regs[ASM_VARIANTS] = {"lds", "v64", "v72", "v80", "v96", "v128", "v168", "v256"};
    if (!loaded_) return "no program loaded";
    if (c_.key.empty() && !c_.useAsm && c_.low.steady.empty()) return "not lowered yet (the first block, fxb_prepare or an fxb_info query does it)";
    if (c_.useAsm && c_.useXlate)
        return std::string("translated to gfx950 code (fx_xlate_") + regs[c_.variant] + (c_.stages > 1 ? ", " + std::to_string(c_.stages) + " stages" : "") +
               (c_.stages <= 1 && c_.prioritySlices ? ", wavefronts of a SIMD by turns" : "") + ")";
    if (c_.useAsm) return std::string("interpreter (fx_interp_") + regs[c_.variant] + "): " + (c_.xlateWhyNot.empty() ? "no translation asked for" : c_.xlateWhyNot);
    return "HIP C++ kernel (" + std::to_string(c_.low.instPerLane) + " instance(s) per lane): " + (c_.asmWhyNot.empty() ? "no assembly tier asked for" : c_.asmWhyNot);
}

int64_t Batch::info(int what) {
    if (what == FXB_INFO_DEVICE) return device_;
    if (what == FXB_INFO_NUM_INSTRUCTIONS) return (int64_t)prog_.instrs.size();
    if (what == FXB_INFO_NUM_REGISTERS) return (int64_t)prog_.regs.size();
    if (what == FXB_INFO_GRID) return lastGrid_;
    if (what == FXB_INFO_WAVES_PER_WG) return (c_.useAsm && c_.useXlate) ? c_.stages : 1;
    if (ensureLowered() != 0) return -1;
    switch (what) {
        case FXB_INFO_INST_PER_LANE: return instPerLane_;
        case FXB_INFO_KERNEL: return c_.useAsm ? (c_.useXlate ? 8 + (int)c_.variant : 1 + (int)c_.variant) : 0;
        case FXB_INFO_XLATE_CODE_BYTES: return c_.useXlate ? (int64_t)c_.codeBytes : 0;
        case FXB_INFO_XLATE_INLINED: return c_.useXlate ? c_.inlined : 0;
        case FXB_INFO_XLATE_CALLED: return c_.useXlate ? c_.called : 0;
        case FXB_INFO_XLATE_BUILDS: return xlateBuilds_;
        case FXB_INFO_XLATE_BACKGROUND_BUILDS: return backgroundBuilds_;
        case FXB_INFO_CONTROL_ROWS: {
            if (c_.key.empty() || lowDirty_) return -1;   // (nothing in force yet / about to change)
            int64_t rows = 0;
            for (size_t r = 0; r < declared_.size() && r < c_.low.rowOfReg.size(); ++r) rows += declared_[r] && !intrinsicLane((int)r) && c_.low.rowOfReg[r] >= 0;
            return rows;
        }
        case FXB_INFO_STAGE_TRIALS: { int64_t n = 0; for (const Tuner& t : tune_) n += t.trials; return n; }
        case FXB_INFO_XLATE_CODE_HASH: return c_.useXlate ? (int64_t)c_.codeHash : 0;
        case FXB_INFO_CODE_CACHE_HITS: return cacheHits_;
        case FXB_INFO_CODE_CACHED: return (int64_t)cache_.size() + (c_.key.empty() ? 0 : 1);
        case FXB_INFO_XLATE_UNSATURATED: return c_.useXlate ? c_.unsaturated : 0;
        case FXB_INFO_XLATE_VALU: return c_.useXlate ? c_.valu : 0;
        case FXB_INFO_XLATE_VALU_SLOW: return c_.useXlate ? c_.valuSlow : 0;
        case FXB_INFO_XLATE_VALU_CLOCKS: return c_.useXlate ? c_.valuClocks : 0;
        case FXB_INFO_XLATE_VGPR_CONSTANTS: return c_.useXlate ? c_.vgprConstants : 0;
        case FXB_INFO_NUM_LANE_REGS: return c_.low.nLaneRegs;
        case FXB_INFO_NUM_UNIFORM_REGS: return c_.low.nUniformRegs;
        case FXB_INFO_LDS_BYTES_PER_WG: return (c_.useAsm && c_.variant != ASM_LDS) ? (c_.useXlate ? (int64_t)c_.ldsBytes : 0) : (int64_t)c_.low.nRows * 256 * instPerLane_;
        case FXB_INFO_NUM_ROWS: return c_.low.nRows;
        case FXB_INFO_NUM_MICROOPS: return (int64_t)c_.low.steady.size();
        case FXB_INFO_ITRAM_SLOTS: return iSlotsAlloc_;
        case FXB_INFO_XTRAM_SLOTS: return xSlotsAlloc_;
        case FXB_INFO_TRAM_OPS: return c_.low.tramOpsPerSample;
        case FXB_INFO_MULTIPASS: return c_.low.multipass ? 1 : 0;
        case FXB_INFO_NUM_SHADOWED: return c_.low.nShadowed;
        case FXB_INFO_NUM_CCR_LIVE: return c_.low.nCcrLive;
        default: return -1;
    }
}

}  // namespace fx
