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

#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <stdexcept>

#include "../../include/fx8010_amd.h"

namespace fx {

namespace {
const Luts& sharedLuts() {
    static const Luts l;
    return l;
}
constexpr size_t kScratchBytes = 1 << 16;
inline uint32_t bitsOf(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
}  // namespace

Batch::Batch(int64_t nInstances, int channels, int device) : prog_(channels) {
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
    if (stream_) (void)hipStreamSynchronize(stream_);
    (void)hipFree(dState_);
    (void)hipFree(dITram_);
    (void)hipFree(dXTram_);
    (void)hipFree(dLut_);
    (void)hipFree(dStream_);
    (void)hipFree(dScratch_);
    (void)hipFree(dTracks_);
    (void)hipFree(dIn_);
    (void)hipFree(dOut_);
    if (hPinIn_) (void)hipHostFree(hPinIn_);
    if (hPinOut_) (void)hipHostFree(hPinOut_);
    for (int k = 0; k < kHostPieces; ++k) {
        if (evIn_[k]) (void)hipEventDestroy(evIn_[k]);
        if (evDone_[k]) (void)hipEventDestroy(evDone_[k]);
    }
    if (copyIn_) (void)hipStreamDestroy(copyIn_);
    if (copyOut_) (void)hipStreamDestroy(copyOut_);
    if (xlateModule_) (void)hipModuleUnload(xlateModule_);
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
    return fail(e == hipErrorOutOfMemory ? FX_E_MEMORY : FX_E_NODEVICE, std::string(where) + ": " + hipGetErrorString(e));
}

bool Batch::loadFile(const std::string& path) { return afterLoad(prog_.loadFile(path)) == 1; }
bool Batch::loadText(const std::string& text) { return afterLoad(prog_.loadText(text)) == 1; }

int Batch::afterLoad(bool ok) {
    (void)hipSetDevice(device_);
    // registers may have been created even when the load failed; keep host mirrors in step
    const size_t old = hostValue_.size();
    hostValue_.resize(prog_.regs.size());
    forcedLane_.resize(prog_.regs.size(), 0);
    for (size_t r = old; r < prog_.regs.size(); ++r) hostValue_[r] = prog_.regs[r].value;
    lowDirty_ = true;
    daneHipOnly_ = false;
    // a load that failed after an earlier good one has still appended registers (literals and declarations are created
    // before the error, as in the reference): the state block must follow, or set_register of a new one would land in
    // the rows behind the registers (output latches, cursors, LFSR, counter)
    if (!ok) {
        if (dState_) (void)ensureState();
        return 0;
    }
    loaded_ = true;
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

int Batch::ensureTram() {
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
    int rc = grow(dITram_, iSlotsAlloc_, low_.iSlots);
    if (rc == 0) rc = grow(dXTram_, xSlotsAlloc_, low_.xSlots);
    return rc;
}

// K = instances stepped by one lane.  More instances per lane amortise the scalar fetch/dispatch of
// a record over more work and widen every LDS access, but shrink the number of wavefronts; the LDS
// register file (rows * 256 * K bytes per wavefront) bounds how many wavefronts a CU can hold.
// Once TRAM has been allocated its [wave][slot][64][K] tiling pins K for the life of the batch.
int Batch::chooseInstPerLane() const {
    if (iSlotsAlloc_ > 0 || xSlotsAlloc_ > 0) return instPerLane_;
    if (const char* env = std::getenv("FX_INST_PER_LANE")) {
        const int k = std::atoi(env);
        if (k == 1 || k == 2 || k == 4) return k;
    }
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
    if ((reg < (int)forcedLane_.size() && forcedLane_[reg]) || tracked(reg)) return true;
    return !lowDirty_ ? low_.rowOfReg[reg] >= 0 : (reg < (int)low_.rowOfReg.size() && low_.rowOfReg[reg] >= 0);
}

// How many pipeline stages to ask the translator for; FX_STAGES pins the number (1 = never).  Measured with config2's filter
// chain at S = 2048 (tools/stage_probe.py, profiles/r03b_stage_policy.txt): 8 stages are worth x 3.1 / 2.7 / 1.7 at 16 / 256 / 512
// wavefronts of instances (up to two groups per CU), 4 stages + 30 % / + 8 % / + 7 % at 768 / 1 024 / 1 536 (one wavefront per SIMD
// becomes four, but every stage adds ~11 instructions per sample), and from 1 792 wavefronts on the plain program is faster.
int Batch::stagesWanted(int variant) const {
    if (const char* knob = std::getenv("FX_STAGES")) return std::max(1, std::min(16, std::atoi(knob)));
    (void)variant;
    const int64_t waves = (n_ + 63) / 64;
    if (waves <= 512) return 8;
    if (waves <= 1536) return 4;
    return 1;
}

int Batch::ensureLowered() {
    if (!loaded_ || !prog_.ready) return fail(FX_E_NOTREADY, "no program loaded");
    if (!lowDirty_) return 0;
    std::vector<int> before = low_.rowOfReg;
    // Preferred: the hand-written gfx950 interpreter (one instance per lane, bookkeeping in VGPRs).
    // Programs it does not cover run on the HIP C++ kernel.  TRAM tiling pins K once allocated.
    Lowered fresh;
    bool asmOk = false;
    const bool tramPinned = iSlotsAlloc_ > 0 || xSlotsAlloc_ > 0;
    const char* forceHip = std::getenv("FX_KERNEL");
    const bool wantAsm = !(forceHip && std::strcmp(forceHip, "hip") == 0) && !std::getenv("FX_INST_PER_LANE") && !daneHipOnly_;
    if (wantAsm && (!tramPinned || instPerLane_ == 1)) {
        // first choice: register file in VGPRs (row pitch 1 = plain indices), else in LDS
        const bool tryVgpr = !(forceHip && std::strcmp(forceHip, "asm_lds") == 0);
        if (tryVgpr) {
            fresh = lowerProgram(prog_, hostValue_, laneForced(), 1, false, 1);
            asmOk = fresh.error.empty() && asmEligible(fresh, &asmWhyNot_);
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
                if (stagesWanted(v) >= 2)
                    while (v < ASM_V128) ++v;
                const char* pin = forceHip ? std::strstr(forceHip, "_v") : nullptr;
                if (pin && (std::strncmp(forceHip, "asm_v", 5) == 0 || std::strncmp(forceHip, "xlate_v", 7) == 0)) {
                    // diagnostics: pin a (large enough) build of the interpreter (asm_vNN) or of the translator (xlate_vNN)
                    static const char* const tags[ASM_VARIANTS] = {"", "_v64", "_v72", "_v80", "_v96", "_v128", "_v168", "_v256"};
                    for (int q = smallest; q < ASM_VARIANTS; ++q)
                        if (std::strcmp(pin, tags[q]) == 0) v = q;
                }
                asmVariant_ = (AsmVariant)v;
            }
        }
        if (!asmOk) {
            fresh = lowerProgram(prog_, hostValue_, laneForced(), 1, false);
            asmOk = fresh.error.empty() && asmEligible(fresh, &asmWhyNot_);
            asmVariant_ = ASM_LDS;
        }
        // the opt-in DANE delay-line model exists as generated code (translated tier) and in the HIP C++ kernel only
        if (asmOk && fresh.tramDane && (asmVariant_ == ASM_LDS || (forceHip && std::strncmp(forceHip, "asm", 3) == 0))) {
            asmOk = false;
            asmWhyNot_ = "opt-in DANE delay-line model: no interpreter handlers";
        }
    } else {
        asmWhyNot_ = "disabled by FX_KERNEL / FX_INST_PER_LANE";
    }
    if (!asmOk) fresh = lowerProgram(prog_, hostValue_, laneForced(), chooseInstPerLane());
    if (!fresh.error.empty()) return fail(FX_E_PROGRAM, fresh.error);
    instPerLane_ = fresh.instPerLane;
    useAsm_ = asmOk;
    int rc = ensureState();
    if (rc != 0) return rc;
    // registers that were uniform and are per-instance from now on: seed their rows
    std::vector<uint32_t> rows, values;
    for (size_t r = 0; r < fresh.rowOfReg.size(); ++r) {
        const bool was = r < before.size() && before[r] >= 0;
        const bool forced = forcedLane_[r] != 0 || tracked((int)r);  // already seeded (every register's state row is kept valid)
        if (fresh.rowOfReg[r] >= 0 && !was && !forced) { rows.push_back((uint32_t)r); values.push_back(bitsOf(hostValue_[r])); }
    }
    low_ = std::move(fresh);
    // registers the program itself keeps per-instance (it writes them): a static property, valid until the next load
    intrinsicLane_.assign(low_.rowOfReg.size(), 0);
    for (size_t r = 0; r < low_.rowOfReg.size(); ++r) intrinsicLane_[r] = low_.rowOfReg[r] >= 0 && !forcedLane_[r] && !tracked((int)r);
    if (!rows.empty() && (rc = fillRows(rows, values)) != 0) return rc;
    if ((rc = ensureTram()) != 0) return rc;

    // upload: steady | last | row table
    useXlate_ = false;
    xlateStages_ = 1;
    xlateDeferred_ = false;
    // (a block of more than ~half a millisecond of translated code pays for its translation at once)
    const double blockMs = (double)n_ * (double)pendingSamples_ * (double)std::max(low_.staticCount, 1) / 1e10;
    if (useAsm_ && asmVariant_ != ASM_LDS && controlHeat_ > 0 && blockMs < 0.5 && !low_.tramDane && !(forceHip && std::strncmp(forceHip, "xlate", 5) == 0)) {
        // controls are moving (a set_register within the last few blocks): a translation costs a module load
        // (~1-2 ms), a re-encode for the interpreter ~0.05 ms - interpret until the controls have been quiet
        xlateDeferred_ = true;
        xlateWhyNot_ = "deferred: control registers are changing";
    } else if (useAsm_ && asmVariant_ != ASM_LDS && !(forceHip && std::strncmp(forceHip, "asm", 3) == 0)) {
        // first choice for a VGPR build: translate the program into gfx950 code (FX_KERNEL=asm* pins the interpreter)
        const std::vector<MicroOp> steadyRecords = encodeAsmStream(low_.steady, nullptr, true), lastRecords = encodeAsmStream(low_.last, nullptr, true);
        std::vector<int> trackRows;
        for (int reg : trackRegs_) trackRows.push_back(low_.rowOfReg[(size_t)reg]);
        XlateProgram xprog = xlateProgramOf(steadyRecords, lastRecords, prog_.iTramSize, prog_.xTramSize, low_.nRows, low_.inRow, low_.latchRow, trackRows);
        // 256 bytes per wavefront and slot; the Infinity Cache holds 256 MiB
        xprog.tramStreaming = ((size_t)iSlotsAlloc_ + (size_t)xSlotsAlloc_) * (((size_t)n_ + 63) / 64) * 256 > ((size_t)512 << 20);
        XlateImage image;
        const XlateTemplate* tmpl = nullptr;
        bool built = false;
        tmpl = xlateTemplate(asmVariant_, &xlateWhyNot_);
        // Small batches leave SIMDs empty (and a lone wavefront issues an instruction every ~4.5 clocks): cut the program
        // into stages run by the wavefronts of one workgroup (fx_xlate.hpp StageInfo) until ~4 wavefronts per SIMD are in
        // flight.  FX_STAGES pins the number asked for (1 = never).
        int wantStages = stagesWanted((int)asmVariant_);
        // Measured with config2 at 4 096 instances (profiles/r03b_stage_blocks.txt): a block of 32 samples takes 27 us unstaged, 32 us
        // in 8 stages with a barrier every 8 samples (3 x 7 steps of 8 samples to fill and drain) and 21 us in 4 stages with a
        // barrier per sample; 128 samples 69 / 48 / 40 us (8 stages, every 2 samples); from 256 samples on the long steps win
        const int blockClass = stageBlockClass(std::max(pendingSamples_, 1));
        const int maxGroup = blockClass == 0 ? 1 : (blockClass == 1 ? 2 : kStageGroupMax);
        if (blockClass == 0 && !std::getenv("FX_STAGES")) wantStages = std::min(wantStages, 4);
        stagedForClass_ = blockClass;
        otherClassBlocks_ = 0;
        stagesWhyNot_.clear();
        if (tmpl && wantStages >= 2) {
            const StagePlan plan = planStages(steadyRecords, lastRecords, xprog, low_.nRows, wantStages);
            stagesWhyNot_ = plan.why;
            std::string why;
            if (!plan.cuts.empty()) {
                // (several workgroups per CU must fit its 160 KiB of LDS together)
                const int64_t groupsPerCu = std::max<int64_t>(1, ((n_ + 63) / 64 + 255) / 256);
                const uint32_t ldsBudget = (uint32_t)std::min<int64_t>(144 * 1024, 160 * 1024 / groupsPerCu - 256);
                built = buildStagedImage(steadyRecords, lastRecords, *tmpl, xprog, plan, &image, nullptr, nullptr, &why, ldsBudget, maxGroup);
                if (!built) { stagesWhyNot_ = why; image = XlateImage(); }
            }
        }
        if (!built) built = tmpl && buildXlateImage(steadyRecords, lastRecords, *tmpl, xprog, &image, &xlateWhyNot_);
        if (built) {
            waitLastLaunch();  // the previous launch may still run the old code
            if (xlateModule_) (void)hipModuleUnload(xlateModule_);
            xlateModule_ = nullptr;
            xlateFn_ = nullptr;
            hipError_t me = hipModuleLoadData(&xlateModule_, image.elf.data());
            if (me == hipSuccess) me = hipModuleGetFunction(&xlateFn_, xlateModule_, tmpl->kernelName.c_str());
            if (me != hipSuccess) return hipFail(me, "loading the translated program");
            xlateSteady_ = (uint64_t)image.steadyFastOff | ((uint64_t)image.steadyOff << 32);
            xlateLast_ = (uint64_t)image.lastFastOff | ((uint64_t)image.lastOff << 32);
            xlateCodeBytes_ = image.codeBytes;
            xlateInitOff_ = image.initOff;
            xlateLdsBytes_ = image.ldsBytes;
            xlateWildRow_ = image.wildRow;
            xlateUnsaturated_ = image.steady.unsaturated;
            xlateInlined_ = image.steady.inlined;
            xlateCalled_ = image.steady.called;
            xlateValu_ = image.steady.valu;
            xlateValuSlow_ = image.steady.valuSlow;
            xlateValuClocks_ = image.steady.valuClocks;
            xlateVgprConstants_ = image.vgprConstants;
            ++xlateBuilds_;
            xlateStages_ = image.stages;
            xlateStageDesc_ = image.stageDesc;
            xlateStageStoreRows_ = image.stageStoreRows;
            useXlate_ = true;
        }
    }
    if (useAsm_ && !useXlate_ && low_.tramDane) {  // (translation failed: e.g. code larger than the hole) -> HIP C++ kernel
        daneHipOnly_ = true;
        return ensureLowered();
    }
    if (useAsm_ && !useXlate_) {
        hipError_t pe = hipSuccess;
        const uint64_t* handlers = asmHandlerTable(asmVariant_, device_, &pe);
        if (!handlers) return hipFail(pe, "probe of the assembly interpreter");
        const bool fold = asmVariant_ != ASM_LDS;
        low_.steady = encodeAsmStream(low_.steady, handlers, fold);
        low_.last = encodeAsmStream(low_.last, handlers, fold);
    }
    const size_t nOps = low_.steady.size();
    const bool staged = useXlate_ && xlateStages_ > 1;
    const size_t words = nOps * 8 * 2 + low_.loadRows.size() + low_.storeRows.size() + low_.zeroRows.size() + (staged ? (size_t)xlateStages_ * 8 : 0);
    if (words > streamCap_) {
        waitLastLaunch();
        (void)hipFree(dStream_);
        dStream_ = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&dStream_), words * 4 + 256);
        if (e != hipSuccess) return hipFail(e, "hipMalloc stream");
        streamCap_ = words;
    }
    std::vector<uint32_t> host(words);
    std::memcpy(host.data(), low_.steady.data(), nOps * 32);
    std::memcpy(host.data() + nOps * 8, low_.last.data(), nOps * 32);
    size_t p = nOps * 16;
    for (const RowCopy& rcp : low_.loadRows) {
        // translated programs: bit 15 marks a row of the BOUNDED class (its state value is checked against 1.0)
        const bool bounded = useXlate_ && rcp.ldsRow < xlateWildRow_.size() && !xlateWildRow_[rcp.ldsRow];
        host[p++] = rcp.ldsRow | (bounded ? 0x8000u : 0u) | ((uint32_t)rcp.stateRow << 16);
    }
    if (staged) {
        // the store rows grouped by the stage that owns them; each stage's descriptor names its slice
        for (int k = 0; k < xlateStages_; ++k) {
            xlateStageDesc_[(size_t)k].storeFirst = (uint32_t)(p - (nOps * 16 + low_.loadRows.size()));
            uint32_t count = 0;
            for (const RowCopy& rcp : low_.storeRows) {
                const std::vector<int>& mine = xlateStageStoreRows_[(size_t)k];
                if (std::find(mine.begin(), mine.end(), (int)rcp.ldsRow) == mine.end()) continue;
                host[p++] = rcp.ldsRow | ((uint32_t)rcp.stateRow << 16);
                ++count;
            }
            xlateStageDesc_[(size_t)k].storeCount = count;
        }
        if (p != nOps * 16 + low_.loadRows.size() + low_.storeRows.size()) return fail(FX_E_PROGRAM, "internal: a store row without a stage");
    } else {
        for (const RowCopy& rcp : low_.storeRows) host[p++] = rcp.ldsRow | ((uint32_t)rcp.stateRow << 16);
    }
    for (int zr : low_.zeroRows) host[p++] = (uint32_t)zr;
    if (staged) {
        static_assert(sizeof(StageDescriptor) == 32, "StageDescriptor layout");
        std::memcpy(host.data() + p, xlateStageDesc_.data(), (size_t)xlateStages_ * 32);
        p += (size_t)xlateStages_ * 8;
    }
    waitLastLaunch();  // the previous launch may still read the old stream
    hipError_t e = hipMemcpy(dStream_, host.data(), words * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) return hipFail(e, "stream upload");
    lowDirty_ = false;
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

int Batch::setRegister(const std::string& key, float v) {
    (void)hipSetDevice(device_);
    const int r = prog_.findRegister(key);
    if (r < 0) return 1;
    hostValue_[r] = v;
    if (tracked(r) || forcedLane_[r] || intrinsicLane(r)) {
        // the register lives in a row (a schedule, an earlier per-instance or moving-control write, or the program writes it):
        // the fill below is all there is to do
    } else if (loaded_ && everLowered_ && movableControl(r)) {
        forcedLane_[r] = 1;  // a moving control: a row from now on (one re-lowering, this one)
        lowDirty_ = true;
    } else {
        lowDirty_ = true;    // immediates (and possibly the classification) change
        if (loaded_ && everLowered_) controlHeat_ = kHeatPerChange;
    }
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
    if (!forcedLane_[r] && !intrinsicLane(r)) {  // (the row is valid, see setRegister; the next lowering keeps the register per-lane)
        forcedLane_[r] = 1;
        lowDirty_ = true;
    }
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
    if (!forcedLane_[r] && !intrinsicLane(r)) {  // from now on a per-instance row (every lane is overwritten below)
        forcedLane_[r] = 1;
        lowDirty_ = true;
    }
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

std::vector<uint8_t> Batch::laneForced() const {
    std::vector<uint8_t> f = forcedLane_;
    for (int reg : trackRegs_)
        if ((size_t)reg < f.size()) f[(size_t)reg] = 1;
    return f;
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
        } else {
            std::memcpy(&trackStage_[at], t.values.data(), (size_t)used[k] * 4);
            at += (size_t)used[k];
            hostValue_[(size_t)reg] = t.values[(size_t)used[k] - 1];  // what every instance holds after the block
            forcedLane_[(size_t)reg] = 0;
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

int Batch::processDevice(const float* dIn, float* dOut, int nSamples, hipStream_t stream) {
    (void)hipSetDevice(device_);
    if (nSamples < 0) return fail(FX_E_ARG, "n_samples < 0");
    if (!piecewise_) {   // (a piece of a pipelined host block: done once for the whole block)
        pendingSamples_ = nSamples;
        if (controlHeat_ > 0 && --controlHeat_ == 0 && xlateDeferred_) lowDirty_ = true;  // quiet again: translate
        // a staged program is generated for a class of block lengths: when the caller has moved to another one for good, again
        if (useXlate_ && xlateStages_ > 1 && !lowDirty_ && nSamples > 0) {
            otherClassBlocks_ = stageBlockClass(nSamples) == stagedForClass_ ? 0 : otherClassBlocks_ + 1;
            if (otherClassBlocks_ >= 4) lowDirty_ = true;
        }
    }
    int rc = ensureLowered();
    if (rc != 0) return rc;
    everLowered_ = true;
    if (nSamples == 0) return 0;
    if (!dIn || !dOut) return fail(FX_E_ARG, "null buffer");
    if (tracksArmed() && !useXlate_) return processWithTrackFallback(dIn, dOut, nSamples, stream);
    hipStream_t s = pick(stream);
    if (useXlate_ && !trackRegs_.empty() && (rc = uploadTracks(nSamples, s)) != 0) return rc;
    KernelArgs a{};
    const size_t nOps = low_.steady.size();
    a.steady = dStream_;
    a.last = dStream_ + nOps * 8;
    a.rowTable = dStream_ + nOps * 16;
    a.state = dState_;
    a.in = dIn;
    a.out = dOut;
    a.itram = dITram_;
    a.xtram = dXTram_;
    a.lut = dLut_;
    a.n = n_;
    a.nPad = nPad_;
    a.nOps = (int)nOps;
    a.nLoad = (int)low_.loadRows.size();
    a.nStore = (int)low_.storeRows.size();
    a.nSamples = nSamples;
    a.channels = prog_.numChannels;
    for (int c = 0; c < kMaxChannels; ++c) {
        a.inRow[c] = c < prog_.numChannels ? low_.inRow[c] : -1;
        a.latchRow[c] = c < prog_.numChannels ? low_.latchRow[c] : 0;
    }
    a.iSlots = iSlotsAlloc_;
    a.xSlots = xSlotsAlloc_;
    a.iSize = prog_.iTramSize;
    a.xSize = prog_.xTramSize;
    a.nZero = (int)low_.zeroRows.size();
    const uint32_t rowBytes = 256u * (uint32_t)instPerLane_;
    a.skipOff = low_.skipRow >= 0 ? (uint32_t)low_.skipRow * rowBytes : 0;
    a.cursorOff = low_.cursorRow >= 0 ? (uint32_t)low_.cursorRow * rowBytes : 0;
    a.noiseOff = low_.noiseRow >= 0 ? (uint32_t)low_.noiseRow * rowBytes : 0;
    a.oodOff = low_.oodRow >= 0 ? (uint32_t)low_.oodRow * rowBytes : 0;
    a.aliveOff = low_.aliveRow >= 0 ? (uint32_t)low_.aliveRow * rowBytes : 0;
    a.hasShadow = low_.skipRow >= 0 ? 1 : 0;
    a.instPerLane = instPerLane_;
    a.tramDane = (low_.tramDane && low_.cursorRow >= 0) ? 1 : 0;
    a.oodRow = stateLayout_.oodRow;
    a.countLo = stateLayout_.countLo;
    a.countHi = stateLayout_.countHi;
    a.staticCount = low_.staticCount;
    a.nRows = low_.nRows;
    hipError_t e = untimed_ ? hipSuccess : hipEventRecord(ev0_, s);
    if (e == hipSuccess) {
        if (useAsm_) {
            AsmArgs g{};
            g.steady = a.steady; g.last = a.last; g.rowTable = a.rowTable; g.state = a.state;
            g.in = a.in; g.out = a.out; g.itram = a.itram; g.xtram = a.xtram; g.lut = a.lut;
            g.n = a.n; g.nPad = a.nPad; g.nLoad = a.nLoad; g.nStore = a.nStore;
            g.nSamples = a.nSamples; g.channels = a.channels;
            for (int c = 0; c < kMaxChannels; ++c) {
                g.inOff[c] = a.inRow[c] >= 0 ? a.inRow[c] * (int)low_.rowPitch : -1;
                g.latchOff[c] = a.latchRow[c] * (int)low_.rowPitch;
            }
            g.iSlots = a.iSlots; g.xSlots = a.xSlots; g.iSize = a.iSize; g.xSize = a.xSize;
            g.cursorRow = stateLayout_.cursorBase; g.noiseRow = stateLayout_.noiseBase;
            g.oodRow = a.oodRow; g.countLo = a.countLo; g.countHi = a.countHi; g.staticCount = a.staticCount;
            g.lutX1Off = kLutX1Off * 8;
            if (useXlate_) {
                // code streams are named by their byte offset from the kernel entry: {fast, exact} per argument
                g.steady = reinterpret_cast<const uint32_t*>((uintptr_t)xlateSteady_);
                g.last = reinterpret_cast<const uint32_t*>((uintptr_t)xlateLast_);
                g.initOff = (int)xlateInitOff_;
                g.tracks = trackRegs_.empty() ? nullptr : dTracks_;
                if (xlateStages_ > 1) {
                    g.stages = dStream_ + nOps * 16 + low_.loadRows.size() + low_.storeRows.size() + low_.zeroRows.size();
                    g.nStages = xlateStages_;
                }
                e = launchAsmFunction(xlateFn_, g, (unsigned)((n_ + 63) / 64), xlateLdsBytes_, s, (unsigned)xlateStages_);
            } else {
                e = launchAsmInterp(g, asmVariant_, asmVariant_ == ASM_LDS ? (size_t)a.nRows * 256 : 0, device_, s);
            }
        } else {
            e = launchStepBlock(a, low_.multipass, s);
        }
    }
    if (e == hipSuccess && !untimed_) e = hipEventRecord(ev1_, s);
    if (e != hipSuccess) return hipFail(e, "launch fx_step_block");
    launched_ = !untimed_;  // (an untimed launch is synchronised by its caller before anything else happens)
    timed_ = !untimed_;
    lastGrid_ = (unsigned)((n_ + 64 * instPerLane_ - 1) / (64 * instPerLane_));
    return 0;
}

// pitch: instances per PCM row of the HOST buffers (>= n_); a shard of a larger batch reads / writes its columns of the
// caller's [sample][channel][all instances] arrays in place (fx_shard.cpp)
namespace {
constexpr size_t kPinnedFloats = 512;
constexpr size_t kPipelinedBytes = (size_t)32 << 20;
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
    static const bool pipelineOff = std::getenv("FX_HOST_PIPELINE") && std::atoi(std::getenv("FX_HOST_PIPELINE")) == 0;  // diagnostics
    if (count * 4 >= kPipelinedBytes && nSamples >= 2 * kHostPieces && !tracksArmed() && !pipelineOff)
        return processHostPipelined(in, out, nSamples, pitch);
    const size_t rows = (size_t)nSamples * prog_.numChannels, width = (size_t)n_ * 4;
    hipError_t e = pitch == n_ ? hipMemcpyAsync(dIn_, in, count * 4, hipMemcpyHostToDevice, stream_)
                               : hipMemcpy2DAsync(dIn_, width, in, (size_t)pitch * 4, width, rows, hipMemcpyHostToDevice, stream_);
    if (e != hipSuccess) return hipFail(e, "H2D");
    int rc = processDevice(dIn_, dOut_, nSamples, stream_);
    if (rc != 0) return rc;
    e = pitch == n_ ? hipMemcpyAsync(out, dOut_, count * 4, hipMemcpyDeviceToHost, stream_)
                    : hipMemcpy2DAsync(out, (size_t)pitch * 4, dOut_, width, width, rows, hipMemcpyDeviceToHost, stream_);
    if (e == hipSuccess) e = hipStreamSynchronize(stream_);
    if (e != hipSuccess) return hipFail(e, "D2H");
    return 0;
}

// A large host block in pieces (dIn_ / dOut_ hold the whole block): while the kernel works on piece p, piece p + 1 is on its way
// in and piece p - 1 on its way out - two copy streams beside the compute stream, ordered by events.  The pieces are
// consecutive blocks to the kernel: state carries over exactly as between two calls.
int Batch::processHostPipelined(const float* in, float* out, int nSamples, int64_t pitch) {
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
    auto lo = [&](int p) { return (int)((int64_t)nSamples * p / kHostPieces); };
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
    pendingSamples_ = nSamples;
    if (controlHeat_ > 0 && --controlHeat_ == 0 && xlateDeferred_) lowDirty_ = true;
    int rc = ensureLowered();
    if (rc != 0) return rc;
    piecewise_ = true;
    struct Reset { bool& f; ~Reset() { f = false; } } reset{piecewise_};
    hipError_t e = copyIn(0);
    if (e != hipSuccess) { drain(); return hipFail(e, "H2D"); }
    rc = launch(0);
    if (rc != 0) { drain(); return rc; }
    for (int p = 0; p < kHostPieces; ++p) {
        if (p + 1 < kHostPieces) {
            if ((e = copyIn(p + 1)) != hipSuccess) { drain(); return hipFail(e, "H2D"); }
            if ((rc = launch(p + 1)) != 0) { drain(); return rc; }
        }
        if ((e = copyOut(p)) != hipSuccess) { drain(); return hipFail(e, "D2H"); }
    }
    e = hipStreamSynchronize(copyOut_);
    if (e == hipSuccess) e = hipStreamSynchronize(stream_);
    return e == hipSuccess ? 0 : hipFail(e, "pipelined host block");
}

int Batch::sync() {
    (void)hipSetDevice(device_);
    hipError_t e = hipStreamSynchronize(stream_);
    if (e == hipSuccess && launched_) e = hipEventSynchronize(ev1_);
    return e == hipSuccess ? 0 : hipFail(e, "sync");
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

int64_t Batch::info(int what) {
    if (what == FXB_INFO_DEVICE) return device_;
    if (what == FXB_INFO_NUM_INSTRUCTIONS) return (int64_t)prog_.instrs.size();
    if (what == FXB_INFO_NUM_REGISTERS) return (int64_t)prog_.regs.size();
    if (what == FXB_INFO_GRID) return lastGrid_;
    if (what == FXB_INFO_WAVES_PER_WG) return (useAsm_ && useXlate_) ? xlateStages_ : 1;
    if (ensureLowered() != 0) return -1;
    switch (what) {
        case FXB_INFO_INST_PER_LANE: return instPerLane_;
        case FXB_INFO_KERNEL: return useAsm_ ? (useXlate_ ? 8 + (int)asmVariant_ : 1 + (int)asmVariant_) : 0;
        case FXB_INFO_XLATE_CODE_BYTES: return useXlate_ ? (int64_t)xlateCodeBytes_ : 0;
        case FXB_INFO_XLATE_INLINED: return useXlate_ ? xlateInlined_ : 0;
        case FXB_INFO_XLATE_CALLED: return useXlate_ ? xlateCalled_ : 0;
        case FXB_INFO_XLATE_BUILDS: return xlateBuilds_;
        case FXB_INFO_XLATE_UNSATURATED: return useXlate_ ? xlateUnsaturated_ : 0;
        case FXB_INFO_XLATE_VALU: return useXlate_ ? xlateValu_ : 0;
        case FXB_INFO_XLATE_VALU_SLOW: return useXlate_ ? xlateValuSlow_ : 0;
        case FXB_INFO_XLATE_VALU_CLOCKS: return useXlate_ ? xlateValuClocks_ : 0;
        case FXB_INFO_XLATE_VGPR_CONSTANTS: return useXlate_ ? xlateVgprConstants_ : 0;
        case FXB_INFO_NUM_LANE_REGS: return low_.nLaneRegs;
        case FXB_INFO_NUM_UNIFORM_REGS: return low_.nUniformRegs;
        case FXB_INFO_LDS_BYTES_PER_WG: return (useAsm_ && asmVariant_ != ASM_LDS) ? (useXlate_ ? (int64_t)xlateLdsBytes_ : 0) : (int64_t)low_.nRows * 256 * instPerLane_;
        case FXB_INFO_NUM_ROWS: return low_.nRows;
        case FXB_INFO_NUM_MICROOPS: return (int64_t)low_.steady.size();
        case FXB_INFO_ITRAM_SLOTS: return iSlotsAlloc_;
        case FXB_INFO_XTRAM_SLOTS: return xSlotsAlloc_;
        case FXB_INFO_TRAM_OPS: return low_.tramOpsPerSample;
        case FXB_INFO_MULTIPASS: return low_.multipass ? 1 : 0;
        case FXB_INFO_NUM_SHADOWED: return low_.nShadowed;
        case FXB_INFO_NUM_CCR_LIVE: return low_.nCcrLive;
        default: return -1;
    }
}

}  // namespace fx
