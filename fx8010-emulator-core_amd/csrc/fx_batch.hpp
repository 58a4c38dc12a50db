// fx_batch.hpp — host engine behind the C ABI: N instances of one program on one GPU.
#pragma once

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "fx_asm.hpp"
#include "fx_xlate.hpp"
#include "fx_decode.hpp"
#include "fx_kernel.hpp"
#include "fx_knobs.hpp"
#include "fx_model.hpp"

namespace fx {

class Batch {
public:
    // throws std::runtime_error when no HIP device is usable (there is no CPU fallback)
    Batch(int64_t nInstances, int channels, int device);
    ~Batch();
    Batch(const Batch&) = delete;
    Batch& operator=(const Batch&) = delete;

    bool loadFile(const std::string& path);
    bool loadText(const std::string& text);

    int setRegister(const std::string& key, float v);                    // 0 found, 1 not found, <0 error
    int setRegisterAt(const std::string& key, int64_t inst, float v);
    float getRegisterAt(const std::string& key, int64_t inst);           // 1.0f when not found
    int setRegisterArray(const std::string& key, const float* values);   // values[n]
    int getRegisterArray(const std::string& key, float* values);
    int seedNoiseAt(int64_t inst, int32_t x1, int32_t x2);
    // A schedule of values for register `key`, applied by the NEXT process call: at sample s of that block, when s is a
    // multiple of `period`, the register takes values[s / period] (perInstance: values[(s / period) * pitch + instance];
    // pitch 0 = n) - what a caller of the reference does with setRegisterValue() between process() calls
    // (source/main.cpp:107-114), without cutting the block.  At most kMaxTracks registers; one-shot.
    int setRegisterTrack(const std::string& key, const float* values, int nSteps, int period, bool perInstance, int64_t pitch = 0);

    int processHost(const float* in, float* out, int nSamples, int64_t pitch = 0);  // synchronous; pitch: instances per host PCM row (0 = n)
    int processDevice(const float* dIn, float* dOut, int nSamples, hipStream_t stream);
    int sync();
    int prepare(int nSamples, bool wait);   // generate the code for blocks of this length now (and wait for the builder thread)

    // State snapshot (the reference keeps all DSP state in plain members, include/FX8010.h:162-217, 288-291: registers, output
    // latches, delay memory and its four positions, the LFSR words, the instruction counter).  The image is laid out by GLOBAL
    // instance, so a batch saved from one partition can be loaded into another: this batch's instances are columns
    // [first, first + n) of an image of nTotal instances.  Sections behind the header: state rows [stateRows][nTotal] u32,
    // iTRAM [nTotal][iSlots] f32, xTRAM [nTotal][xSlots] f32.
    struct SnapshotHeader {
        uint32_t magic = 0x54535846u, version = 1;   // "FXST"
        int64_t n = 0;
        int32_t channels = 0, nRegs = 0, stateRows = 0, iSlots = 0, xSlots = 0, reserved[7] = {};
    };
    static_assert(sizeof(SnapshotHeader) == 64, "SnapshotHeader layout");
    int snapshotShape(SnapshotHeader* hdr);   // what this batch would save (lowers the program first); n = this batch's instances
    // bytes of an image with this header; -1 for a header no batch could have written (a damaged file: negative or absurd
    // counts - nothing of it may enter an address computation)
    static int64_t snapshotBytes(const SnapshotHeader& hdr) {
        constexpr int64_t kMaxRows = 1 << 20, kMaxSlots = 1 << 20, kMaxInstances = (int64_t)1 << 40;
        if (hdr.n < 1 || hdr.n > kMaxInstances || hdr.stateRows < 1 || hdr.stateRows > kMaxRows || hdr.nRegs < 0 || hdr.nRegs > hdr.stateRows ||
            hdr.channels < 1 || hdr.channels > 4 || hdr.iSlots < 0 || hdr.iSlots > kMaxSlots || hdr.xSlots < 0 || hdr.xSlots > kMaxSlots)
            return -1;
        const uint64_t words = (uint64_t)hdr.stateRows + (uint64_t)hdr.iSlots + (uint64_t)hdr.xSlots;   // <= 2^20 + 2^21
        return (int64_t)(sizeof(SnapshotHeader) + (uint64_t)hdr.n * 4u * words);                            // < 2^40 * 2^24: no overflow
    }
    int saveStateColumns(uint8_t* image, const SnapshotHeader& hdr, int64_t first);
    int loadStateColumns(const uint8_t* image, const SnapshotHeader& hdr, int64_t first);
    // one instance's delay memory as the reference holds it (which: 0 smallDelayBuffer, 1 largeDelayBuffer), and its positions
    int getTramAt(int which, int64_t inst, float* out, int nSlots);
    int getCursorsAt(int64_t inst, int32_t out4[4]);

    int64_t instructionCounter();
    int64_t instructionCounterAt(int64_t inst);
    uint32_t oodFlags();
    float lastKernelMs();
    int64_t info(int what);
    std::string tierNote() const;   // which tier runs the program as it stands, and why not a faster one (fxb_tier_note)

    const Program& program() const { return prog_; }
    int64_t instances() const { return n_; }
    int device() const { return device_; }
    const std::string& lastError() const { return lastError_; }
    int channels() const { return prog_.numChannels; }        // fixed at construction: PCM layout of process()
    int loaderChannels() const { return prog_.loaderChannels; }  // reference getChannels()
    // reference setChannels(): only the loader's I/O-index bound changes (include/FX8010.h:73, source/FX8010.cpp:447)
    void setChannels(int c) { prog_.loaderChannels = c; }
    void noteError(const std::string& what) { lastError_ = what; }
    // behaviour beyond the reference (fx_model.hpp kOpt*): takes effect for programs loaded afterwards
    int setOption(unsigned option, bool on);

private:
    std::string tierNotePlain() const;
    int fail(int code, const std::string& what);
    int hipFail(hipError_t e, const char* where);
    int afterLoad(bool ok);
    int ensureLowered();          // (re)lower + upload the stream when dirty
    int ensureState();            // allocate / grow the state block for the current register count
    int ensureTram(const Lowered& low);
    int uploadTracks(int nSamples, hipStream_t s);   // translated tier: header + values -> dTracks_
    int processWithTrackFallback(const float* dIn, float* dOut, int nSamples, hipStream_t stream);  // other tiers: cut the block
    bool tracked(int reg) const;
    std::vector<uint8_t> laneForced() const;      // the registers with rows in the code that is wanted now: laneForcedFull(), minus the controls a lean variant in force has folded in
    std::vector<uint8_t> laneForcedFull() const;  // forcedLane_ plus the trackable registers
    std::vector<uint8_t> coldControls() const;    // the declared controls that have a row only for company and could be folded into the code (coldControl)
    std::vector<uint8_t> forcedWithout(const std::vector<uint8_t>& folded) const;   // laneForcedFull() minus those
    int fillRows(const std::vector<uint32_t>& rows, const std::vector<uint32_t>& values);
    bool laneResident(int reg) const;
    bool intrinsicLane(int reg) const;
    int chooseInstPerLane() const;
    hipStream_t pick(hipStream_t s) const { return s ? s : stream_; }
    void waitLastLaunch();        // host waits for the most recent kernel (an event of ours, not the caller's stream handle)

    Program prog_;
    const ReleaseKnobs knobs_;          // the environment's release knobs as they were when the handle was created (fx_knobs.hpp)
    std::vector<float> hostValue_;      // current value of every register as the host knows it
    std::vector<uint8_t> forcedLane_;   // registers that live in a row of the register file although no instruction writes them:
                                        // per-instance values (setRegisterAt / setRegisterArray / a per-instance schedule) or a moving control
    std::vector<uint8_t> laneWritten_;  // ... of those, the ones whose instances may really hold different values (a broadcast write clears it)
    bool lowDirty_ = true;
    bool loaded_ = false;

    int64_t n_ = 0, nPad_ = 0;
    int device_ = 0;
    hipStream_t stream_ = nullptr;
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
    bool launched_ = false;  // ev1_ marks the end of the most recent launch (on whatever stream it ran)
    bool timed_ = false;

    uint32_t* dState_ = nullptr;
    StateLayout stateLayout_;
    int stateRows_ = 0;
    float* dITram_ = nullptr;
    float* dXTram_ = nullptr;
    int iSlotsAlloc_ = 0, xSlotsAlloc_ = 0;
    int instPerLane_ = 1;

    struct StageOption { int wanted = 1, stages = 1, group = 0; double predicted = 0.0; };   // wanted: what planStages is asked for (1: the plain program)
    // Everything a lowering produces - the lowered streams, the tier that runs them, the loaded code object of a translated
    // program and the device copy of its tables.  It is a pure function of a KEY (codeKey(): the program, the values folded
    // into the code, which registers have rows, the number of stages and the block-length class), so finished ones are
    // kept: a caller that comes back to a shape it has used before - a block length, a set of moving controls - gets a
    // pointer swap, not a translation and a module load.
    struct Code {
        std::string key;               // empty: nothing built
        Lowered low;
        bool useAsm = false;           // runs on the hand-written gfx950 kernel
        AsmVariant variant = ASM_LDS;
        std::string asmWhyNot;
        // translated program (fx_xlate.hpp): a code object of its own
        bool useXlate = false;
        hipModule_t module = nullptr;
        hipFunction_t fn = nullptr;
        uint64_t steady = 0, last = 0;  // {fast, exact} stream offsets as the kernel takes them
        uint32_t codeBytes = 0, initOff = 0, ldsBytes = 0;
        uint64_t codeHash = 0;          // imageHash() of the code object
        int stages = 1;                 // wavefronts per workgroup of the translated program (fx_xlate.hpp StageInfo)
        int blockClass = -1;            // the class of block lengths the staged code was generated for
        bool classMatters = false;      // the program can be cut: code for another class of block lengths would differ
        std::vector<StageDescriptor> stageDesc;
        std::vector<std::vector<int>> stageStoreRows;
        std::string stagesWhyNot;
        std::vector<StageOption> stageOptions;   // what rankStages made of the program (cheapest first)
        int stagePick = 1;                       // ... and the one this code was built for
        int inlined = 0, called = 0, unsaturated = 0, valu = 0, valuSlow = 0, valuClocks = 0, vgprConstants = 0;
        std::vector<uint8_t> wildRow;
        std::string xlateWhyNot;
        bool prioritySlices = false;   // generated code: the wavefronts of a SIMD take turns at the top priority (fx_xlate.hpp)
        bool deferred = false;          // the interpreter runs this one because controls were moving when it was built
        uint32_t* dStream = nullptr;    // records / row table / stage descriptors on the device
        size_t streamCap = 0;
        uint64_t lastUse = 0;
    };
    Code c_;                                     // the code in force
    std::vector<std::unique_ptr<Code>> cache_;   // finished ones that are not (at most kCodeCache)
    static constexpr size_t kCodeCache = 8;
    uint64_t useClock_ = 0;
    int loadGen_ = 0;
    int cacheHits_ = 0;
    std::string codeKey(int blockClass, bool defer) const;
    std::string codeKeyFor(const std::vector<uint8_t>& forced, int blockClass, bool defer, int pick) const;
    struct Builder;                              // the thread that generates code off the caller's thread (fx_batch.cpp)
    std::unique_ptr<Builder> builder_;
    struct BuildInputs;
    bool builderWanted() const;
    void requestBuild(BuildInputs&& in);
    void collectBuilt();                         // what the builder has finished -> cache_
    bool buildPending(const std::string& key);
    bool buildFailed(const std::string& key);
    bool waitBuild(const std::string& key);
    void drainBuilder(bool stop);
    void prebuildControlVariant();
    int keyClass() const;
    void noteBlockLength(int nSamples);
    bool deferWanted() const;
    bool cachedCode(const std::string& key) const;
    void stashCode();                            // c_ -> cache_ (evicting the least recently used), c_ = empty
    bool adoptCode(const std::string& key);      // cache_ -> c_
    void releaseCode(Code& c);                   // unload / free what a Code holds on the device
    void clearCodeCache();
    struct BuildInputs {                         // what a build reads of the batch's changing state (a snapshot: builds also run on the builder thread)
        std::string key;
        int blockClass = -1;
        bool defer = false;
        std::vector<float> hostValue;
        std::vector<uint8_t> forced;             // laneForced()
        std::vector<int> trackRegs;
        int stagePick = 0;                       // stages to ask the planner for; 0: the cheapest by its costs (rankStages)
        // the batch's device state as it was when the build was asked for: a build on the builder thread must not change any of
        // it and reads only this copy (the caller's thread may be growing the state or the delay lines meanwhile)
        int instPerLane = 1, iSlotsAlloc = 0, xSlotsAlloc = 0, stateRows = 0;
        bool stagingOff = false;
    };
    BuildInputs buildInputs(const std::string& key, int blockClass, bool defer) const;
    int buildCodeInto(Code& c, const BuildInputs& in, bool offline, std::string* err);
    int wantedClass_ = -1;                       // block-length class the code should be for (sticky: see processDevice)
    bool readByProgram(int reg) const;
    bool declaredControl(int reg) const;
    void markControls();
    std::vector<uint8_t> intrinsicLane_, readByProgram_;   // per register, as of the last load
    bool controlMode_ = false;                   // the declared controls have rows (the host has moved one)
    // Control mode puts EVERY declared control in a row at the first touch of one (one change of code for the panel, built ahead:
    // no stall).  A row costs what the value folded into the code saves - config5 with `damp` in a row: 100 INTERPs that convert X
    // and form 1 - X every sample and keep their saturation, +33 % per block - so the controls that actually move ("hot": written
    // since the load) keep their rows and the others go back into the code: the LEAN variant, built on the builder thread while
    // the full one runs, adopted at a block boundary (a pointer swap), dropped for the full one (cached, never evicted) the moment
    // a folded control moves.  Same words either way (tests: the knob / control-variant parity tests, tests/hipstub controls scenario).
    // A control is hot from a write until it has been left alone for kCoolSamples sample periods (a slider rests most of the
    // time; a preset recall writes the whole panel once): then its value is folded back in as well.
    std::vector<uint8_t> declared_;              // per register: a declared control (as of the last load)
    std::vector<uint8_t> hotControl_;            // ... that the host has written lately
    std::vector<int64_t> lastControlWrite_;      // sampleClock_ of that write
    int64_t sampleClock_ = 0, lastCoolCheck_ = 0;   // sample periods processed by this handle
    static constexpr int64_t kCoolSamples = 8192;   // 171 ms at 48 kHz: the first rest after which a control is folded in again
    static constexpr int64_t kCoolSamplesMost = int64_t(1) << 24;   // ... doubled every time it moved again after cooling, up to 5.8 minutes
    std::vector<int64_t> coolAfter_;             // per control: the rest it needs now
    std::vector<uint8_t> cooledOnce_;            // ... it has cooled down before
    bool leanActive_ = false;                    // a lean variant is the code wanted now: laneForced() == forcedWithout(leanFolded_)
    std::vector<uint8_t> leanFolded_;            // the controls it has folded in
    bool leanPending_ = false;                   // a lean variant has been asked of the builder thread: leanWant_ folded, leanKey_
    std::vector<uint8_t> leanWant_;
    std::string leanKey_;
    bool leanStale_ = false;                     // the set of hot controls may differ from the rows of the code in force / on order
    bool coldControl(int reg) const;
    void coldSetChanged();
    void controlWritten(int reg);
    void leanStep();                             // head of a block: cool controls down, ask for / adopt the lean variant
    size_t lruVictim() const;

    // ---- how many stages (rankStages: the planner's costs; noteLaunchTime: options the model cannot tell apart are measured)
    bool stagingPossible() const { return stagingPossibleGiven(stagingOff_); }
    bool stagingPossibleGiven(bool stagingOff) const;
    bool stagingOff_ = false;                    // a staged launch failed to start on this device: the plain program from then on
    std::vector<StageOption> rankStages(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords, const XlateProgram& xprog,
                                        int nRows, int blockClass, int wavesPerSimdCap, bool stagingOff) const;
    static constexpr double kTuneBand = 1.6;     // options predicted within this factor of the cheapest are tried
    static constexpr int kTuneRuns = 3, kTuneMinSamples = 8;
    struct Tuner {
        bool init = false, done = false;
        std::vector<StageOption> options;        // on trial (the model's cheapest first)
        std::vector<float> bestNs;               // per option: the fastest launch seen, ns per sample
        std::vector<int> runs;
        int pick = 0;                            // StageOption::wanted in force; 0: nothing built yet (the build picks the model's cheapest)
        int trials = 0;
    };
    Tuner tune_[3];                              // per block-length class
    int pickFor(int blockClass) const { return (blockClass >= 0 && blockClass < 3) ? tune_[blockClass].pick : 0; }
    void adoptStageOptions();                    // after a build / an adoption: the options the code came with start the class's tuner
    void noteLaunchTime();                       // the previous launch's time -> the tuner; move to the next option / settle
    int lastLaunchPick_ = 0, lastLaunchSamples_ = 0, lastLaunchClass_ = -1;
    bool lastLaunchTimed_ = false;
    bool movableControl(int reg) const;
    bool piecewise_ = false;   // processDevice is being called for the pieces of one pipelined host block
    int xlateBuilds_ = 0;                    // translations on the caller's thread since the handle was created
    std::atomic<int> backgroundBuilds_{0};   // ... and on the builder thread
    // a pipeline fills and drains in 3 (K - 1) steps: short blocks get short steps and fewer stages (stageBlockClass); the code is
    // generated for the class of the block that triggered the translation and again when the blocks stay in another class
    int otherClassBlocks_ = 0;
    static int stageBlockClass(int nSamples) { return nSamples <= 48 ? 0 : (nSamples <= 256 ? 1 : 2); }
    // control changes re-lower; while they keep coming the interpreter tier is used (see ensureLowered)
    static constexpr int kHeatPerChange = 8;  // blocks a change keeps the translation deferred
    int controlHeat_ = 0;
    int pendingSamples_ = 0;  // block length of the call that triggered the lowering
    bool everLowered_ = false;
    double* dLut_ = nullptr;
    // control tracks (fx_xlate.hpp TrackEvent): registers the generated loop can re-load by itself, and what is armed
    struct PendingTrack { int period = 0, steps = 0; bool perInstance = false; std::vector<float> values; };
    std::vector<int> trackRegs_;           // register of slot t
    std::vector<PendingTrack> pendingTracks_;  // per slot; steps == 0: not armed
    uint32_t* dTracks_ = nullptr;
    size_t tracksCap_ = 0;                 // bytes
    bool tracksClear_ = false;             // the device header holds no armed schedule
    std::vector<uint32_t> trackStage_;     // host image of dTracks_ for the block being launched
    bool tracksArmed() const { for (const PendingTrack& t : pendingTracks_) if (t.steps > 0) return true; return false; }
    uint32_t* dScratch_ = nullptr;  // small device scratch: fill lists, reductions
    float* dIn_ = nullptr;
    float* dOut_ = nullptr;
    size_t ioCap_ = 0;
    // small blocks (the reference's one process() per sample): pinned host buffers the kernel reads and writes directly
    float* hPinIn_ = nullptr;
    float* hPinOut_ = nullptr;
    bool pinTried_ = false, untimed_ = false;
    // large host blocks: pieces of the block are copied in, processed and copied out concurrently
    static constexpr int kHostPieces = 8;
    hipStream_t copyIn_ = nullptr, copyOut_ = nullptr;
    hipEvent_t evIn_[kHostPieces] = {}, evDone_[kHostPieces] = {};
    int processHostPipelined(const float* in, float* out, int nSamples, int64_t pitch, int pieces);
    unsigned lastGrid_ = 0;
#ifdef FX_DIAGNOSTICS
  public:
    // diagnostics build (fx_knobs.hpp): one word per wavefront, written by generated code behind its last sample when
    // FX_XLATE_ENDSTAMP is set (fx_xlate.cpp) - the low word of the 100 MHz clock at which the wavefront finished
    int readEndStamps(uint32_t* out, int64_t nWords);
  private:
    uint32_t* dStamps_ = nullptr;
    size_t stampWords_ = 0;
#endif

    std::string lastError_;
};

}  // namespace fx
