// fx_batch.hpp — host engine behind the C ABI: N instances of one program on one GPU.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "fx_asm.hpp"
#include "fx_xlate.hpp"
#include "fx_decode.hpp"
#include "fx_kernel.hpp"
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

    int64_t instructionCounter();
    int64_t instructionCounterAt(int64_t inst);
    uint32_t oodFlags();
    float lastKernelMs();
    int64_t info(int what);

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
    int setOption(unsigned option, bool on) {
        if (option & ~kOptAll) return -3;
        prog_.options = on ? (prog_.options | option) : (prog_.options & ~option);
        lowDirty_ = true;
        return 0;
    }

private:
    int fail(int code, const std::string& what);
    int hipFail(hipError_t e, const char* where);
    int afterLoad(bool ok);
    int ensureLowered();          // (re)lower + upload the stream when dirty
    int ensureState();            // allocate / grow the state block for the current register count
    int ensureTram();
    int uploadTracks(int nSamples, hipStream_t s);   // translated tier: header + values -> dTracks_
    int processWithTrackFallback(const float* dIn, float* dOut, int nSamples, hipStream_t stream);  // other tiers: cut the block
    bool tracked(int reg) const;
    std::vector<uint8_t> laneForced() const;      // forcedLane_ plus the trackable registers
    int fillRows(const std::vector<uint32_t>& rows, const std::vector<uint32_t>& values);
    bool laneResident(int reg) const;
    bool intrinsicLane(int reg) const;
    int chooseInstPerLane() const;
    hipStream_t pick(hipStream_t s) const { return s ? s : stream_; }
    void waitLastLaunch();        // host waits for the most recent kernel (an event of ours, not the caller's stream handle)

    Program prog_;
    std::vector<float> hostValue_;      // current value of every register as the host knows it
    std::vector<uint8_t> forcedLane_;   // registers given per-instance values by setRegisterAt
    std::vector<uint8_t> intrinsicLane_; // registers per-instance because the program writes them (as of the last lowering)
    Lowered low_;
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
    bool useAsm_ = false;          // the current lowering runs on the hand-written gfx950 kernel
    AsmVariant asmVariant_ = ASM_LDS;
    std::string asmWhyNot_;
    // translated program (fx_xlate.hpp): a code object of its own per lowering
    bool useXlate_ = false;
    hipModule_t xlateModule_ = nullptr;
    hipFunction_t xlateFn_ = nullptr;
    uint64_t xlateSteady_ = 0, xlateLast_ = 0;  // {fast, exact} stream offsets as the kernel takes them
    uint32_t xlateCodeBytes_ = 0, xlateInitOff_ = 0, xlateLdsBytes_ = 0;
    int stagesWanted(int variant) const;
    bool movableControl(int reg) const;
    bool piecewise_ = false;   // processDevice is being called for the pieces of one pipelined host block
    int xlateBuilds_ = 0;      // translations since the handle was created
    int xlateStages_ = 1;                        // wavefronts per workgroup of the translated program (fx_xlate.hpp StageInfo)
    // a pipeline fills and drains in 3 (K - 1) steps: short blocks get short steps and fewer stages (stageBlockClass); the code is
    // generated for the class of the block that triggered the translation and again when the blocks stay in another class
    int stagedForClass_ = -1, otherClassBlocks_ = 0;
    static int stageBlockClass(int nSamples) { return nSamples <= 48 ? 0 : (nSamples <= 256 ? 1 : 2); }
    std::vector<StageDescriptor> xlateStageDesc_;
    std::vector<std::vector<int>> xlateStageStoreRows_;
    std::string stagesWhyNot_;
    int xlateInlined_ = 0, xlateCalled_ = 0, xlateUnsaturated_ = 0, xlateValu_ = 0, xlateValuSlow_ = 0, xlateValuClocks_ = 0, xlateVgprConstants_ = 0;
    std::vector<uint8_t> xlateWildRow_;
    std::string xlateWhyNot_;
    // control changes re-lower; while they keep coming the interpreter tier is used (see ensureLowered)
    static constexpr int kHeatPerChange = 8;  // blocks a change keeps the translation deferred
    int controlHeat_ = 0;
    int pendingSamples_ = 0;  // block length of the call that triggered the lowering
    bool xlateDeferred_ = false, everLowered_ = false;
    bool daneHipOnly_ = false;  // a DANE-model program that the translator could not take: HIP C++ kernel from now on
    double* dLut_ = nullptr;
    uint32_t* dStream_ = nullptr;
    size_t streamCap_ = 0;
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
    int processHostPipelined(const float* in, float* out, int nSamples, int64_t pitch);
    unsigned lastGrid_ = 0;

    std::string lastError_;
};

}  // namespace fx
