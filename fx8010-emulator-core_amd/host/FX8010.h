// FX8010.h — host-side mirror of the reference's public header, over the C ABI of libfx8010_amd.so.
//
// Same namespace, class name, member names, argument meaning and return conventions as the
// reference (include/FX8010.h:47-75), plus everything else that header hands to a translation unit
// that includes it (include/FX8010.h:12-25 the standard headers and `using namespace std`, :33-42 the
// macros E, PI, SAMPLERATE, AUDIOBLOCKSIZE, DEBUG, PRINT_REGISTERS, MAX_IDELAY_SIZE, MAX_XDELAY_SIZE,
// :50 the default constructor), so that a caller written against the reference - its console harness
// source/main.cpp, or the VST block loop it is meant for - compiles unchanged against this header and
// runs the instruction loop on an MI355X instead of the host CPU.  tests/test_dropin_harness.py compiles
// and links the reference's own source/main.cpp + source/helpers.cpp, where they lie, against it:
//     g++ -std=c++17 -DFX8010_REFERENCE_COMPAT -include fx8010-emulator-core_amd/host/FX8010.h -I include \
//         /root/reference/source/main.cpp /root/reference/source/helpers.cpp -L fx8010-emulator-core_amd -lfx8010_amd
// (with FX8010_REFERENCE_COMPAT this header defines the reference's include guard FX8010_H, so main.cpp's own
// #include "../include/FX8010.h" contributes nothing; without it only the classes below are declared).
//
//   Klangraum::FX8010         one emulated DSP, one process() call per sample period
//   Klangraum::FX8010Batch    N independent DSPs stepping one program on one GPU (the data-parallel path) or, built
//                             with a device list / mask, over several GPUs of one node (contiguous instance
//                             ranges, one host thread + stream per device, no collective)
//
// Header-only; link with -lfx8010_amd.  Differences that remain, by design:
//   * construction prints no banner (the reference prints ~7 lines, source/FX8010.cpp:18-24);
//   * when no HIP device is usable the constructor throws std::runtime_error - there is no CPU path;
//   * getInstructionCounter() is computed in 64 bits and truncated to int like the reference's field;
//   * the default constructor, declared but never defined by the reference (include/FX8010.h:50), makes a
//     one-channel DSP.
#ifndef FX8010_AMD_HOST_FX8010_H
#define FX8010_AMD_HOST_FX8010_H

#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "fx8010_amd.h"

// ---- opt-in: everything ELSE the reference header hands its includers (define FX8010_REFERENCE_COMPAT before including, or
// pass -DFX8010_REFERENCE_COMPAT).  The reference's own callers need it - source/main.cpp relies on `using namespace std`, PI,
// DEBUG, AUDIOBLOCKSIZE and on <iostream> / <chrono> / <math.h> coming with the header - but a new includer (a plug-in host)
// should not get a namespace dump and macros called E and DEBUG, so the class below is available without it.
#ifdef FX8010_REFERENCE_COMPAT
// the reference header's own guard: whoever includes this file has "included FX8010.h" (a later #include of the reference's
// include/FX8010.h contributes nothing)
#ifndef FX8010_H
#define FX8010_H
#endif

// what include/FX8010.h:12-24 pulls in for its includers
#include <stdio.h>
#include <iostream>
#include <chrono>
#include <math.h>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <regex>
#include <map>
#include <array>
#include <unordered_map>

using namespace std;  // include/FX8010.h:25 - part of what the reference header exports (main.cpp relies on it)

// the reference's compile-time settings (include/FX8010.h:33-42)
#ifndef E
#define E 2.71828182845
#endif
#ifndef PI
#define PI 3.14159265359
#endif
#ifndef SAMPLERATE
#define SAMPLERATE 48000
#endif
#ifndef AUDIOBLOCKSIZE
#define AUDIOBLOCKSIZE 32
#endif
#ifndef DEBUG
#define DEBUG 0
#endif
#ifndef PRINT_REGISTERS
#define PRINT_REGISTERS 0
#endif
#ifndef MAX_IDELAY_SIZE
#define MAX_IDELAY_SIZE 8192
#endif
#ifndef MAX_XDELAY_SIZE
#define MAX_XDELAY_SIZE 1048576
#endif
#endif  // FX8010_REFERENCE_COMPAT

namespace Klangraum {

// the reference's settings as constants (include/FX8010.h:35-36, :41-42), whether or not the macros are exported
constexpr int kSampleRate = 48000, kAudioBlockSize = 32, kMaxIDelaySize = 8192, kMaxXDelaySize = 1048576;

class FX8010 {
public:
    // reference: FX8010(), include/FX8010.h:50 (declared only there)
    FX8010() : FX8010(1) {}
    // reference: FX8010(int numChannels), include/FX8010.h:51
    FX8010(int numChannels) : channels_(numChannels), h_(fx_create(numChannels)) {
        if (!h_) throw std::runtime_error(std::string("FX8010: ") + fx_last_create_error());
    }
    ~FX8010() { fx_destroy(h_); }
    FX8010(const FX8010&) = delete;
    FX8010& operator=(const FX8010&) = delete;

    // reference: initialize() builds the LOG/EXP tables; here they are built when the library loads
    void initialize() {}

    // reference: std::vector<float> process(const std::vector<float>&), include/FX8010.h:57
    std::vector<float> process(const std::vector<float>& inputSamples) {
        std::vector<float> out((size_t)channels_, 0.0f);  // the reference's outputBuffer keeps its constructed size (FX8010.cpp:122)
        if (fx_process(h_, inputSamples.data(), out.data()) < 0) throw std::runtime_error(std::string("FX8010::process: ") + fx_last_error(h_));
        return out;
    }
    // extension: nSamples consecutive sample periods in one launch; in/out are [nSamples][channels]
    std::vector<float> processBlock(const std::vector<float>& in, int nSamples) {
        std::vector<float> out((size_t)nSamples * (size_t)channels_, 0.0f);
        if (fx_process_block(h_, in.data(), out.data(), nSamples) < 0) throw std::runtime_error(std::string("FX8010::processBlock: ") + fx_last_error(h_));
        return out;
    }

    int getInstructionCounter() { return (int)fx_instruction_counter(h_); }
    bool loadFile(const std::string& path) { return fx_load_file(h_, path.c_str()) == 1; }
    bool load(const std::string& path) { return loadFile(path); }  // alias (the reference's member is loadFile, include/FX8010.h:62)

    struct MyError {
        std::string errorDescription = "";
        int errorRow = 1;
    };
    std::vector<MyError> getErrorList() {
        std::vector<MyError> v;
        for (int i = 0, n = fx_error_count(h_); i < n; ++i) v.push_back({fx_error_desc(h_, i), fx_error_row(h_, i)});
        return v;
    }
    int setRegisterValue(const std::string& key, float value) { return fx_set_register(h_, key.c_str(), value); }
    float getRegisterValue(const std::string& key) { return fx_get_register(h_, key.c_str()); }
    std::vector<std::string> getControlRegisters() {
        std::vector<std::string> v;
        for (int i = 0, n = fx_control_count(h_); i < n; ++i) v.emplace_back(fx_control_at(h_, i));
        return v;
    }
    std::unordered_map<std::string, std::string> getMetaData() {
        std::unordered_map<std::string, std::string> m;
        static const char* const keys[] = {"name", "copyright", "created", "engine", "comment", "guid"};
        char buf[1024];
        for (const char* k : keys)
            if (fx_meta_get(h_, k, buf, (int)sizeof buf)) m[k] = buf;
        return m;
    }
    inline void setChannels(int numChannels_) { fx_set_channels(h_, numChannels_); }
    inline int getChannels() { return fx_get_channels(h_); }
    bool getReadyStatus() { return fx_ready(h_) != 0; }

private:
    int channels_;
    fx_handle* h_;
};

// N instances of one program on one GPU - or, with a device list / mask, spread over several GPUs of the node
// (contiguous instance ranges, one host thread + stream per device inside the library, no exchange between them).
// PCM layout: buf[(sample * channels + channel) * N + instance].
class FX8010Batch {
public:
    FX8010Batch(int64_t nInstances, int numChannels, int device = -1) : n_(nInstances), ch_(numChannels), h_(fxb_create(nInstances, numChannels, device)) {
        if (!h_) throw std::runtime_error(std::string("FX8010Batch: ") + fx_last_create_error());
    }
    // one shard per entry of `devices` (HIP ordinals; an ordinal may repeat)
    FX8010Batch(int64_t nInstances, int numChannels, const std::vector<int>& devices)
        : n_(nInstances), ch_(numChannels), h_(fxb_create_on_devices(nInstances, numChannels, devices.data(), (int)devices.size())) {
        if (!h_) throw std::runtime_error(std::string("FX8010Batch: ") + fx_last_create_error());
    }
    // one shard per set bit of deviceMask (bit d = HIP ordinal d): SURVEY.md section 8b's fxb_create(nInstances, ch, deviceMask)
    static FX8010Batch* sharded(int64_t nInstances, int numChannels, uint64_t deviceMask) {
        std::vector<int> devices;
        for (int d = 0; d < 64; ++d)
            if (deviceMask & (1ull << d)) devices.push_back(d);
        return new FX8010Batch(nInstances, numChannels, devices);
    }
    int shardCount() { return fxb_shard_count(h_); }
    // device-resident buffers of a sharded batch: dIn[k] / dOut[k] live on shard k's device, [nSamples][channels][instances of shard k]
    void processDeviceShards(const float* const* dIn, float* const* dOut, int nSamples) {
        if (fxb_process_block_dev_shards(h_, dIn, dOut, nSamples) < 0) throw std::runtime_error(std::string("FX8010Batch::processDeviceShards: ") + fxb_last_error(h_));
    }
    ~FX8010Batch() { fxb_destroy(h_); }
    FX8010Batch(const FX8010Batch&) = delete;
    FX8010Batch& operator=(const FX8010Batch&) = delete;

    bool loadFile(const std::string& path) { return fxb_load_file(h_, path.c_str()) == 1; }
    bool load(const std::string& path) { return loadFile(path); }
    bool loadText(const std::string& text) { return fxb_load_text(h_, text.c_str()) == 1; }
    int setRegisterValue(const std::string& key, float value) { return fxb_set_register(h_, key.c_str(), value); }
    int setRegisterValue(const std::string& key, int64_t instance, float value) { return fxb_set_register_i(h_, key.c_str(), instance, value); }
    float getRegisterValue(const std::string& key, int64_t instance) { return fxb_get_register_i(h_, key.c_str(), instance); }
    // one value per instance (values.size() == instances): per-instance control automation between blocks
    int setRegisterValues(const std::string& key, const std::vector<float>& values) { return fxb_set_register_array(h_, key.c_str(), values.data()); }
    // a schedule for the next process call: at its sample s, s % period == 0, the register takes values[s / period]
    // (perInstance: values[(s / period) * instances + instance]); the slider of source/main.cpp:107-114 without cutting the block
    int setRegisterTrack(const std::string& key, const std::vector<float>& values, int period, bool perInstance = false) {
        const int steps = (int)(perInstance ? values.size() / (size_t)n_ : values.size());
        return fxb_set_register_track(h_, key.c_str(), values.data(), steps, period, perInstance ? 1 : 0);
    }
    // nSamples sample periods for every instance (host buffers, synchronous)
    void process(const float* in, float* out, int nSamples) {
        if (fxb_process_block(h_, in, out, nSamples) < 0) throw std::runtime_error(std::string("FX8010Batch::process: ") + fxb_last_error(h_));
    }
    // device-resident buffers, asynchronous on `stream` (hipStream_t)
    void processDevice(const float* dIn, float* dOut, int nSamples, void* stream = nullptr) {
        if (fxb_process_block_dev(h_, dIn, dOut, nSamples, stream) < 0) throw std::runtime_error(std::string("FX8010Batch::processDevice: ") + fxb_last_error(h_));
    }
    void sync() { fxb_sync(h_); }
    // generate the code for blocks of nSamples samples now, not in the first process call (callers with a deadline per block)
    void prepare(int nSamples, bool wait = true) {
        if (fxb_prepare(h_, nSamples, wait ? 1 : 0) < 0) throw std::runtime_error(std::string("FX8010Batch::prepare: ") + fxb_last_error(h_));
    }
    // the whole batch's DSP state (the reference: plain members, include/FX8010.h:162-217, 288-291) as one image, laid out by global
    // instance: it loads into any batch of the same size with the same program, whatever its partition into shards
    std::vector<unsigned char> saveState() {
        const int64_t bytes = fxb_state_size(h_);
        if (bytes < 0) throw std::runtime_error(std::string("FX8010Batch::saveState: ") + fxb_last_error(h_));
        std::vector<unsigned char> image((size_t)bytes);
        if (fxb_save_state(h_, image.data(), bytes) < 0) throw std::runtime_error(std::string("FX8010Batch::saveState: ") + fxb_last_error(h_));
        return image;
    }
    void loadState(const std::vector<unsigned char>& image) {
        if (fxb_load_state(h_, image.data(), (int64_t)image.size()) < 0) throw std::runtime_error(std::string("FX8010Batch::loadState: ") + fxb_last_error(h_));
    }
    int64_t getInstructionCounter() { return fxb_instruction_counter(h_); }
    int64_t getInstructionCounter(int64_t instance) { return fxb_instruction_counter_i(h_, instance); }
    float lastKernelMs() { return fxb_last_kernel_ms(h_); }
    // which tier runs the program as it stands, and why not a faster one (text for the host's log)
    std::string tierNote() {
        char buf[512] = {0};
        fxb_tier_note(h_, buf, (int)sizeof buf);
        return buf;
    }
    int64_t instances() const { return n_; }
    int channels() const { return ch_; }
    fxb_handle* handle() { return h_; }

    // A PCM buffer [nSamples][channels][instances] in pinned host memory (fxb_host_alloc): process() on such buffers runs in place -
    // no staging copies - which is what a host with a deadline per block wants.  Owns its memory; movable, not copyable.
    class PcmBuffer {
    public:
        PcmBuffer() = default;
        PcmBuffer(int64_t instances, int channels, int nSamples)
            : floats_((size_t)instances * (size_t)channels * (size_t)nSamples), p_(static_cast<float*>(fxb_host_alloc((int64_t)floats_ * 4))) {
            if (!p_) throw std::runtime_error(std::string("FX8010Batch::PcmBuffer: ") + fx_last_create_error());
        }
        ~PcmBuffer() { fxb_host_free(p_); }
        PcmBuffer(PcmBuffer&& o) noexcept : floats_(o.floats_), p_(o.p_) { o.p_ = nullptr; o.floats_ = 0; }
        PcmBuffer& operator=(PcmBuffer&& o) noexcept { if (this != &o) { fxb_host_free(p_); p_ = o.p_; floats_ = o.floats_; o.p_ = nullptr; o.floats_ = 0; } return *this; }
        PcmBuffer(const PcmBuffer&) = delete;
        PcmBuffer& operator=(const PcmBuffer&) = delete;
        float* data() { return p_; }
        const float* data() const { return p_; }
        size_t size() const { return floats_; }
    private:
        size_t floats_ = 0;
        float* p_ = nullptr;
    };
    PcmBuffer pcmBuffer(int nSamples) const { return PcmBuffer(n_, ch_, nSamples); }

private:
    int64_t n_;
    int ch_;
    fxb_handle* h_;
};

}  // namespace Klangraum

#endif
