// fx_shard.cpp — fan-out / fan-in over the shards of a multi-GPU batch (see fx_shard.hpp).
#include "fx_shard.hpp"

#include <cstring>

#include <algorithm>
#include <stdexcept>

#include "../../include/fx8010_amd.h"

namespace fx {

namespace {
// restores the calling thread's current HIP device: fx::Batch selects its own device in every call
struct DeviceGuard {
    int prev = -1;
    bool armed = false;
    DeviceGuard() { armed = hipGetDevice(&prev) == hipSuccess; }
    ~DeviceGuard() { if (armed) (void)hipSetDevice(prev); }
};
}  // namespace

std::vector<std::pair<int64_t, int64_t>> Sharded::plan(int64_t nInstances, int nShards) {
    if (nShards < 1) throw std::runtime_error("no device given");
    if (nInstances < (int64_t)nShards) throw std::runtime_error("fewer instances than shards");
    const int64_t k = nShards;
    // contiguous ranges, whole wavefronts (64 instances) per shard: the first (waves mod k) shards get one wavefront more,
    // the last shard ends at nInstances (a ragged last wavefront)
    const int64_t waves = (nInstances + 63) / 64;
    std::vector<std::pair<int64_t, int64_t>> out;
    int64_t first = 0;
    for (int64_t i = 0; i < k; ++i) {
        const int64_t wavesHere = waves / k + (i < waves % k ? 1 : 0);
        const int64_t count = i + 1 == k ? nInstances - first : std::min<int64_t>(wavesHere * 64, nInstances - first);
        if (count < 1) throw std::runtime_error("a shard would be empty: use fewer devices for this few instances");
        out.emplace_back(first, count);
        first += count;
    }
    return out;
}

Sharded::Sharded(int64_t nInstances, int channels, const std::vector<int>& devices) : n_(nInstances) {
    const auto ranges = plan(nInstances, (int)devices.size());
    DeviceGuard guard;  // constructing a Batch selects its device on this (the caller's) thread
    for (size_t i = 0; i < devices.size(); ++i) {
        auto w = std::make_unique<Worker>();
        w->first = ranges[i].first;
        w->count = ranges[i].second;
        w->batch = std::make_unique<Batch>(w->count, channels, devices[i]);
        shards_.push_back(std::move(w));
    }
    if (shards_.size() > 1) {
        try {
            for (auto& w : shards_) w->thread = std::thread(loop, w.get());
        } catch (...) {
            stopThreads();  // a joinable std::thread must not be destroyed: stop the workers that did start, then report
            throw;
        }
    }
}

void Sharded::stopThreads() {
    for (auto& w : shards_) {
        if (!w->thread.joinable()) continue;
        {
            std::lock_guard<std::mutex> lock(w->mu);
            w->quit = true;
        }
        w->cv.notify_all();
        w->thread.join();
    }
}

Sharded::~Sharded() {
    stopThreads();
    // the batches are destroyed on this (the caller's) thread, and a Batch selects its own device to free what it holds: the
    // caller's current device is restored afterwards (found by the stand-in's multi-device scenario, tests/hipstub)
    DeviceGuard guard;
    shards_.clear();
}

void Sharded::loop(Worker* w) {
    std::unique_lock<std::mutex> lock(w->mu);
    for (;;) {
        w->cv.wait(lock, [&] { return w->pending || w->quit; });
        if (w->quit) return;
        std::function<int()> task = std::move(w->task);
        w->pending = false;
        lock.unlock();
        int r;
        try {
            r = task();
        } catch (const std::exception& e) {
            w->batch->noteError(e.what());
            r = FX_E_MEMORY;
        }
        lock.lock();
        w->result = r;
        w->done = true;
        w->cv.notify_all();
    }
}

int Sharded::fan(const std::function<int(int, Batch&)>& f) {
    if (shards_.size() == 1) {
        DeviceGuard guard;
        return f(0, *shards_[0]->batch);
    }
    // one post at a time per handle: every worker has ONE mailbox slot, and a second host thread (a UI thread reading a register
    // while the audio thread processes a block) must not overwrite a task that has not been picked up yet - its caller would
    // wait for `done` forever.  (A handle is still not meant for concurrent use: calls are serialised, not made independent.)
    Serial post(api_);
    for (size_t k = 0; k < shards_.size(); ++k) {
        Worker* w = shards_[k].get();
        std::lock_guard<std::mutex> lock(w->mu);
        w->task = [&f, k, w] { return f((int)k, *w->batch); };
        w->pending = true;
        w->done = false;
        w->cv.notify_all();
    }
    int first = 0;
    for (auto& w : shards_) {
        std::unique_lock<std::mutex> lock(w->mu);
        w->cv.wait(lock, [&] { return w->done; });
        if (first == 0 && w->result != 0) {
            first = w->result;
            lastError_ = w->batch->lastError();
        }
    }
    return first;
}

int Sharded::runOn(int k, const std::function<int(Batch&)>& f) {
    Worker* w = shards_[(size_t)k].get();
    if (shards_.size() == 1) {
        DeviceGuard guard;
        return f(*w->batch);
    }
    Serial post(api_);
    {
        std::lock_guard<std::mutex> lock(w->mu);
        w->task = [&f, w] { return f(*w->batch); };
        w->pending = true;
        w->done = false;
        w->cv.notify_all();
    }
    std::unique_lock<std::mutex> lock(w->mu);
    w->cv.wait(lock, [&] { return w->done; });
    if (w->result != 0) lastError_ = w->batch->lastError();
    return w->result;
}

int Sharded::shardOf(int64_t inst) const {
    for (size_t k = 0; k < shards_.size(); ++k)
        if (inst >= shards_[k]->first && inst < shards_[k]->first + shards_[k]->count) return (int)k;
    return -1;
}

const std::string& Sharded::lastError() { return lastError_.empty() ? shards_.front()->batch->lastError() : lastError_; }

bool Sharded::loadFile(const std::string& path) {
    Serial serial(api_);
    lastError_.clear();
    return fan([&](int, Batch& b) { return b.loadFile(path) ? 0 : 1; }) == 0;
}
bool Sharded::loadText(const std::string& text) {
    Serial serial(api_);
    lastError_.clear();
    return fan([&](int, Batch& b) { return b.loadText(text) ? 0 : 1; }) == 0;
}
int Sharded::setRegister(const std::string& key, float v) {
    Serial serial(api_);
    lastError_.clear();
    return fan([&](int, Batch& b) { return b.setRegister(key, v); });
}
void Sharded::setChannels(int c) {
    Serial serial(api_);
    fan([&](int, Batch& b) { b.setChannels(c); return 0; });
}
int Sharded::setOption(unsigned option, bool on) {
    Serial serial(api_);
    lastError_.clear();
    return fan([&](int, Batch& b) { return b.setOption(option, on) ? -3 : 0; });
}
int Sharded::setRegisterAt(const std::string& key, int64_t inst, float v) {
    Serial serial(api_);
    lastError_.clear();
    const int k = shardOf(inst);
    if (k < 0) { lastError_ = "instance out of range"; return front().program().findRegister(key) < 0 ? 1 : FX_E_ARG; }
    return runOn(k, [&](Batch& b) { return b.setRegisterAt(key, inst - shards_[(size_t)k]->first, v); });
}
float Sharded::getRegisterAt(const std::string& key, int64_t inst) {
    Serial serial(api_);
    lastError_.clear();
    const int k = shardOf(inst);
    float out = 1.0f;
    runOn(k < 0 ? 0 : k, [&](Batch& b) { out = b.getRegisterAt(key, k < 0 ? -1 : inst - shards_[(size_t)k]->first); return 0; });
    return out;
}
int Sharded::setRegisterArray(const std::string& key, const float* values) {
    Serial serial(api_);
    lastError_.clear();
    if (!values) { lastError_ = "null buffer"; return FX_E_ARG; }
    return fan([&](int k, Batch& b) { return b.setRegisterArray(key, values + shards_[(size_t)k]->first); });
}
int Sharded::getRegisterArray(const std::string& key, float* values) {
    Serial serial(api_);
    lastError_.clear();
    if (!values) { lastError_ = "null buffer"; return FX_E_ARG; }
    return fan([&](int k, Batch& b) { return b.getRegisterArray(key, values + shards_[(size_t)k]->first); });
}
int Sharded::seedNoiseAt(int64_t inst, int32_t x1, int32_t x2) {
    Serial serial(api_);
    lastError_.clear();
    const int k = shardOf(inst);
    if (k < 0) { lastError_ = "instance out of range"; return FX_E_ARG; }
    return runOn(k, [&](Batch& b) { return b.seedNoiseAt(inst - shards_[(size_t)k]->first, x1, x2); });
}

int Sharded::setRegisterTrack(const std::string& key, const float* values, int nSteps, int period, bool perInstance) {
    Serial serial(api_);
    lastError_.clear();
    if (!values) { lastError_ = "null buffer"; return FX_E_ARG; }
    // per-instance schedules are [step][all instances]: a shard takes its columns
    return fan([&](int k, Batch& b) { return b.setRegisterTrack(key, perInstance ? values + shards_[(size_t)k]->first : values, nSteps, period, perInstance, n_); });
}

int Sharded::processHost(const float* in, float* out, int nSamples) {
    Serial serial(api_);
    lastError_.clear();
    if (shards_.size() == 1) return runOn(0, [&](Batch& b) { return b.processHost(in, out, nSamples); });
    if (nSamples > 0 && (!in || !out)) { lastError_ = "null buffer"; return FX_E_ARG; }
    return fan([&](int k, Batch& b) {
        const int64_t first = shards_[(size_t)k]->first;
        return b.processHost(in ? in + first : in, out ? out + first : out, nSamples, n_);
    });
}
int Sharded::processDeviceShards(const float* const* dIn, float* const* dOut, int nSamples) {
    Serial serial(api_);
    lastError_.clear();
    if (nSamples > 0 && (!dIn || !dOut)) { lastError_ = "null buffer table"; return FX_E_ARG; }
    return fan([&](int k, Batch& b) { return b.processDevice(dIn ? dIn[k] : nullptr, dOut ? dOut[k] : nullptr, nSamples, nullptr); });
}
int Sharded::processDevice(const float* dIn, float* dOut, int nSamples, hipStream_t stream) {
    Serial serial(api_);
    lastError_.clear();
    if (shards_.size() != 1) { lastError_ = "a batch of several shards takes one buffer pair per shard: fxb_process_block_dev_shards"; return FX_E_ARG; }
    return runOn(0, [&](Batch& b) { return b.processDevice(dIn, dOut, nSamples, stream); });
}
int Sharded::sync() {
    Serial serial(api_);
    lastError_.clear();
    return fan([](int, Batch& b) { return b.sync(); });
}

int64_t Sharded::instructionCounter() {
    Serial serial(api_);
    std::vector<int64_t> part(shards_.size(), 0);
    fan([&](int k, Batch& b) { part[(size_t)k] = b.instructionCounter(); return 0; });
    int64_t sum = 0;
    for (int64_t p : part) {
        if (p < 0) return -1;
        sum += p;
    }
    return sum;
}
int64_t Sharded::instructionCounterAt(int64_t inst) {
    Serial serial(api_);
    const int k = shardOf(inst);
    if (k < 0) return 0;
    int64_t out = 0;
    runOn(k, [&](Batch& b) { out = b.instructionCounterAt(inst - shards_[(size_t)k]->first); return 0; });
    return out;
}
uint32_t Sharded::oodFlags() {
    Serial serial(api_);
    std::vector<uint32_t> part(shards_.size(), 0);
    fan([&](int k, Batch& b) { part[(size_t)k] = b.oodFlags(); return 0; });
    uint32_t all = 0;
    for (uint32_t p : part) all |= p;
    return all;
}
float Sharded::lastKernelMs() {
    Serial serial(api_);
    std::vector<float> part(shards_.size(), -1.0f);
    fan([&](int k, Batch& b) { part[(size_t)k] = b.lastKernelMs(); return 0; });  // (a shard's events belong to its thread's device)
    float worst = -1.0f;
    for (float p : part) worst = std::max(worst, p);
    return worst;
}
int64_t Sharded::stateBytes() {
    Serial serial(api_);
    lastError_.clear();
    Batch::SnapshotHeader hdr;
    if (runOn(0, [&](Batch& b) { return b.snapshotShape(&hdr); }) != 0) return -1;
    hdr.n = n_;
    return Batch::snapshotBytes(hdr);
}
int Sharded::saveState(void* buf, int64_t cap) {
    Serial serial(api_);
    lastError_.clear();
    Batch::SnapshotHeader hdr;
    int rc = runOn(0, [&](Batch& b) { return b.snapshotShape(&hdr); });
    if (rc != 0) return rc;
    hdr.n = n_;
    if (!buf || cap < Batch::snapshotBytes(hdr)) { lastError_ = "save_state: buffer smaller than fxb_state_size()"; return FX_E_ARG; }
    std::memcpy(buf, &hdr, sizeof(hdr));
    return fan([&](int k, Batch& b) { return b.saveStateColumns(static_cast<uint8_t*>(buf), hdr, shards_[(size_t)k]->first); });
}
int Sharded::loadState(const void* buf, int64_t bytes) {
    Serial serial(api_);
    lastError_.clear();
    Batch::SnapshotHeader hdr;
    if (!buf || bytes < (int64_t)sizeof(hdr)) { lastError_ = "load_state: no image"; return FX_E_ARG; }
    std::memcpy(&hdr, buf, sizeof(hdr));
    // the header is the caller's (a file that may be damaged): version, every count and the size they imply are checked here,
    // before a shard computes a single address from them
    const Batch::SnapshotHeader ours;
    const int64_t need = Batch::snapshotBytes(hdr);
    if (hdr.magic != ours.magic || hdr.version != ours.version || hdr.n != n_ || need < 0 || bytes < need) {
        lastError_ = "load_state: not a state image of a batch of this many instances (wrong version, damaged header or truncated)";
        return FX_E_ARG;
    }
    return fan([&](int k, Batch& b) { return b.loadStateColumns(static_cast<const uint8_t*>(buf), hdr, shards_[(size_t)k]->first); });
}
int Sharded::getTramAt(int which, int64_t inst, float* out, int nSlots) {
    Serial serial(api_);
    lastError_.clear();
    const int k = shardOf(inst);
    if (k < 0) { lastError_ = "instance out of range"; return FX_E_ARG; }
    return runOn(k, [&](Batch& b) { return b.getTramAt(which, inst - shards_[(size_t)k]->first, out, nSlots); });
}
int Sharded::getCursorsAt(int64_t inst, int32_t out4[4]) {
    Serial serial(api_);
    lastError_.clear();
    const int k = shardOf(inst);
    if (k < 0) { lastError_ = "instance out of range"; return FX_E_ARG; }
    return runOn(k, [&](Batch& b) { return b.getCursorsAt(inst - shards_[(size_t)k]->first, out4); });
}
int Sharded::prepare(int nSamples, bool wait) {
    Serial serial(api_);
    lastError_.clear();
    return fan([&](int, Batch& b) { return b.prepare(nSamples, wait); });
}
float Sharded::lastKernelMsOf(int k) {
    Serial serial(api_);
    float ms = -1.0f;
    runOn(k, [&](Batch& b) { ms = b.lastKernelMs(); return 0; });
    return ms;
}
int64_t Sharded::info(int what) {
    Serial serial(api_);
    std::vector<int64_t> part(shards_.size(), 0);
    fan([&](int k, Batch& b) { part[(size_t)k] = b.info(what); return 0; });
    if (what == FXB_INFO_GRID) {
        int64_t sum = 0;
        for (int64_t p : part) sum += p;
        return sum;
    }
    return part[0];
}

}  // namespace fx
