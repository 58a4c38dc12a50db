// fx_shard.hpp — one batch of instances spread over several GPUs of a node (SURVEY.md section 8 b/e).
//
// Instances share nothing mutable, so the partition is contiguous instance ranges: shard k owns instances
// [first_k, first_k + count_k) on device devices[k]; the program, its LUTs and the control values are replicated.
// There is NO exchange step and no collective: every operation is a fan-out to the shards and a fan-in of their
// results.  Each shard has its own host thread (so that the devices' copies, launches and waits overlap) and its
// own HIP stream (inside fx::Batch).  Every call that reaches a shard's device - broadcast or per instance - is posted
// to that shard's thread (fan / runOn), so the CALLER's current HIP device is never changed by a multi-shard handle; reads of
// replicated host state (program, error list, controls: front()) stay on the caller's thread.  A "sharded" batch with a
// single shard (what fxb_create() makes) runs inline on the caller's thread, under a guard that restores the caller's
// current device afterwards.
#pragma once

#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "fx_batch.hpp"

namespace fx {

class Sharded {
public:
    // devices: HIP ordinals, one shard each (an ordinal may repeat: several shards on one GPU); -1 = the calling thread's
    // current device.  Throws std::runtime_error when a device is unusable or there are more shards than instances.
    Sharded(int64_t nInstances, int channels, const std::vector<int>& devices);
    ~Sharded();
    Sharded(const Sharded&) = delete;
    Sharded& operator=(const Sharded&) = delete;

    // The partition: contiguous ranges of whole wavefronts (64 instances), the remainder in the last shard.  Pure
    // arithmetic (no device): fxb_shard_plan() exposes it, the constructor uses it.  Throws when a shard would be empty.
    static std::vector<std::pair<int64_t, int64_t>> plan(int64_t nInstances, int nShards);

    int shards() const { return (int)shards_.size(); }
    int64_t instances() const { return n_; }
    int64_t firstOf(int k) const { return shards_[(size_t)k]->first; }
    int64_t countOf(int k) const { return shards_[(size_t)k]->count; }
    int deviceOf(int k) const { return shards_[(size_t)k]->batch->device(); }
    Batch& shard(int k) { return *shards_[(size_t)k]->batch; }
    Batch& front() { return *shards_.front()->batch; }  // replicated state (program, errors, controls) reads from here

    bool loadFile(const std::string& path);
    bool loadText(const std::string& text);
    int setRegister(const std::string& key, float v);
    int setRegisterAt(const std::string& key, int64_t inst, float v);
    float getRegisterAt(const std::string& key, int64_t inst);
    int setRegisterArray(const std::string& key, const float* values);
    int getRegisterArray(const std::string& key, float* values);
    int seedNoiseAt(int64_t inst, int32_t x1, int32_t x2);
    int setRegisterTrack(const std::string& key, const float* values, int nSteps, int period, bool perInstance);
    void setChannels(int c);
    int setOption(unsigned option, bool on);

    // host buffers [sample][channel][all instances]: every shard copies its columns in, runs, copies them out
    int processHost(const float* in, float* out, int nSamples);
    // device-resident buffers, one pair per shard: dIn[k] / dOut[k] are [sample][channel][count_k] on shard k's device;
    // asynchronous (pair with sync())
    int processDeviceShards(const float* const* dIn, float* const* dOut, int nSamples);
    // single shard only: the caller's stream
    int processDevice(const float* dIn, float* dOut, int nSamples, hipStream_t stream);
    int sync();
    int prepare(int nSamples, bool wait);

    // state snapshot of the whole batch, laid out by global instance (fx_batch.hpp SnapshotHeader): an image saved from one
    // partition loads into any other with the same program and instance count
    int64_t stateBytes();
    int saveState(void* buf, int64_t cap);
    int loadState(const void* buf, int64_t bytes);
    int getTramAt(int which, int64_t inst, float* out, int nSlots);
    int getCursorsAt(int64_t inst, int32_t out4[4]);

    int64_t instructionCounter();
    int64_t instructionCounterAt(int64_t inst);
    uint32_t oodFlags();
    float lastKernelMs();  // slowest shard
    float lastKernelMsOf(int k);
    int64_t info(int what);
    const std::string& lastError();

private:
    struct Worker {
        std::unique_ptr<Batch> batch;
        int64_t first = 0, count = 0;
        // a one-slot mailbox: the owner posts a task, the thread runs it, the owner waits for `done`
        std::thread thread;
        std::mutex mu;
        std::condition_variable cv;
        std::function<int()> task;
        bool pending = false, done = false, quit = false;
        int result = 0;
    };
    int shardOf(int64_t inst) const;
    // run f(k, batch) on every shard's thread, wait for all; returns the first non-zero result (shard order)
    int fan(const std::function<int(int, Batch&)>& f);
    // run f(batch) on shard k's thread and wait (a single shard: inline, caller's device restored)
    int runOn(int k, const std::function<int(Batch&)>& f);
    void stopThreads();
    static void loop(Worker* w);

    int64_t n_ = 0;
    std::vector<std::unique_ptr<Worker>> shards_;
    std::string lastError_;
    // One call at a time per handle.  A handle is not meant for concurrent use (the reference's objects are not either), but a host
    // with a UI thread that reads a register while its audio thread processes a block must get serialised calls, not corrupted
    // ones: every public call that touches the shards or lastError_ holds this for its whole duration (recursive: fan / runOn
    // take it again for their posts to the one-slot mailboxes).  Found wanting by ThreadSanitizer: lastError_.clear() used to run
    // in front of the lock (tests/hipstub/host_threads.cpp, scenario "shards").
    std::recursive_mutex api_;
    using Serial = std::lock_guard<std::recursive_mutex>;
};

}  // namespace fx
