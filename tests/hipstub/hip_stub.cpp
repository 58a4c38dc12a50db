// hip_stub.cpp — a device-free stand-in for the HIP runtime calls the host side of libfx8010_amd.so makes (TEST INFRASTRUCTURE).
//
// The product's host engine (fx_batch.cpp, fx_shard.cpp, fx_asm.cpp: code cache, builder thread, launch-timing tuner, shard
// mailboxes) is ordinary multi-threaded C++ whose only contact with the GPU is the HIP runtime API.  This file implements
// that API - the ~40 functions the library calls, nothing more - on the host, so that the SAME host sources, untouched, can be
// linked into a library that runs without a GPU under ThreadSanitizer / AddressSanitizer (csrc/Makefile `tsan`, `stubasan`;
// tests/test_host_sanitizers.py).  It is the seam VERDICT r4 #4 asks for, put where the device boundary already is.
//
// Model:  "device memory" is host memory (hipMalloc = calloc, accounted against a capacity so that out-of-memory can be
// provoked);  a stream is a worker thread with a FIFO of operations (copies, memsets, event records, waits, kernels) - work
// queued on a stream really runs concurrently with the caller, which is what the sanitizer is there to watch;  an event is a
// counter of records with the completion time of the latest;  a module "loads" any image (a counter lets a test fail the N-th
// load);  a kernel takes a configurable time and TOUCHES what the real one would: it reads the PCM input, the row table, the
// control-track list and the stage descriptors and writes the PCM output (out = in) and the first state words - so freeing or
// rewriting a buffer a queued launch still uses shows up as a race.  The library's three own kernels that the host relies on
// for VALUES (fill rows, reduce a row, the handler-address probe) do their real work (fx_kernel_stub.cpp).
// Nothing here computes FX8010 results: parity is the GPU tests' business.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hip_stub.h"

namespace {

using Clock = std::chrono::steady_clock;

struct Stream;
struct Event {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t recorded = 0, completed = 0;   // records issued / records the stream has reached
    Clock::time_point at{};
};

struct Stream {
    int device = 0;            // the device that was current when the stream was created
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool busy = false, quit = false;
    std::thread worker;
    Stream() {
        worker = std::thread([this] {
            std::unique_lock<std::mutex> lock(mu);
            for (;;) {
                cv.wait(lock, [this] { return quit || !q.empty(); });
                if (q.empty() && quit) return;
                std::function<void()> op = std::move(q.front());
                q.pop_front();
                busy = true;
                lock.unlock();
                op();
                lock.lock();
                busy = false;
                cv.notify_all();
            }
        });
    }
    ~Stream() {
        {
            std::lock_guard<std::mutex> lock(mu);
            quit = true;
        }
        cv.notify_all();
        worker.join();
    }
    void push(std::function<void()> op) {
        {
            std::lock_guard<std::mutex> lock(mu);
            q.push_back(std::move(op));
        }
        cv.notify_all();
    }
    void drain() {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [this] { return q.empty() && !busy; });
    }
};

struct Module { std::string tag; int device = 0; };   // a module belongs to the device it was loaded on
struct Function { Module* module; std::string name; };

struct Global {
    std::mutex mu;
    std::map<void*, size_t> allocations;        // device + pinned host
    std::map<void*, size_t> pinned;             // ... of those, the pinned host ones (start -> bytes)
    std::map<void*, int> deviceOf;              // device allocations: the device that was current at hipMalloc
    long crossDevice = 0;                       // launches / accesses that mixed devices (each also fails the call)
    std::atomic<long> badPcm{0};                // launches whose PCM was not wholly device-addressable, or overlapped without being one buffer
    size_t used = 0, capacity = (size_t)4 << 30;
    int devices = 1;
    std::vector<std::unique_ptr<Stream>> streams;
    std::vector<std::unique_ptr<Event>> events;
    std::vector<std::unique_ptr<Module>> modules;
    std::vector<std::unique_ptr<Function>> functions;
    std::unique_ptr<Stream> null;               // the default stream (lazily)
    // fault injection
    long moduleLoads = 0, failLoadFrom = -1, failLoadCount = 0;
    long mallocs = 0, failMallocFrom = -1, failMallocCount = 0;
    long launches = 0, failLaunchFrom = -1, failLaunchCount = 0;
    int failLaunchCode = (int)hipErrorLaunchOutOfResources;
    std::atomic<int> kernelMicros{150};
    std::atomic<long> kernelsRun{0};
    Global() {
        if (const char* v = std::getenv("FXSTUB_DEVICES")) devices = std::max(1, std::atoi(v));
        if (const char* v = std::getenv("FXSTUB_CAPACITY_MB")) capacity = (size_t)std::max(1, std::atoi(v)) << 20;
        if (const char* v = std::getenv("FXSTUB_KERNEL_US")) kernelMicros = std::max(0, std::atoi(v));
    }
};
Global& G() {
    static Global* g = new Global;   // (leaked on purpose: worker threads may outlive static destruction)
    return *g;
}

thread_local int tlsDevice = 0;
thread_local hipError_t tlsLastError = hipSuccess;

hipError_t fail(hipError_t e) {
    tlsLastError = e;
    return e;
}

Stream* streamOf(hipStream_t s) {
    if (s) return reinterpret_cast<Stream*>(s);
    Global& g = G();
    std::lock_guard<std::mutex> lock(g.mu);
    if (!g.null) g.null.reset(new Stream);
    return g.null.get();
}

bool known(Stream* s) {
    Global& g = G();
    std::lock_guard<std::mutex> lock(g.mu);
    if (g.null.get() == s) return true;
    for (auto& p : g.streams)
        if (p.get() == s) return true;
    return false;
}

void copy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height) {
    for (size_t r = 0; r < height; ++r) std::memcpy(static_cast<char*>(dst) + r * dpitch, static_cast<const char*>(src) + r * spitch, width);
}

// the kernarg block of the library's assembly kernels (fx_asm.hpp AsmArgs): only the fields the stand-in kernel touches
struct AsmArgsView {
    const uint32_t* steady; const uint32_t* last; const uint32_t* rowTable; uint32_t* state; const float* in; float* out;
    float* itram; float* xtram; const double* lut; long long n; long long nPad; int nLoad, nStore; int nSamples, channels;
    int inOff[4]; int latchOff[4]; int iSlots, xSlots, iSize, xSize; int cursorRow, noiseRow; int oodRow, countLo, countHi, staticCount;
    int lutX1Off; int initOff; const uint32_t* tracks; const uint32_t* stages; int nStages; int tramDane;
};
static_assert(sizeof(AsmArgsView) == 0xd0, "AsmArgs layout (fx_asm.hpp)");

void runAsmKernel(const std::string& name, const std::vector<unsigned char>& kernarg) {
    Global& g = G();
    AsmArgsView a;
    if (kernarg.size() < sizeof(a)) return;
    std::memcpy(&a, kernarg.data(), sizeof(a));
    if (name.size() > 6 && name.compare(name.size() - 6, 6, "_probe") == 0) {
        // the handler-address probe (fx_asm.cpp asmHandlerTable): kAsmSets * kAsmSlots distinct non-zero "addresses"
        uint64_t* out = reinterpret_cast<uint64_t*>(a.out);
        for (int k = 0; k < 4 * 84; ++k) out[k] = 0x7f0000000000ull + (uint64_t)k * 64u;
        return;
    }
    const int us = g.kernelMicros.load();
    if (us > 0) std::this_thread::sleep_for(std::chrono::microseconds(us));
    volatile uint32_t sink = 0;
    const size_t count = (size_t)std::max(a.nSamples, 0) * (size_t)std::max(a.channels, 0) * (size_t)std::max<long long>(a.n, 0);
    if (a.rowTable)
        for (int k = 0; k < a.nLoad + a.nStore; ++k) sink = sink + a.rowTable[k];
    if (a.tracks) sink = sink + a.tracks[0] + a.tracks[1] + a.tracks[2] + a.tracks[3];
    if (a.stages && a.nStages > 1)
        for (int k = 0; k < a.nStages * 8; ++k) sink = sink + a.stages[k];
    if (a.state && a.nPad > 0) { a.state[0] = a.state[0]; a.state[(size_t)a.nPad - 1] = a.state[(size_t)a.nPad - 1]; }
    if (a.in && a.out && count) {
        // the PCM of a launch: wholly inside memory the device can address (a device allocation, pinned host memory - a real GPU
        // faults on anything else), and either one buffer or two that do not overlap (instances read and write sample by sample)
        const size_t bytes = count * sizeof(float);
        const char *pi = reinterpret_cast<const char*>(a.in), *po = reinterpret_cast<const char*>(a.out);
        bool ok = pi == po || pi + bytes <= po || po + bytes <= pi;
        {
            std::lock_guard<std::mutex> lock(g.mu);
            for (const char* p : {pi, po}) {
                bool inside = false;
                for (const auto& m : g.allocations) inside = inside || (p >= static_cast<char*>(m.first) && p + bytes <= static_cast<char*>(m.first) + m.second);
                for (const auto& m : g.pinned) inside = inside || (p >= static_cast<char*>(m.first) && p + bytes <= static_cast<char*>(m.first) + m.second);
                ok = ok && inside;
            }
        }
        if (!ok) {
            g.badPcm.fetch_add(1);
            if (std::getenv("FXSTUB_ABORT_ON_BAD_PCM")) {   // (drivers that do not read the counter: the API fuzzer)
                std::fprintf(stderr, "hip stub: a launch on PCM the device cannot address, or on overlapping input and output ranges\n");
                std::abort();
            }
        }
        else if (pi != po) std::memcpy(a.out, a.in, bytes);   // "out = in": data does flow through a launch
    }
    (void)sink;
    g.kernelsRun.fetch_add(1);
}

}  // namespace

// ---- test controls (hip_stub.h) ---------------------------------------------------------------------------------------------
extern "C" {
void fxstub_fail_module_loads(long from_nth, long count) { Global& g = G(); std::lock_guard<std::mutex> l(g.mu); g.failLoadFrom = from_nth < 0 ? -1 : g.moduleLoads + from_nth; g.failLoadCount = count; }
void fxstub_fail_mallocs(long from_nth, long count) { Global& g = G(); std::lock_guard<std::mutex> l(g.mu); g.failMallocFrom = from_nth < 0 ? -1 : g.mallocs + from_nth; g.failMallocCount = count; }
void fxstub_fail_launches(long from_nth, long count, int hip_error) { Global& g = G(); std::lock_guard<std::mutex> l(g.mu); g.failLaunchFrom = from_nth < 0 ? -1 : g.launches + from_nth; g.failLaunchCount = count; g.failLaunchCode = hip_error; }
void fxstub_set_capacity(unsigned long long bytes) { Global& g = G(); std::lock_guard<std::mutex> l(g.mu); g.capacity = (size_t)bytes; }
void fxstub_set_kernel_micros(int us) { G().kernelMicros = us < 0 ? 0 : us; }
unsigned long long fxstub_bytes_in_use(void) { Global& g = G(); std::lock_guard<std::mutex> l(g.mu); return g.used; }
long fxstub_live_allocations(void) { Global& g = G(); std::lock_guard<std::mutex> l(g.mu); return (long)g.allocations.size(); }
long fxstub_module_loads(void) { Global& g = G(); std::lock_guard<std::mutex> l(g.mu); return g.moduleLoads; }
long fxstub_live_modules(void) { Global& g = G(); std::lock_guard<std::mutex> l(g.mu); return (long)g.modules.size(); }
long fxstub_kernels_run(void) { return G().kernelsRun.load(); }
long fxstub_bad_pcm_launches(void) { return G().badPcm.load(); }
long fxstub_cross_device_errors(void) { Global& g = G(); std::lock_guard<std::mutex> l(g.mu); return g.crossDevice; }
}

// what fx_kernel_stub.cpp needs: run a host function in stream order
void fxstubEnqueue(hipStream_t stream, std::function<void()> op) { streamOf(stream)->push(std::move(op)); }

// ---- the HIP runtime subset -----------------------------------------------------------------------------------------------
extern "C" {

hipError_t hipGetDeviceCount(int* count) { *count = G().devices; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= G().devices) return fail(hipErrorInvalidDevice); tlsDevice = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = tlsDevice; return hipSuccess; }
hipError_t hipGetLastError(void) { const hipError_t e = tlsLastError; tlsLastError = hipSuccess; return e; }
const char* hipGetErrorString(hipError_t e) {
    switch (e) {
        case hipSuccess: return "no error (stub)";
        case hipErrorOutOfMemory: return "out of memory (stub)";
        case hipErrorInvalidValue: return "invalid argument (stub)";
        case hipErrorNotReady: return "device not ready (stub)";
        case hipErrorLaunchOutOfResources: return "too many resources requested for launch (stub)";
        case hipErrorSharedObjectInitFailed: return "shared object initialization failed (stub)";
        default: return "error (stub)";
    }
}

static hipError_t allocate(void** p, size_t bytes) {
    Global& g = G();
    std::lock_guard<std::mutex> lock(g.mu);
    const long nth = g.mallocs++;
    if (g.failMallocFrom >= 0 && nth >= g.failMallocFrom && nth < g.failMallocFrom + g.failMallocCount) { *p = nullptr; return fail(hipErrorOutOfMemory); }
    if (bytes > g.capacity || g.used > g.capacity - bytes) { *p = nullptr; return fail(hipErrorOutOfMemory); }
    void* m = std::calloc(1, bytes ? bytes : 1);
    if (!m) { *p = nullptr; return fail(hipErrorOutOfMemory); }
    g.allocations[m] = bytes;
    g.used += bytes;
    *p = m;
    return hipSuccess;
}
static hipError_t release(void* p) {
    if (!p) return hipSuccess;
    Global& g = G();
    {
        std::lock_guard<std::mutex> lock(g.mu);
        auto it = g.allocations.find(p);
        if (it == g.allocations.end()) return fail(hipErrorInvalidValue);
        g.used -= it->second;
        g.allocations.erase(it);
    }
    std::free(p);
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t bytes) {
    const hipError_t e = allocate(p, bytes);
    if (e == hipSuccess) { Global& g = G(); std::lock_guard<std::mutex> lock(g.mu); g.deviceOf[*p] = tlsDevice; }
    return e;
}
hipError_t hipFree(void* p) {
    { Global& g = G(); std::lock_guard<std::mutex> lock(g.mu); g.deviceOf.erase(p); }
    return release(p);
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
    const hipError_t e = allocate(p, bytes);
    if (e == hipSuccess) { Global& g = G(); std::lock_guard<std::mutex> lock(g.mu); g.pinned[*p] = bytes; }
    return e;
}
hipError_t hipHostFree(void* p) {
    { Global& g = G(); std::lock_guard<std::mutex> lock(g.mu); g.pinned.erase(p); }
    return release(p);
}
// caller memory pinned after the fact: device-visible like hipHostMalloc's, not the stand-in's to free
hipError_t hipHostRegister(void* p, size_t bytes, unsigned) {
    if (!p || !bytes) return fail(hipErrorInvalidValue);
    Global& g = G();
    std::lock_guard<std::mutex> lock(g.mu);
    if (g.pinned.count(p)) return fail(hipErrorHostMemoryAlreadyRegistered);
    g.pinned[p] = bytes;
    return hipSuccess;
}
hipError_t hipHostUnregister(void* p) {
    Global& g = G();
    std::lock_guard<std::mutex> lock(g.mu);
    if (!g.pinned.count(p) || g.allocations.count(p)) return fail(hipErrorHostMemoryNotRegistered);
    g.pinned.erase(p);
    return hipSuccess;
}
// pinned host memory (hipHostMalloc above) is device-visible at its own address; anything else the stand-in did not allocate is
// pageable: "invalid value", as the runtime says
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* attr, const void* ptr) {
    Global& g = G();
    std::lock_guard<std::mutex> lock(g.mu);
    std::memset(attr, 0, sizeof(*attr));
    for (const auto& a : g.pinned)
        if (ptr >= a.first && ptr < static_cast<char*>(a.first) + a.second) {
            attr->type = hipMemoryTypeHost;
            attr->devicePointer = const_cast<void*>(ptr);
            attr->hostPointer = const_cast<void*>(ptr);
            return hipSuccess;
        }
    for (const auto& a : g.allocations)
        if (ptr >= a.first && ptr < static_cast<char*>(a.first) + a.second) {
            attr->type = hipMemoryTypeDevice;
            attr->devicePointer = const_cast<void*>(ptr);
            return hipSuccess;
        }
    return fail(hipErrorInvalidValue);
}
hipError_t hipMemGetAddressRange(hipDeviceptr_t* base, size_t* size, hipDeviceptr_t ptr) {
    Global& g = G();
    std::lock_guard<std::mutex> lock(g.mu);
    for (const auto* m : {&g.pinned, &g.allocations})
        for (const auto& a : *m)
            if (static_cast<char*>(ptr) >= static_cast<char*>(a.first) && static_cast<char*>(ptr) < static_cast<char*>(a.first) + a.second) {
                *base = a.first;
                *size = a.second;
                return hipSuccess;
            }
    return fail(hipErrorInvalidValue);
}
hipError_t hipMemGetInfo(size_t* freeBytes, size_t* total) {
    Global& g = G();
    std::lock_guard<std::mutex> lock(g.mu);
    *total = g.capacity;
    *freeBytes = g.capacity - g.used;
    return hipSuccess;
}

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    Global& g = G();
    std::unique_ptr<Stream> st(new Stream);
    st->device = tlsDevice;
    *s = reinterpret_cast<hipStream_t>(st.get());
    std::lock_guard<std::mutex> lock(g.mu);
    g.streams.push_back(std::move(st));
    return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t* s) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamDestroy(hipStream_t s) {
    Global& g = G();
    std::unique_ptr<Stream> mine;
    {
        std::lock_guard<std::mutex> lock(g.mu);
        for (auto it = g.streams.begin(); it != g.streams.end(); ++it)
            if (it->get() == reinterpret_cast<Stream*>(s)) { mine = std::move(*it); g.streams.erase(it); break; }
    }
    if (!mine) return fail(hipErrorInvalidValue);
    mine->drain();   // (HIP lets queued work finish)
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
    Stream* st = streamOf(s);
    if (!known(st)) return fail(hipErrorInvalidValue);
    st->drain();
    return hipSuccess;
}

hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
    Global& g = G();
    std::unique_ptr<Event> ev(new Event);
    *e = reinterpret_cast<hipEvent_t>(ev.get());
    std::lock_guard<std::mutex> lock(g.mu);
    g.events.push_back(std::move(ev));
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) {
    Global& g = G();
    std::unique_ptr<Event> mine;
    {
        std::lock_guard<std::mutex> lock(g.mu);
        for (auto it = g.events.begin(); it != g.events.end(); ++it)
            if (it->get() == reinterpret_cast<Event*>(e)) { mine = std::move(*it); g.events.erase(it); break; }
    }
    if (!mine) return fail(hipErrorInvalidValue);
    // (a record still queued on a stream refers to the event: wait for it, as the runtime keeps the object alive)
    std::unique_lock<std::mutex> lock(mine->mu);
    mine->cv.wait(lock, [&] { return mine->completed == mine->recorded; });
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    Event* ev = reinterpret_cast<Event*>(e);
    if (!ev) return fail(hipErrorInvalidValue);
    {
        std::lock_guard<std::mutex> lock(ev->mu);
        ++ev->recorded;
    }
    streamOf(s)->push([ev] {
        std::lock_guard<std::mutex> lock(ev->mu);
        ++ev->completed;
        ev->at = Clock::now();
        ev->cv.notify_all();
    });
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    Event* ev = reinterpret_cast<Event*>(e);
    if (!ev) return fail(hipErrorInvalidValue);
    std::unique_lock<std::mutex> lock(ev->mu);
    const uint64_t want = ev->recorded;
    ev->cv.wait(lock, [&] { return ev->completed >= want; });
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) {
    Event* ev = reinterpret_cast<Event*>(e);
    if (!ev) return fail(hipErrorInvalidValue);
    std::lock_guard<std::mutex> lock(ev->mu);
    return ev->completed >= ev->recorded ? hipSuccess : fail(hipErrorNotReady);
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    Event* ea = reinterpret_cast<Event*>(a);
    Event* eb = reinterpret_cast<Event*>(b);
    if (!ea || !eb) return fail(hipErrorInvalidValue);
    Clock::time_point ta, tb;
    {
        std::lock_guard<std::mutex> lock(ea->mu);
        if (ea->completed == 0 || ea->completed < ea->recorded) return fail(ea->completed == 0 ? hipErrorInvalidValue : hipErrorNotReady);
        ta = ea->at;
    }
    {
        std::lock_guard<std::mutex> lock(eb->mu);
        if (eb->completed == 0 || eb->completed < eb->recorded) return fail(eb->completed == 0 ? hipErrorInvalidValue : hipErrorNotReady);
        tb = eb->at;
    }
    *ms = std::chrono::duration<float, std::milli>(tb - ta).count();
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    Event* ev = reinterpret_cast<Event*>(e);
    if (!ev) return fail(hipErrorInvalidValue);
    uint64_t want;
    {
        std::lock_guard<std::mutex> lock(ev->mu);
        want = ev->recorded;
    }
    streamOf(s)->push([ev, want] {
        std::unique_lock<std::mutex> lock(ev->mu);
        ev->cv.wait(lock, [&] { return ev->completed >= want; });
    });
    return hipSuccess;
}

// copies: the synchronous forms run on the caller's thread at once (every stream of the library is non-blocking: no implicit
// ordering with the null stream), the asynchronous ones in stream order
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
    if (bytes && (!dst || !src)) return fail(hipErrorInvalidValue);
    std::memcpy(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s) {
    if (bytes && (!dst || !src)) return fail(hipErrorInvalidValue);
    streamOf(s)->push([=] { std::memcpy(dst, src, bytes); });
    return hipSuccess;
}
hipError_t hipMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
    if (width > dpitch || width > spitch) return fail(hipErrorInvalidValue);
    copy2D(dst, dpitch, src, spitch, width, height);
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t s) {
    if (width > dpitch || width > spitch) return fail(hipErrorInvalidValue);
    streamOf(s)->push([=] { copy2D(dst, dpitch, src, spitch, width, height); });
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s) {
    streamOf(s)->push([=] { std::memset(dst, value, bytes); });
    return hipSuccess;
}

hipError_t hipModuleLoadData(hipModule_t* module, const void* image) {
    Global& g = G();
    if (!image) return fail(hipErrorInvalidValue);
    std::lock_guard<std::mutex> lock(g.mu);
    const long nth = g.moduleLoads++;
    if (g.failLoadFrom >= 0 && nth >= g.failLoadFrom && nth < g.failLoadFrom + g.failLoadCount) { *module = nullptr; return fail(hipErrorSharedObjectInitFailed); }
    // (an ELF image: the loader reads it - so does the stand-in, so that a freed or half-written image is seen)
    volatile unsigned char sink = 0;
    const unsigned char* bytes = static_cast<const unsigned char*>(image);
    for (int k = 0; k < 64; ++k) sink = sink + bytes[k];
    (void)sink;
    std::unique_ptr<Module> m(new Module);
    m->device = tlsDevice;
    *module = reinterpret_cast<hipModule_t>(m.get());
    g.modules.push_back(std::move(m));
    return hipSuccess;
}
hipError_t hipModuleUnload(hipModule_t module) {
    Global& g = G();
    std::lock_guard<std::mutex> lock(g.mu);
    for (auto it = g.modules.begin(); it != g.modules.end(); ++it)
        if (it->get() == reinterpret_cast<Module*>(module)) {
            for (auto f = g.functions.begin(); f != g.functions.end();)
                f = (*f)->module == it->get() ? g.functions.erase(f) : f + 1;
            g.modules.erase(it);
            return hipSuccess;
        }
    return fail(hipErrorInvalidValue);
}
hipError_t hipModuleGetFunction(hipFunction_t* fn, hipModule_t module, const char* name) {
    Global& g = G();
    std::lock_guard<std::mutex> lock(g.mu);
    for (auto& m : g.modules)
        if (m.get() == reinterpret_cast<Module*>(module)) {
            std::unique_ptr<Function> f(new Function{m.get(), name ? name : ""});
            *fn = reinterpret_cast<hipFunction_t>(f.get());
            g.functions.push_back(std::move(f));
            return hipSuccess;
        }
    return fail(hipErrorInvalidValue);
}
hipError_t hipModuleLaunchKernel(hipFunction_t fn, unsigned gx, unsigned, unsigned, unsigned bx, unsigned, unsigned, unsigned sharedBytes, hipStream_t s,
                                 void**, void** extra) {
    Global& g = G();
    std::string name;
    {
        std::lock_guard<std::mutex> lock(g.mu);
        const long nth = g.launches++;
        if (g.failLaunchFrom >= 0 && nth >= g.failLaunchFrom && nth < g.failLaunchFrom + g.failLaunchCount) return fail((hipError_t)g.failLaunchCode);
        bool found = false;
        int moduleDevice = 0;
        for (auto& f : g.functions)
            if (f.get() == reinterpret_cast<Function*>(fn)) { name = f->name; moduleDevice = f->module->device; found = true; }
        if (!found) return fail(hipErrorInvalidValue);   // (a function of an unloaded module)
        // a kernel runs on the device its module was loaded on: the stream must be that device's, and so must the calling
        // thread's current device (a module of device 0 launched for a shard of device 2 is the multi-GPU bug this catches)
        const int streamDevice = s ? reinterpret_cast<Stream*>(s)->device : tlsDevice;
        if (moduleDevice != streamDevice || moduleDevice != tlsDevice) { ++g.crossDevice; return fail(hipErrorInvalidResourceHandle); }
        // ... and every device pointer of its argument block must be memory of that device
        if (extra) {
            const void* ptr = nullptr;
            size_t size = 0;
            for (int k = 0; extra[k] != HIP_LAUNCH_PARAM_END && k < 8; k += 2) {
                if (extra[k] == HIP_LAUNCH_PARAM_BUFFER_POINTER) ptr = extra[k + 1];
                else if (extra[k] == HIP_LAUNCH_PARAM_BUFFER_SIZE) size = *static_cast<size_t*>(extra[k + 1]);
            }
            if (ptr && size >= sizeof(AsmArgsView)) {
                AsmArgsView a;
                std::memcpy(&a, ptr, sizeof(a));
                for (const void* q : {(const void*)a.rowTable, (const void*)a.state, (const void*)a.in, (const void*)a.out, (const void*)a.itram, (const void*)a.xtram,
                                      (const void*)a.lut, (const void*)a.tracks}) {
                    if (!q) continue;
                    for (const auto& d : g.deviceOf)
                        if (q >= d.first && q < static_cast<const char*>(d.first) + g.allocations[d.first] && d.second != moduleDevice) {
                            ++g.crossDevice;
                            return fail(hipErrorInvalidValue);
                        }
                }
            }
        }
    }
    if (gx == 0 || bx == 0 || bx > 1024 || sharedBytes > 160u * 1024u) return fail(hipErrorInvalidValue);
    std::vector<unsigned char> kernarg;
    if (extra) {   // {HIP_LAUNCH_PARAM_BUFFER_POINTER, ptr, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END}
        const void* ptr = nullptr;
        size_t size = 0;
        for (int k = 0; extra[k] != HIP_LAUNCH_PARAM_END && k < 8; k += 2) {
            if (extra[k] == HIP_LAUNCH_PARAM_BUFFER_POINTER) ptr = extra[k + 1];
            else if (extra[k] == HIP_LAUNCH_PARAM_BUFFER_SIZE) size = *static_cast<size_t*>(extra[k + 1]);
        }
        if (ptr && size) kernarg.assign(static_cast<const unsigned char*>(ptr), static_cast<const unsigned char*>(ptr) + size);   // (copied at launch, as HIP does)
    }
    streamOf(s)->push([name, kernarg] { runAsmKernel(name, kernarg); });
    return hipSuccess;
}

}  // extern "C"
