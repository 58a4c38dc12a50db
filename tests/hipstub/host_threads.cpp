// host_threads.cpp — the host threads of libfx8010_amd.so driven without a GPU, for ThreadSanitizer and AddressSanitizer
// (TEST INFRASTRUCTURE: csrc/Makefile `tsan` / `stubasan` link this file with the library's host sources and tests/hipstub/).
//
// The reference is single-threaded (/root/reference/include/FX8010.h:47-75: plain members, no locks); everything concurrent in
// this library is its own: a builder thread per handle that generates code ahead of the caller (fx_batch.cpp Batch::Builder), a
// cache of finished code handed across threads, a tuner that times the caller's launches while the builder loads modules, one
// worker thread + mailbox per shard of a multi-device handle (fx_shard.cpp) - and the stand-in's stream threads, which play the
// GPU.  Scenarios:
//   controls      builder + cache eviction while the caller keeps lowering: block-length classes, first control touch, compiled-in
//                 values that change the key (eviction), fxb_prepare with and without waiting, a load while builds are queued
//                 (drainBuilder), destruction with the builder busy
//   queued        tuner trials with launches queued back to back on device-resident buffers (no sync per block)
//   hostpipe      a host block cut into pieces on three streams (copy-in, kernel, copy-out overlapping), a launch failing in the middle
//   shards        a 3-shard handle with two host threads posting: blocks on one, register / counter reads on the other
//   handles       independent handles on independent threads (process-wide tables: templates, interpreter module, LUTs)
//   memory        a load whose delay memory cannot be allocated -> FX_E_MEMORY, handle stays usable; 50 create / load / run /
//                 destroy cycles leave the allocation count where it was
//   modules       a module load that fails on the caller's thread is reported and the next call recovers; one that fails on the
//                 builder thread is retried on the caller's; the builder's list of failures stays bounded
//   images        damaged state images (fxb_load_state) are refused before any address is computed from them
// Exit code 0 = every check held (a sanitizer report turns it non-zero by itself).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fx8010_amd.h"
#include "hip_stub.h"

namespace {

int g_failures = 0;
#define CHECK(cond)                                                                      \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            ++g_failures;                                                                \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
        }                                                                                \
    } while (0)

// a filter chain that can be cut into stages (small batches), with declared controls, a LOG whose table number is a control
// (a value that shapes the code: every change is a new key) and a delay line
std::string chainProgram(int links, bool withTram) {
    std::string t = "name \"host threads\"\n";
    if (withTram) t += "itramsize 50 \n";
    t += "input in 0\noutput out 0\ncontrol vol = 0.5\ncontrol mix = 0.25\ncontrol cut = 0.1\ncontrol tbl = 3\nstatic t\nstatic a\nstatic rd\n";
    for (int k = 0; k < links; ++k) t += "static s" + std::to_string(k) + "\n";
    if (withTram) t += "idelay read, rd, at, 0\n";
    t += "log a, in, tbl, 0\n";
    t += "interp s0, s0, cut, a\nmacs t, 0, s0, vol\n";
    for (int k = 1; k < links; ++k) t += "interp s" + std::to_string(k) + ", s" + std::to_string(k) + ", cut, t\nmacs t, 0, s" + std::to_string(k) + ", 0.5\n";
    if (withTram) t += "macs t, t, rd, mix\nidelay write, t, at, 0\n";
    t += "macs out, t, mix, 0.125\nend";
    return t;
}

std::vector<float> ramp(size_t n) {
    std::vector<float> x(n);
    for (size_t i = 0; i < n; ++i) x[i] = (float)((int)(i * 2654435761u >> 8 & 0xffff) - 32768) / 40000.0f;
    return x;
}

void scenarioControls() {
    const int64_t N = 300;
    fxb_handle* h = fxb_create(N, 1, 0);
    CHECK(h != nullptr);
    if (!h) return;
    CHECK(fxb_load_text(h, chainProgram(14, false).c_str()) == 1);
    const std::vector<float> x = ramp((size_t)N * 512);
    std::vector<float> y(x.size());
    const int lengths[5] = {32, 32, 128, 512, 7};
    for (int it = 0; it < 260; ++it) {
        const int S = lengths[it % 5];
        CHECK(fxb_process_block(h, x.data(), y.data(), S) == 0);
        CHECK(std::memcmp(x.data(), y.data(), (size_t)S * N * 4) == 0);   // (the stand-in kernel copies in to out: the launch did run on these buffers)
        if (it % 3 == 0) CHECK(fxb_set_register(h, it % 6 ? "vol" : "mix", 0.1f + 0.01f * (float)(it % 50)) == 0);
        if (it % 7 == 0) CHECK(fxb_set_register_i(h, "cut", it % N, 0.2f) == 0);
        if (it % 11 == 0) CHECK(fxb_set_register(h, "tbl", (float)(it / 11 % 13)) == 0);   // compiled in: a new key each time -> eviction
        if (it % 13 == 0) CHECK(fxb_get_register_i(h, "vol", 5) > 0.0f);
        if (it % 50 == 49) CHECK(fxb_prepare(h, lengths[(it / 50) % 4], it % 100 == 99) == 0);
        if (it == 120) {
            // a further load while builds may be queued (drainBuilder): the reference accumulates declarations and instructions
            CHECK(fxb_load_text(h, "static z0\nstatic z1\ninterp z0, z0, cut, out\nmacs z1, z0, vol, 0.5\nend") == 1);
        }
    }
    CHECK(fxb_info(h, FXB_INFO_CODE_CACHED) <= 9);   // kCodeCache + the one in force
    CHECK(fxb_info(h, FXB_INFO_XLATE_BACKGROUND_BUILDS) >= 1);
    std::printf("  controls: %lld translations on the caller's thread, %lld on the builder's, %lld cache hits, %lld code objects held, %lld stage trials\n",
                (long long)fxb_info(h, FXB_INFO_XLATE_BUILDS), (long long)fxb_info(h, FXB_INFO_XLATE_BACKGROUND_BUILDS), (long long)fxb_info(h, FXB_INFO_CODE_CACHE_HITS),
                (long long)fxb_info(h, FXB_INFO_CODE_CACHED), (long long)fxb_info(h, FXB_INFO_STAGE_TRIALS));
    char note[256];
    CHECK(fxb_tier_note(h, note, sizeof(note)) > 0);
    // destroy with the builder (possibly) busy: ask for builds, leave at once
    CHECK(fxb_set_register(h, "tbl", 5.0f) == 0);
    CHECK(fxb_process_block(h, x.data(), y.data(), 300) == 0);
    fxb_destroy(h);
}

void scenarioQueued() {
    const int64_t N = 300;
    const int S = 64;
    fxb_handle* h = fxb_create(N, 1, 0);
    CHECK(h != nullptr);
    if (!h) return;
    CHECK(fxb_load_text(h, chainProgram(20, false).c_str()) == 1);
    float *dIn = nullptr, *dOut = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void**>(&dIn), (size_t)S * N * 4) == hipSuccess);
    CHECK(hipMalloc(reinterpret_cast<void**>(&dOut), (size_t)S * N * 4) == hipSuccess);
    hipStream_t stream = nullptr;
    CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess);
    for (int it = 0; it < 120; ++it) {
        CHECK(fxb_process_block_dev(h, dIn, dOut, S, stream) == 0);   // no sync: launches pile up on the stream, the tuner reads what it can
        if (it == 60) CHECK(fxb_set_register(h, "vol", 0.75f) == 0);
    }
    CHECK(fxb_sync(h) == 0);
    CHECK(hipStreamSynchronize(stream) == hipSuccess);
    CHECK(fxb_last_kernel_ms(h) >= 0.0f);
    CHECK(fxb_info(h, FXB_INFO_STAGE_TRIALS) >= 0);
    std::printf("  queued: %lld stage trials, %lld translations on the builder's thread, %d wavefronts per workgroup in force\n", (long long)fxb_info(h, FXB_INFO_STAGE_TRIALS),
                (long long)fxb_info(h, FXB_INFO_XLATE_BACKGROUND_BUILDS), (int)fxb_info(h, FXB_INFO_WAVES_PER_WG));
    fxb_destroy(h);
    CHECK(hipStreamDestroy(stream) == hipSuccess);
    CHECK(hipFree(dIn) == hipSuccess);
    CHECK(hipFree(dOut) == hipSuccess);
}

// a host block large enough to be cut into pieces whose copy-in, kernel and copy-out overlap on three streams
// (Batch::processHostPipelined): 4 096 instances x 2 048 samples = 32 MiB each way -> eight pieces
void scenarioHostPipe() {
    const int64_t N = 4096;
    const int S = 2048;
    fxb_handle* h = fxb_create(N, 1, 0);
    CHECK(h != nullptr);
    if (!h) return;
    CHECK(fxb_load_text(h, chainProgram(6, true).c_str()) == 1);
    const std::vector<float> x = ramp((size_t)N * S);
    for (int it = 0; it < 6; ++it) {
        std::vector<float> y(x.size(), -7.0f);
        CHECK(fxb_process_block(h, x.data(), y.data(), S) == 0);
        CHECK(std::memcmp(x.data(), y.data(), x.size() * 4) == 0);   // every piece came back (the stand-in kernel copies in to out)
        if (it == 2) CHECK(fxb_set_register(h, "vol", 0.8f) == 0);
        if (it == 3) CHECK(fxb_process_block(h, x.data(), y.data(), 7) == 0);   // a short block in between: the plain path
    }
    // the caller's buffers in pinned host memory: the kernel works on them in place (no staging copies)
    {
        float *px = nullptr, *py = nullptr;
        CHECK(hipHostMalloc(reinterpret_cast<void**>(&px), x.size() * 4, 0) == hipSuccess);
        CHECK(hipHostMalloc(reinterpret_cast<void**>(&py), x.size() * 4, 0) == hipSuccess);
        std::memcpy(px, x.data(), x.size() * 4);
        const long before = fxstub_kernels_run();
        CHECK(fxb_process_block(h, px, py, S) == 0);
        CHECK(fxstub_kernels_run() == before + 1);                   // one launch for the whole block
        CHECK(std::memcmp(px, py, x.size() * 4) == 0);
        CHECK(fxb_process_block(h, px, py, 32) == 0);
        CHECK(fxb_process_block(h, px + 64, py + 64, 32) == 0);     // (any address inside a pinned allocation)
        // ranges that overlap without being one buffer, and buffers only part of which is pinned: the staged copies (the whole
        // input is read before the first output is written; the kernel never sees memory it cannot address)
        {
            std::memcpy(px, x.data(), x.size() * 4);
            CHECK(fxb_process_block(h, px, px + N, 32) == 0);                                  // output one sample period behind the input
            CHECK(std::memcmp(px + N, x.data(), (size_t)N * 32 * 4) == 0);
            std::memcpy(px, x.data(), x.size() * 4);
            CHECK(fxb_process_block(h, px + 5 * N, px, 32) == 0);                              // ... five in front of it
            CHECK(std::memcmp(px, x.data() + 5 * N, (size_t)N * 32 * 4) == 0);
            CHECK(fxb_process_block(h, px, px, 32) == 0);                                      // one buffer: in place
            std::vector<float> half(x.begin(), x.begin() + (size_t)N * 64), back((size_t)N * 64, -3.0f);
            CHECK(hipHostRegister(half.data(), (size_t)N * 32 * 4, 0) == hipSuccess);           // the first 32 sample periods only
            CHECK(hipHostRegister(back.data(), back.size() * 4, 0) == hipSuccess);
            long launches = fxstub_kernels_run();
            CHECK(fxb_process_block(h, half.data(), back.data(), 32) == 0);                    // inside the registration: in place
            CHECK(fxstub_kernels_run() == launches + 1 && std::memcmp(half.data(), back.data(), (size_t)N * 32 * 4) == 0);
            CHECK(fxb_process_block(h, half.data(), back.data(), 64) == 0);                    // beyond it: staged
            CHECK(std::memcmp(half.data(), back.data(), half.size() * 4) == 0);
            CHECK(fxb_process_block(h, half.data() + (size_t)N * 16, back.data(), 32) == 0);   // straddling its end
            CHECK(std::memcmp(half.data() + (size_t)N * 16, back.data(), (size_t)N * 32 * 4) == 0);
            CHECK(hipHostUnregister(half.data()) == hipSuccess && hipHostUnregister(back.data()) == hipSuccess);
            CHECK(fxstub_bad_pcm_launches() == 0);
        }
        CHECK(hipHostFree(px) == hipSuccess);
        CHECK(hipHostFree(py) == hipSuccess);
        // ... and the library's own allocator for hosts without the HIP runtime (fxb_host_alloc)
        float* qx = static_cast<float*>(fxb_host_alloc((int64_t)N * 32 * 4));
        float* qy = static_cast<float*>(fxb_host_alloc((int64_t)N * 32 * 4));
        CHECK(qx && qy);
        if (qx && qy) {
            std::memcpy(qx, x.data(), (size_t)N * 32 * 4);
            const long launches = fxstub_kernels_run();
            CHECK(fxb_process_block(h, qx, qy, 32) == 0);
            CHECK(fxstub_kernels_run() == launches + 1 && std::memcmp(qx, qy, (size_t)N * 32 * 4) == 0);
        }
        fxb_host_free(qx);
        fxb_host_free(qy);
        fxb_host_free(nullptr);
        CHECK(fxb_host_alloc(0) == nullptr && fxb_host_alloc(-5) == nullptr);
    }
    // a failing piece: whatever went wrong, no copy may still touch the caller's buffers when the call returns
    fxstub_fail_launches(1, 1, (int)hipErrorLaunchFailure);
    {
        std::vector<float> xin(x), yout(x.size());
        const int rc = fxb_process_block(h, xin.data(), yout.data(), S);
        CHECK(rc == FX_E_NODEVICE);
    }   // (freed here)
    fxstub_fail_launches(-1, 0, 0);
    std::vector<float> y(x.size());
    CHECK(fxb_process_block(h, x.data(), y.data(), S) == 0);
    fxb_destroy(h);
}

// distinct: the three shards on three different devices (the stand-in has four: FXSTUB_DEVICES, set in main) - every module, stream
// and buffer of a shard must then be its own device's: the stand-in refuses a launch that mixes devices and counts it
void shardsOn(const int devices[3]) {
    const int64_t N = 300;
    const long mixed = fxstub_cross_device_errors();
    fxb_handle* h = fxb_create_on_devices(N, 1, devices, 3);
    CHECK(h != nullptr);
    if (!h) return;
    CHECK(fxb_shard_count(h) == 3);
    CHECK(fxb_load_text(h, chainProgram(6, true).c_str()) == 1);
    const std::vector<float> x = ramp((size_t)N * 32);
    std::atomic<bool> done{false};
    std::thread audio([&] {
        std::vector<float> y(x.size());
        for (int it = 0; it < 150; ++it) CHECK(fxb_process_block(h, x.data(), y.data(), 32) == 0);
        done = true;
    });
    std::thread ui([&] {
        int k = 0;
        while (!done) {
            (void)fxb_get_register_i(h, "vol", (k * 37) % N);
            if (k % 5 == 0) CHECK(fxb_set_register_i(h, "mix", (k * 11) % N, 0.3f) == 0);
            if (k % 9 == 0) CHECK(fxb_instruction_counter(h) >= 0);
            if (k % 17 == 0) (void)fxb_ood_flags(h);
            ++k;
        }
    });
    audio.join();
    ui.join();
    std::vector<float> values((size_t)N);
    CHECK(fxb_get_register_array(h, "mix", values.data()) == 0);
    // a state image taken from three devices loads back, and into a single-device handle
    const int64_t bytes = fxb_state_size(h);
    std::vector<unsigned char> img((size_t)std::max<int64_t>(bytes, 0));
    CHECK(bytes > 0 && fxb_save_state(h, img.data(), bytes) == 0 && fxb_load_state(h, img.data(), bytes) == 0);
    int dev = -1;
    int64_t first = -1, count = -1;
    CHECK(fxb_shard_info(h, 2, &dev, &first, &count) == 0 && dev == devices[2] && first == 256 && count == 44);
    fxb_destroy(h);
    CHECK(fxstub_cross_device_errors() == mixed);
}

void scenarioShards() {
    const int same[3] = {0, 0, 0}, distinct[3] = {1, 3, 2};
    shardsOn(same);
    shardsOn(distinct);
    // the caller's current device is its own business: a multi-shard handle must leave it alone
    int before = -1, after = -1;
    CHECK(hipSetDevice(3) == hipSuccess && hipGetDevice(&before) == hipSuccess);
    shardsOn(distinct);
    CHECK(hipGetDevice(&after) == hipSuccess && after == before);
    CHECK(hipSetDevice(0) == hipSuccess);
    // fxb_create_sharded by device mask: devices 0, 1 and 3
    fxb_handle* h = fxb_create_sharded(1000, 1, 0xbull);
    CHECK(h != nullptr);
    if (h) {
        CHECK(fxb_shard_count(h) == 3);
        CHECK(fxb_load_text(h, chainProgram(5, true).c_str()) == 1);
        const std::vector<float> x = ramp(1000 * 40);
        std::vector<float> y(x.size());
        for (int it = 0; it < 6; ++it) CHECK(fxb_process_block(h, x.data(), y.data(), 40) == 0);
        CHECK(std::memcmp(x.data(), y.data(), x.size() * 4) == 0);   // every shard's columns came back
        fxb_destroy(h);
    }
    CHECK(fxstub_cross_device_errors() == 0);
}

void scenarioHandles() {
    std::vector<std::thread> threads;
    for (int t = 0; t < 4; ++t)
        threads.emplace_back([t] {
            fxb_handle* h = fxb_create(200 + 64 * t, 1, 0);
            CHECK(h != nullptr);
            if (!h) return;
            CHECK(fxb_load_text(h, chainProgram(4 + 3 * t, t % 2 == 1).c_str()) == 1);
            const int64_t N = 200 + 64 * t;
            const std::vector<float> x = ramp((size_t)N * 48);
            std::vector<float> y(x.size());
            for (int it = 0; it < 30; ++it) {
                CHECK(fxb_process_block(h, x.data(), y.data(), it % 2 ? 48 : 16) == 0);
                if (it == 10) CHECK(fxb_set_register(h, "vol", 0.9f) == 0);
            }
            fxb_destroy(h);
        });
    for (std::thread& th : threads) th.join();
}

void scenarioMemory() {
    // (a) delay memory that cannot be allocated: 4 096 instances x 1 048 576 slots x 4 B = 16 GiB against a 64 MiB "device"
    fxstub_set_capacity(64ull << 20);
    fxb_handle* h = fxb_create(4096, 1, 0);
    CHECK(h != nullptr);
    if (h) {
        const std::string big = "xtramsize 1048576 \ninput in 0\noutput out 0\nstatic xd\nstatic a\nxdelay read, xd, at, 0\nmacs a, in, xd, 0.5\nxdelay write, a, at, 0\nmacs out, a, 0, 0\nend";
        CHECK(fxb_load_text(h, big.c_str()) == 1);
        const std::vector<float> x = ramp(4096 * 8);
        std::vector<float> y(x.size());
        const unsigned long long before = fxstub_bytes_in_use();
        CHECK(fxb_process_block(h, x.data(), y.data(), 8) == FX_E_MEMORY);
        CHECK(std::strlen(fxb_last_error(h)) > 0);
        CHECK(fxb_process_block(h, x.data(), y.data(), 8) == FX_E_MEMORY);      // (and again: nothing half-built is left in force)
        CHECK(fxstub_bytes_in_use() <= before + (1u << 20));                     // no leak of the attempt
        CHECK(fxb_prepare(h, 8, 1) == FX_E_MEMORY);
        // the handle stays usable: a further load shrinks the line (the reference's loader: the size is whatever the last
        // xtramsize line said), the accumulated program then fits
        CHECK(fxb_load_text(h, "xtramsize 64 \nend") == 1);
        CHECK(fxb_process_block(h, x.data(), y.data(), 8) == 0);
        CHECK(fxb_ood_flags(h) == 0);              // (the failed allocation must not come back as a later launch's "last error")
        CHECK(fxb_instruction_counter(h) >= 0);
        CHECK(fxb_set_register(h, "a", 0.5f) == 0);
        CHECK(fxb_info(h, FXB_INFO_XTRAM_SLOTS) >= 1 && fxb_info(h, FXB_INFO_XTRAM_SLOTS) <= 64);
        fxb_destroy(h);
    }
    fxstub_set_capacity(4ull << 30);
    // (b) 50 create / load / run / destroy cycles: allocations, modules and bytes come back to where they were
    long allocs0 = -1, modules0 = -1;
    unsigned long long bytes0 = 0;
    for (int cycle = 0; cycle < 51; ++cycle) {
        fxb_handle* c = fxb_create(500, 1, 0);
        CHECK(c != nullptr);
        if (!c) break;
        CHECK(fxb_load_text(c, chainProgram(5, true).c_str()) == 1);
        const std::vector<float> x = ramp(500 * 16);
        std::vector<float> y(x.size());
        CHECK(fxb_process_block(c, x.data(), y.data(), 16) == 0);
        CHECK(fxb_set_register(c, "vol", 0.3f) == 0);
        CHECK(fxb_process_block(c, x.data(), y.data(), 16) == 0);
        fxb_destroy(c);
        if (cycle == 0) { allocs0 = fxstub_live_allocations(); modules0 = fxstub_live_modules(); bytes0 = fxstub_bytes_in_use(); }   // (process-wide tables exist from here on)
    }
    CHECK(fxstub_live_allocations() == allocs0);
    CHECK(fxstub_live_modules() == modules0);
    CHECK(fxstub_bytes_in_use() == bytes0);
}

void scenarioModules() {
    const int64_t N = 300;
    const std::vector<float> x = ramp((size_t)N * 300);
    std::vector<float> y(x.size());
    // (a) the caller's own translation cannot be loaded: the call reports it, the next one recovers
    {
        fxb_handle* h = fxb_create(N, 1, 0);
        CHECK(h != nullptr);
        if (!h) return;
        CHECK(fxb_load_text(h, chainProgram(8, false).c_str()) == 1);
        fxstub_fail_module_loads(0, 1);
        // (the caller's buffers are its own again the moment a call returns, failed or not: freed at once here - a copy still
        // queued on the handle's stream would be a use-after-free)
        float* xin = new float[(size_t)N * 32];
        float* yout = new float[(size_t)N * 32];
        std::memcpy(xin, x.data(), (size_t)N * 32 * 4);
        const int rc = fxb_process_block(h, xin, yout, 32);
        delete[] xin;
        delete[] yout;
        CHECK(rc == FX_E_NODEVICE);
        CHECK(std::strstr(fxb_last_error(h), "loading the translated program") != nullptr);
        CHECK(fxb_process_block(h, x.data(), y.data(), 32) == 0);
        CHECK(fxb_info(h, FXB_INFO_KERNEL) >= 9);
        fxb_destroy(h);
    }
    // (b) a builder-thread image fails to load: nothing is reported (it was speculative), the caller's thread builds the
    // variant itself when it is wanted
    {
        fxb_handle* h = fxb_create(N, 1, 0);
        CHECK(h != nullptr);
        if (!h) return;
        CHECK(fxb_load_text(h, chainProgram(8, false).c_str()) == 1);
        fxstub_fail_module_loads(1, 64);                                   // the caller's first load works, everything the builder tries fails
        CHECK(fxb_prepare(h, 32, 1) == 0);                                 // (waits until the builder has given up on its queue)
        CHECK(fxb_info(h, FXB_INFO_XLATE_BACKGROUND_BUILDS) == 0);
        fxstub_fail_module_loads(-1, 0);
        const int64_t mine = fxb_info(h, FXB_INFO_XLATE_BUILDS);
        CHECK(fxb_set_register(h, "vol", 0.7f) == 0);                      // first touch of a control: the variant the builder failed to make
        CHECK(fxb_process_block(h, x.data(), y.data(), 32) == 0);
        CHECK(fxb_info(h, FXB_INFO_XLATE_BUILDS) == mine + 1);             // ... built here, on the caller's thread
        CHECK(fxb_info(h, FXB_INFO_KERNEL) >= 9);
        // many failing requests: the builder's list of failures stays bounded (it is cleared by a load and capped in between)
        fxstub_fail_module_loads(0, 1 << 20);
        for (int k = 0; k < 80; ++k) {
            CHECK(fxb_set_register(h, "tbl", (float)(k % 13)) == 0);
            (void)fxb_process_block(h, x.data(), y.data(), k % 2 ? 32 : 300);   // (may fail: its own load fails too)
        }
        fxstub_fail_module_loads(-1, 0);
        CHECK(fxb_process_block(h, x.data(), y.data(), 32) == 0);
        fxb_destroy(h);
    }
}

void scenarioImages() {
    const int64_t N = 130;
    fxb_handle* h = fxb_create(N, 1, 0);
    CHECK(h != nullptr);
    if (!h) return;
    CHECK(fxb_load_text(h, chainProgram(3, true).c_str()) == 1);
    const std::vector<float> x = ramp((size_t)N * 16);
    std::vector<float> y(x.size());
    CHECK(fxb_process_block(h, x.data(), y.data(), 16) == 0);
    const int64_t bytes = fxb_state_size(h);
    CHECK(bytes > 64);
    std::vector<unsigned char> img((size_t)bytes);
    CHECK(fxb_save_state(h, img.data(), bytes) == 0);
    CHECK(fxb_load_state(h, img.data(), bytes) == 0);
    struct Header { uint32_t magic, version; int64_t n; int32_t channels, nRegs, stateRows, iSlots, xSlots, reserved[7]; } hdr;
    static_assert(sizeof(Header) == 64, "header");
    std::memcpy(&hdr, img.data(), sizeof(hdr));
    auto refused = [&](const Header& bad, int64_t size) {
        // an exactly-sized heap copy: a read beyond the image (or in front of it) is a sanitizer report
        std::vector<unsigned char> copy(img.begin(), img.begin() + (size < bytes ? size : bytes));
        if (copy.size() >= sizeof(bad)) std::memcpy(copy.data(), &bad, sizeof(bad));
        CHECK(fxb_load_state(h, copy.data(), (int64_t)copy.size()) == FX_E_ARG);
        CHECK(std::strlen(fxb_last_error(h)) > 0);
    };
    Header b = hdr;
    b.iSlots = -hdr.iSlots; refused(b, bytes); b = hdr;
    b.xSlots = -100; refused(b, bytes); b = hdr;
    b.iSlots = -(hdr.stateRows + 7); refused(b, bytes); b = hdr;
    b.iSlots = -(hdr.stateRows + 7); refused(b, 64 + N * 4 * 2); b = hdr;
    b.xSlots = INT32_MIN; refused(b, bytes); b = hdr;
    b.iSlots = 1 << 30; refused(b, bytes); b = hdr;
    b.version = 2; refused(b, bytes); b = hdr;
    b.magic = 0; refused(b, bytes); b = hdr;
    b.n = N + 1; refused(b, bytes); b = hdr;
    b.n = -N; refused(b, bytes); b = hdr;
    b.channels = 0; refused(b, bytes); b = hdr;
    b.nRegs = hdr.nRegs + 1; refused(b, bytes); b = hdr;
    b.stateRows = hdr.stateRows + 1; refused(b, bytes); b = hdr;
    b.stateRows = -1; refused(b, bytes); b = hdr;
    b.iSlots = hdr.iSlots + 1; refused(b, bytes); b = hdr;
    refused(b, bytes - 4);
    refused(b, 64);
    refused(b, 10);
    CHECK(fxb_load_state(h, img.data(), bytes) == 0);
    CHECK(fxb_process_block(h, x.data(), y.data(), 16) == 0);
    fxb_destroy(h);
}

}  // namespace

int main(int argc, char** argv) {
    setenv("FXSTUB_DEVICES", "4", 1);   // (read by the stand-in at its first call)
    struct { const char* name; void (*fn)(); } all[] = {
        {"controls", scenarioControls}, {"queued", scenarioQueued}, {"hostpipe", scenarioHostPipe}, {"shards", scenarioShards}, {"handles", scenarioHandles},
        {"memory", scenarioMemory}, {"modules", scenarioModules}, {"images", scenarioImages},
    };
    // arguments: scenario names (none = all) and --kernel-us=N (how long the stand-in kernels take: other interleavings)
    int kernelUs = 60;
    std::vector<std::string> wantedNames;
    for (int k = 1; k < argc; ++k) {
        if (std::strncmp(argv[k], "--kernel-us=", 12) == 0) kernelUs = std::atoi(argv[k] + 12);
        else wantedNames.push_back(argv[k]);
    }
    fxstub_set_kernel_micros(kernelUs);
    for (auto& s : all) {
        bool wanted = wantedNames.empty();
        for (const std::string& w : wantedNames) wanted = wanted || w == s.name;
        if (!wanted) continue;
        const int before = g_failures;
        s.fn();
        std::printf("%-9s %s\n", s.name, g_failures == before ? "ok" : "FAILED");
        std::fflush(stdout);
    }
    std::printf("%d check(s) failed\n", g_failures);
    return g_failures ? 1 : 0;
}
