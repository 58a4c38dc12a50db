// xcu_pipeline.hip - what would a program pipelined over wavefronts on DIFFERENT compute units cost?  (the only
// lever left for small batches - a group's stages share one CU's LDS today, 64 of 256 CUs have work at 4 096 instances.)
//
// G groups of 64 lanes, each a chain of K stages; stage k of group g is a one-wavefront workgroup of its own (blockIdx = g * K + k:
// a group's stages are dispatched in order, all G * K workgroups are resident together).  Per sample a stage receives ROWS rows of
// 64 floats from the stage in front of it, runs a dependent chain of U x (16 / K) vector instructions on them (the filter chain's
// mix: fp32 multiply / add / med3 and an fp64 fma with its conversions), and hands ROWS rows on - through global memory:
//   producer  payload stores sc1 -> every BURST samples: s_waitcnt vmcnt(0) -> ONE lane stores the progress counter sc1
//   consumer  polls that counter with sc1 loads (bounded spin, s_sleep) before it requests a burst; payload loads sc1
// (MI355X_MICROARCH.md, inter-workgroup visibility, valid form "sc1 payload -> drained -> sc1 flag; sc1 flag poll; sc1 payload loads").
// The packets of a launch lie in a linear buffer [group][cut][sample][row][64]: no ring, the producer never waits.
// Every spin is bounded (kSpinLimit): a stage that times out says so in `timeouts` and goes on with whatever it reads.
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcu_pipeline.hip -o tools/micro/xcu_pipeline && tools/micro/xcu_pipeline
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#ifndef BURST_SAMPLES
#define BURST_SAMPLES 8
#endif
constexpr int ROWS = 2, BURST = BURST_SAMPLES, UNITS = 16;   // rows per packet; samples per burst / progress update; work units per sample of a whole group
constexpr unsigned kSpinLimit = 1u << 22;

__device__ __forceinline__ float ldSc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stSc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one work unit: the dependent chain of one filter section (interp + macs of the 64-instruction filter chain: 7 vector instructions)
__device__ __forceinline__ void unit(float& t, float& s, float in) {
    float p = 0.1f * t;
    double d = __builtin_fma(0.9, (double)s, (double)p);
    s = (float)d;
    float q = s + 0.05f * in;
    t = __builtin_fminf(__builtin_fmaxf(q, -1.0f), 1.0f);
}

template <int unitsPerStage>
__global__ void __launch_bounds__(64) chain(float* packets, unsigned* progress, unsigned* timeouts, const float* pcm, float* out, int K, int S) {
    const int g = blockIdx.x / K, k = blockIdx.x % K, lane = threadIdx.x;
    const size_t sampleStride = (size_t)(gridDim.x / K) * 64;   // one sample of every group (computed once: a 64-bit divide per sample would swamp the measurement)
    const size_t cutFloats = (size_t)S * ROWS * 64;
    const float* recv = packets + ((size_t)g * K + (k - 1)) * cutFloats + lane;   // cut k-1 (k > 0)
    float* send = packets + ((size_t)g * K + k) * cutFloats + lane;               // cut k (k < K-1)
    const unsigned* theirs = progress + (size_t)g * K + (k - 1);
    unsigned* mine = progress + (size_t)g * K + k;
    float state[unitsPerStage];
#pragma unroll
    for (int u = 0; u < unitsPerStage; ++u) state[u] = 0.0f;
    float nextT[BURST], nextIn[BURST], curT[BURST], curIn[BURST];
    unsigned spins = 0;
    auto waitFor = [&](unsigned samples) {   // until the stage in front has published `samples` samples
        if (k == 0) return;
        unsigned seen = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (seen < samples && spins < kSpinLimit) {
            __builtin_amdgcn_s_sleep(2);
            ++spins;
            seen = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto request = [&](int first) {          // samples first .. first + BURST - 1 into next* (S is a multiple of BURST; no branch per
        if (k == 0) {                        //  sample, or the compiler waits for every load by itself)
#pragma unroll
            for (int j = 0; j < BURST; ++j) { nextIn[j] = pcm[(size_t)(first + j) * sampleStride + (size_t)g * 64 + lane]; nextT[j] = nextIn[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < BURST; ++j) {
                nextT[j] = ldSc1(recv + ((size_t)(first + j) * ROWS + 0) * 64);
                nextIn[j] = ldSc1(recv + ((size_t)(first + j) * ROWS + 1) * 64);
            }
        }
    };
    waitFor(BURST);
    request(0);
    for (int s0 = 0; s0 < S; s0 += BURST) {
#pragma unroll
        for (int j = 0; j < BURST; ++j) { curT[j] = nextT[j]; curIn[j] = nextIn[j]; }
        if (s0 + BURST < S) {
            waitFor((unsigned)(s0 + 2 * BURST));
            request(s0 + BURST);             // lands behind this burst's work
        }
        if (k + 1 < K) {
#pragma unroll
            for (int j = 0; j < BURST; ++j) {
                float t = curT[j];
                const float in = curIn[j];
#pragma unroll
                for (int u = 0; u < unitsPerStage; ++u) unit(t, state[u], in);
                stSc1(send + ((size_t)(s0 + j) * ROWS + 0) * 64, t);
                stSc1(send + ((size_t)(s0 + j) * ROWS + 1) * 64, in);
            }
        } else {
#pragma unroll
            for (int j = 0; j < BURST; ++j) {
                float t = curT[j];
                const float in = curIn[j];
#pragma unroll
                for (int u = 0; u < unitsPerStage; ++u) unit(t, state[u], in);
                out[(size_t)(s0 + j) * sampleStride + (size_t)g * 64 + lane] = t;
            }
        }
        if (k + 1 < K) {
            __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this burst's payload stores have left
            if (lane == 0) __hip_atomic_store(mine, (unsigned)(s0 + BURST), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (spins >= kSpinLimit && lane == 0) atomicAdd(timeouts, 1u);
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 64, S = (argc > 2 ? atoi(argv[2]) : 2048) / BURST * BURST;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("%s: %d groups of 64 lanes, %d samples in bursts of %d, %d work units (7 dependent vector instructions each) per sample and group\n", prop.name, G, S, BURST, UNITS);
    float *packets, *pcm, *out;
    unsigned *progress, *timeouts;
    const int maxK = 16;
    hipMalloc(&packets, (size_t)G * maxK * S * ROWS * 64 * 4);
    hipMalloc(&pcm, (size_t)S * G * 64 * 4);
    hipMalloc(&out, (size_t)S * G * 64 * 4);
    hipMalloc(&progress, (size_t)G * maxK * 4);
    hipMalloc(&timeouts, 4);
    std::vector<float> h((size_t)S * G * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.9f * (float)((int)((i * 2654435761u) >> 8 & 0xffff) - 32768) / 32768.0f;
    hipMemcpy(pcm, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> ref;
    for (int K : {1, 2, 4, 8, 16}) {
        float best = 1e30f;
        unsigned to = 0;
        for (int rep = 0; rep < 4; ++rep) {
            hipMemset(progress, 0, (size_t)G * maxK * 4);
            hipMemset(timeouts, 0, 4);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            switch (K) {
                case 1: hipLaunchKernelGGL(chain<16>, dim3(G * K), dim3(64), 0, 0, packets, progress, timeouts, pcm, out, K, S); break;
                case 2: hipLaunchKernelGGL(chain<8>, dim3(G * K), dim3(64), 0, 0, packets, progress, timeouts, pcm, out, K, S); break;
                case 4: hipLaunchKernelGGL(chain<4>, dim3(G * K), dim3(64), 0, 0, packets, progress, timeouts, pcm, out, K, S); break;
                case 8: hipLaunchKernelGGL(chain<2>, dim3(G * K), dim3(64), 0, 0, packets, progress, timeouts, pcm, out, K, S); break;
                default: hipLaunchKernelGGL(chain<1>, dim3(G * K), dim3(64), 0, 0, packets, progress, timeouts, pcm, out, K, S); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
            hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost);
        }
        std::vector<float> y((size_t)S * G * 64);
        hipMemcpy(y.data(), out, y.size() * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        if (K == 1) ref = y;
        else for (size_t i = 0; i < y.size(); ++i) bad += (y[i] != ref[i]);
        const double nsPerSample = best * 1e6 / S;
        printf("K = %2d stages on %4d workgroups: %8.3f ms, %7.1f ns per sample = %6.0f clocks at 2.35 GHz (%5.1f per instruction of a stage's %3d); outputs differing from K = 1: %zu; timeouts %u\n",
               K, G * K, best, nsPerSample, nsPerSample * 2.35, nsPerSample * 2.35 / (7.0 * UNITS / K + 6), 7 * UNITS / K, bad, to);
    }
    return 0;
}
