// fx_kernel_stub.cpp — host stand-ins for the launch functions of csrc/fx_kernel.hip (TEST INFRASTRUCTURE, see hip_stub.cpp).
// The two small kernels whose RESULTS the host logic depends on do their real work, in stream order, on the stand-in's
// "device" memory: fill rows (register writes, state initialisation) and the row reduction (instruction counter, flags).
// The HIP C++ interpreter kernel itself only takes time and touches its buffers, like the assembly kernels' stand-in.
#include <chrono>
#include <cstring>
#include <functional>
#include <thread>

#include "../../fx8010-emulator-core_amd/csrc/fx_kernel.hpp"

void fxstubEnqueue(hipStream_t stream, std::function<void()> op);   // hip_stub.cpp
extern "C" void fxstub_set_kernel_micros(int);

namespace fx {

hipError_t launchStepBlock(const KernelArgs& args, bool, hipStream_t stream) {
    const KernelArgs a = args;
    fxstubEnqueue(stream, [a] {
        std::this_thread::sleep_for(std::chrono::microseconds(150));
        volatile uint32_t sink = 0;
        for (int k = 0; k < a.nLoad + a.nStore + a.nZero; ++k) sink = sink + a.rowTable[k];
        for (int k = 0; k < a.nOps * 8; ++k) sink = sink + a.steady[k] + a.last[k];
        const size_t count = (size_t)a.nSamples * (size_t)a.channels * (size_t)a.n;
        if (a.in && a.out && count) std::memcpy(a.out, a.in, count * sizeof(float));
        if (a.state && a.nPad > 0) { a.state[0] = a.state[0]; a.state[(size_t)a.nPad - 1] = a.state[(size_t)a.nPad - 1]; }
        (void)sink;
    });
    return hipGetLastError();   // as the real helpers do after hipLaunchKernelGGL: a stale "last error" of the thread comes back here
}

hipError_t launchFillRows(uint32_t* state, long long nPad, const uint32_t* d_rows, const uint32_t* d_values, int nRows, hipStream_t stream) {
    fxstubEnqueue(stream, [=] {
        for (int k = 0; k < nRows; ++k) {
            uint32_t* row = state + (size_t)d_rows[k] * (size_t)nPad;
            for (long long i = 0; i < nPad; ++i) row[i] = d_values[k];
        }
    });
    return hipGetLastError();   // as the real helpers do after hipLaunchKernelGGL: a stale "last error" of the thread comes back here
}

hipError_t launchReduceRow(const uint32_t* state, long long nPad, long long n, int rowLo, int rowHi, int rowOr, unsigned long long* d_sum, uint32_t* d_or,
                           hipStream_t stream) {
    fxstubEnqueue(stream, [=] {
        unsigned long long sum = 0;
        uint32_t all = 0;
        for (long long i = 0; i < n; ++i) {
            sum += (unsigned long long)state[(size_t)rowLo * (size_t)nPad + (size_t)i] | ((unsigned long long)state[(size_t)rowHi * (size_t)nPad + (size_t)i] << 32);
            all |= state[(size_t)rowOr * (size_t)nPad + (size_t)i];
        }
        *d_sum += sum;
        *d_or |= all;
    });
    return hipGetLastError();   // as the real helpers do after hipLaunchKernelGGL: a stale "last error" of the thread comes back here
}

}  // namespace fx
