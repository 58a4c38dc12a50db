// fx_kernel.hpp — launch interface of the interpreter kernel (device code: fx_kernel.hip).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace fx {

constexpr int kMaxChannels = 4;
constexpr int kPassCap = 64;  // multipass safety net: the reference would spin forever
constexpr int kLdsBytesPerCU = 160 * 1024;

// sticky per-instance "outside the parity domain" bits (same values as the oracle's FXO_OOD_*)
enum : uint32_t {
    OOD_TRAM_READ_NEG = 1u << 0,
    OOD_TRAM_WRITE_OOB = 1u << 1,
    OOD_TRAM_SIZE0 = 1u << 2,
    OOD_LUT_TABLE = 1u << 3,
    OOD_LUT_INDEX = 1u << 4,
    OOD_PASS_CAP = 1u << 5
};

struct KernelArgs {
    const uint32_t* steady;    // device opcode stream, samples 0..S-2 (8 dwords per record)
    const uint32_t* last;      // stream for the final sample of the block
    const uint32_t* rowTable;  // nLoad + nStore entries ldsRow | stateRow << 16, then nZero LDS rows to clear
    uint32_t* state;           // [rows][nPad] 32-bit words, instance-fastest
    const float* in;           // [S][CH][N]
    float* out;                // [S][CH][N]
    float* itram;              // [wave][iSlots][64][K]
    float* xtram;              // [wave][xSlots][64][K]
    const double* lut;         // LutDevice blob (fx_model.hpp): thresholds, x1, per-table {slope, y1}
    long long n;               // instances
    long long nPad;            // n rounded up to a multiple of 256
    int nOps, nLoad, nStore, nZero;
    int nSamples, channels;
    int inRow[kMaxChannels];     // LDS row of channel's input sample, -1 unused
    int latchRow[kMaxChannels];  // LDS row of channel's output latch
    int iSlots, xSlots, iSize, xSize;
    // byte offsets (row * 256 * K) of the bookkeeping rows; valid only when the program needs them
    uint32_t skipOff;    // numSkip, ran, dynCount (3 rows)      — programs with SKIP shadows
    uint32_t cursorOff;  // iTRAM write, iTRAM read, xTRAM write, xTRAM read (4 rows)
    uint32_t noiseOff;   // g_x1, g_x2 (2 rows)
    uint32_t oodOff;     // sticky out-of-domain flags (1 row, always)
    uint32_t aliveOff;   // alive, ended (2 rows)                 — multipass programs
    int oodRow, countLo, countHi;  // state rows
    int staticCount;     // unshadowed instructions per sample
    int nRows;
    int hasShadow;
    int instPerLane;     // K: 1, 2 or 4
    int tramDane;        // opt-in DANE delay-line model: the two write-cursor rows are per-sample address counters
};

// grid = ceil(n / (64*K)) workgroups of one wavefront; dynamic LDS = nRows*256*K bytes.
hipError_t launchStepBlock(const KernelArgs& a, bool multipass, hipStream_t stream);

// state[row][0..nPad) = value for each listed row (initialisation / broadcast set)
hipError_t launchFillRows(uint32_t* state, long long nPad, const uint32_t* d_rows, const uint32_t* d_values,
                          int nRows, hipStream_t stream);
// sum / OR reductions over state rows (instruction counter, ood flags)
hipError_t launchReduceRow(const uint32_t* state, long long nPad, long long n, int rowLo, int rowHi, int rowOr,
                           unsigned long long* d_sum, uint32_t* d_or, hipStream_t stream);

}  // namespace fx
