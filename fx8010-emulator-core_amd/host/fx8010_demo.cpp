// fx8010_demo.cpp — console harness over the drop-in class, in the spirit of the reference's
// source/main.cpp: build a test signal, step a .da program one AUDIOBLOCKSIZE block, change a
// control every 8 samples, print timing, instruction count, a register, metadata and controls.
// A second part steps a batch of instances through FX8010Batch and prints its throughput; a third one is the reference's
// real-time question asked of a batch: AUDIOBLOCKSIZE-sample blocks against the 666.667 us budget the harness prints
// (source/main.cpp:155), a slider moving every 8th block, PCM in pinned host buffers (processed in place).
//
//   make -C fx8010-emulator-core_amd/csrc demo     (plain g++: the header needs no HIP toolchain)
//   fx8010-emulator-core_amd/host/fx8010_demo program.da [instances [realtime-blocks [control]]]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "FX8010.h"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s program.da [instances]\n", argv[0]);
        return 2;
    }
    const int numChannels = 1;
    try {
        Klangraum::FX8010 fx(numChannels);
        if (!fx.loadFile(argv[1])) {
            std::cout << "load failed:\n";
            for (const auto& e : fx.getErrorList()) std::cout << "  " << e.errorDescription << " (" << e.errorRow << ")\n";
            return 1;
        }
        // bipolar ramp -1 .. +1, the stimulus the reference harness uses for LOG/EXP
        std::vector<float> ramp;
        for (int i = -Klangraum::kAudioBlockSize / 2; i < Klangraum::kAudioBlockSize / 2; ++i) ramp.push_back((float)i / (Klangraum::kAudioBlockSize / 2.0f));
        const float sliders[4] = {0.1f, 0.25f, 0.5f, 1.0f};
        std::vector<float> in(numChannels), out;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < Klangraum::kAudioBlockSize; ++i) {
            if (i % 8 == 0) fx.setRegisterValue("volume", sliders[i / 8]);
            in[0] = ramp[i];
            out = fx.process(in);
            std::cout << ramp[i] << "," << out[0] << "\n";
        }
        auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "one instance, " << Klangraum::kAudioBlockSize << " process() calls: " << us << " us, " << fx.getInstructionCounter()
                  << " instructions (real-time budget " << 1e6 * Klangraum::kAudioBlockSize / Klangraum::kSampleRate << " us)\n";
        std::cout << "filter_cutoff = " << fx.getRegisterValue("filter_cutoff") << "\n";
        for (const auto& kv : fx.getMetaData()) std::cout << kv.first << ": " << kv.second << "\n";
        for (const auto& c : fx.getControlRegisters()) std::cout << "control: " << c << "\n";

        const int64_t n = argc > 2 ? std::atoll(argv[2]) : 65536;
        const int S = 256;
        Klangraum::FX8010Batch batch(n, numChannels);
        if (!batch.loadFile(argv[1])) return 1;
        std::vector<float> bin((size_t)S * n), bout((size_t)S * n);
        for (int s = 0; s < S; ++s)
            for (int64_t k = 0; k < n; ++k) bin[(size_t)s * n + k] = ramp[(s + k) % Klangraum::kAudioBlockSize] * 0.9f;
        batch.process(bin.data(), bout.data(), S);  // warm-up (also uploads the program)
        const int64_t c0 = batch.getInstructionCounter();
        const auto p0 = std::chrono::steady_clock::now();
        batch.process(bin.data(), bout.data(), S);
        // (the call's own time, copies over PCIe included: a large host block runs as several overlapping pieces, so the last
        // launch's kernel time alone would say nothing about the block)
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - p0).count();
        const double instr = (double)(batch.getInstructionCounter() - c0);
        std::cout << n << " instances x " << S << " samples from host buffers: " << ms << " ms per block, " << instr / (ms * 1e-3) / 1e6 << " emulated MIPS\n";
        // ---- real time: blocks of AUDIOBLOCKSIZE samples, each done before the next one is due?
        const int blocks = argc > 3 ? std::atoi(argv[3]) : 0;
        if (blocks > 0) {
            const std::string control = argc > 4 ? argv[4] : "volume";
            const int B = Klangraum::kAudioBlockSize;
            const double budgetUs = 1e6 * B / Klangraum::kSampleRate;
            auto pin = batch.pcmBuffer(B), pout = batch.pcmBuffer(B);       // pinned host memory: no staging copies
            for (int s = 0; s < B; ++s)
                for (int64_t k = 0; k < n; ++k) pin.data()[(size_t)s * n + k] = ramp[(s + k) % B] * 0.9f;
            batch.prepare(B);                                                // the code for B-sample blocks exists before the stream starts
            std::vector<double> us2;
            for (int k = 0; k < blocks + 100; ++k) {
                const auto b0 = std::chrono::steady_clock::now();
                if (k % 8 == 0) batch.setRegisterValue(control, sliders[(k / 8) % 4]);
                batch.process(pin.data(), pout.data(), B);
                const double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - b0).count();
                if (k >= 100) us2.push_back(t);                              // (the first hundred: first control touch, clocks)
            }
            std::sort(us2.begin(), us2.end());
            auto at = [&](double q) { return us2[std::min(us2.size() - 1, (size_t)(q * us2.size()))]; };
            const size_t late = us2.end() - std::upper_bound(us2.begin(), us2.end(), budgetUs);
            std::cout << "real time: " << n << " instances, " << blocks << " blocks of " << B << " samples from pinned host buffers: median " << at(0.5) << " us, p99 "
                      << at(0.99) << " us, max " << us2.back() << " us, budget " << budgetUs << " us, " << late << " late (" << batch.tierNote() << ")\n";
        }
    } catch (const std::exception& e) {
        std::cerr << e.what() << "\n";
        return 1;
    }
    return 0;
}
