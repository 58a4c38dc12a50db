// fx8010_demo.cpp — console harness over the drop-in class, in the spirit of the reference's
// source/main.cpp: build a test signal, step a .da program one AUDIOBLOCKSIZE block, change a
// control every 8 samples, print timing, instruction count, a register, metadata and controls.
// A second part steps a batch of instances through FX8010Batch and prints its throughput.
//
//   make -C fx8010-emulator-core_amd/csrc demo     (plain g++: the header needs no HIP toolchain)
//   fx8010-emulator-core_amd/host/fx8010_demo program.da [instances]
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
        batch.process(bin.data(), bout.data(), S);
        const double ms = batch.lastKernelMs();
        const double instr = (double)(batch.getInstructionCounter() - c0);
        std::cout << n << " instances x " << S << " samples: kernel " << ms << " ms, " << instr / (ms * 1e-3) / 1e6 << " emulated MIPS\n";
    } catch (const std::exception& e) {
        std::cerr << e.what() << "\n";
        return 1;
    }
    return 0;
}
