// fx8010_wav — WAV block front-end of the batch emulator (the reference processes one sample per
// call from its console harness, source/main.cpp; its README names a VST/WAV front-end as the goal).
//
//   fx8010_wav program.da in.wav out.wav [--instances N] [--block S] [--pick K]
//              [--set name=value]... [--sweep name=lo:hi]...
//
// Runs N instances of the program over the file (every instance hears the same input; --sweep gives instance i
// the control value lo + (hi - lo) * i / (N - 1), --set the same value to all) and writes instance K's output as
// a 32-bit float WAV.  Input: RIFF/WAVE, PCM 16-bit or IEEE float 32-bit, 1..4 channels = the program's channels.
// 16-bit samples become s / 32768.0f exactly.  Built with plain g++ over host/FX8010.h (C ABI underneath).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "FX8010.h"

namespace {

struct Wav {
    int channels = 0;
    int sampleRate = 0;
    std::vector<float> frames;  // [frame][channel]
};

uint32_t rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

Wav readWav(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<unsigned char> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (d.size() < 12 || std::memcmp(d.data(), "RIFF", 4) != 0 || std::memcmp(d.data() + 8, "WAVE", 4) != 0) throw std::runtime_error(path + ": not a RIFF/WAVE file");
    Wav w;
    int format = 0, bits = 0;
    size_t pos = 12;
    bool haveFmt = false;
    while (pos + 8 <= d.size()) {
        const uint32_t len = rd32(&d[pos + 4]);
        const size_t body = pos + 8;
        if (body + len > d.size()) throw std::runtime_error(path + ": truncated chunk");
        if (std::memcmp(&d[pos], "fmt ", 4) == 0 && len >= 16) {
            format = rd16(&d[body]);
            w.channels = rd16(&d[body + 2]);
            w.sampleRate = (int)rd32(&d[body + 4]);
            bits = rd16(&d[body + 14]);
            if (format == 0xfffe && len >= 26) format = rd16(&d[body + 24]);  // WAVE_FORMAT_EXTENSIBLE: sub-format
            haveFmt = true;
        } else if (std::memcmp(&d[pos], "data", 4) == 0) {
            if (!haveFmt) throw std::runtime_error(path + ": data before fmt");
            if (w.channels < 1 || w.channels > 4) throw std::runtime_error(path + ": 1..4 channels supported");
            if (format == 1 && bits == 16) {
                const size_t n = len / 2;
                w.frames.resize(n);
                for (size_t i = 0; i < n; ++i) w.frames[i] = (float)(int16_t)rd16(&d[body + 2 * i]) / 32768.0f;
            } else if (format == 3 && bits == 32) {
                const size_t n = len / 4;
                w.frames.resize(n);
                std::memcpy(w.frames.data(), &d[body], n * 4);
            } else {
                throw std::runtime_error(path + ": only PCM 16-bit and IEEE float 32-bit are supported");
            }
            w.frames.resize(w.frames.size() / w.channels * w.channels);
            return w;
        }
        pos = body + len + (len & 1);
    }
    throw std::runtime_error(path + ": no data chunk");
}

void writeWavFloat(const std::string& path, const Wav& w) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write " + path);
    const uint32_t dataBytes = (uint32_t)(w.frames.size() * 4);
    auto u32 = [&](uint32_t v) { f.write(reinterpret_cast<const char*>(&v), 4); };
    auto u16 = [&](uint16_t v) { f.write(reinterpret_cast<const char*>(&v), 2); };
    f.write("RIFF", 4); u32(36 + dataBytes); f.write("WAVE", 4);
    f.write("fmt ", 4); u32(16); u16(3); u16((uint16_t)w.channels); u32((uint32_t)w.sampleRate);
    u32((uint32_t)(w.sampleRate * w.channels * 4)); u16((uint16_t)(w.channels * 4)); u16(32);
    f.write("data", 4); u32(dataBytes);
    f.write(reinterpret_cast<const char*>(w.frames.data()), dataBytes);
}

}  // namespace

int main(int argc, char** argv) {
    try {
        if (argc < 4) {
            std::cerr << "usage: fx8010_wav program.da in.wav out.wav [--instances N] [--block S] [--pick K] [--set name=value]... [--sweep name=lo:hi]...\n";
            return 2;
        }
        const std::string program = argv[1], inPath = argv[2], outPath = argv[3];
        int64_t instances = 1, pick = 0;
        int block = 4096;
        std::vector<std::pair<std::string, float>> sets;
        struct Sweep { std::string name; float lo, hi; };
        std::vector<Sweep> sweeps;
        for (int i = 4; i < argc; ++i) {
            const std::string a = argv[i];
            auto next = [&]() -> std::string { if (i + 1 >= argc) throw std::runtime_error(a + " needs a value"); return argv[++i]; };
            if (a == "--instances") instances = std::atoll(next().c_str());
            else if (a == "--block") block = std::atoi(next().c_str());
            else if (a == "--pick") pick = std::atoll(next().c_str());
            else if (a == "--set") {
                const std::string v = next();
                const size_t eq = v.find('=');
                if (eq == std::string::npos) throw std::runtime_error("--set name=value");
                sets.emplace_back(v.substr(0, eq), std::strtof(v.c_str() + eq + 1, nullptr));
            } else if (a == "--sweep") {
                const std::string v = next();
                const size_t eq = v.find('='), colon = v.find(':');
                if (eq == std::string::npos || colon == std::string::npos || colon < eq) throw std::runtime_error("--sweep name=lo:hi");
                sweeps.push_back({v.substr(0, eq), std::strtof(v.c_str() + eq + 1, nullptr), std::strtof(v.c_str() + colon + 1, nullptr)});
            } else throw std::runtime_error("unknown option " + a);
        }
        if (instances < 1 || block < 1 || pick < 0 || pick >= instances) throw std::runtime_error("bad --instances / --block / --pick");

        const Wav in = readWav(inPath);
        const int ch = in.channels;
        const size_t frames = in.frames.size() / ch;
        Klangraum::FX8010Batch dsp(instances, ch);
        if (!dsp.loadFile(program)) throw std::runtime_error("program failed to load: " + program);
        for (const auto& s : sets)
            if (dsp.setRegisterValue(s.first, s.second) != 0) throw std::runtime_error("no register " + s.first);
        for (const Sweep& s : sweeps) {
            std::vector<float> v((size_t)instances);
            for (int64_t i = 0; i < instances; ++i) v[(size_t)i] = instances > 1 ? s.lo + (s.hi - s.lo) * (float)i / (float)(instances - 1) : s.lo;
            if (dsp.setRegisterValues(s.name, v) != 0) throw std::runtime_error("no register " + s.name);
        }

        Wav out;
        out.channels = ch;
        out.sampleRate = in.sampleRate;
        out.frames.resize(frames * ch);
        std::vector<float> bin((size_t)block * ch * instances), bout(bin.size());
        for (size_t f0 = 0; f0 < frames; f0 += (size_t)block) {
            const int S = (int)std::min<size_t>((size_t)block, frames - f0);
            for (int s = 0; s < S; ++s)
                for (int c = 0; c < ch; ++c) {
                    const float v = in.frames[(f0 + s) * ch + c];
                    float* row = &bin[((size_t)s * ch + c) * instances];
                    for (int64_t i = 0; i < instances; ++i) row[i] = v;
                }
            dsp.process(bin.data(), bout.data(), S);
            for (int s = 0; s < S; ++s)
                for (int c = 0; c < ch; ++c) out.frames[(f0 + s) * ch + c] = bout[((size_t)s * ch + c) * instances + pick];
        }
        writeWavFloat(outPath, out);
        std::cout << frames << " frames x " << ch << " channel(s) x " << instances << " instance(s): " << dsp.getInstructionCounter()
                  << " emulated instructions, wrote instance " << pick << " to " << outPath << "\n";
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "fx8010_wav: " << e.what() << "\n";
        return 1;
    }
}
