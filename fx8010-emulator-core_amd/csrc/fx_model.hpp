// fx_model.hpp — host-side program model and front-end of the MI355X batch FX8010 interpreter.
//
// The model mirrors what the reference's loader leaves behind for its hot loop
// (reference: include/FX8010.h:79-100 Opcode, :127-142 RegisterType, :167-174 GPR,
// :180-191 Instruction).  Register INDEX ORDER is part of the contract: instructions
// address registers by index, and indices are handed out in order of first appearance
// (4 specials, then declarations and numeric literals as encountered).
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace fx {

enum Opcode : int {
    MACS = 0x0, MACSN = 0x1, MACW = 0x2, MACWN = 0x3, MACINTS = 0x4, MACINTW = 0x5, ACC3 = 0x6,
    MACMV = 0x7, ANDXOR = 0x8, TSTNEG = 0x9, LIMIT = 0xa, LIMITN = 0xb, LOG = 0xc, EXP = 0xd,
    INTERP = 0xe, SKIP = 0xf, IDELAY = 0x10, XDELAY = 0x11, END = 0x12
};

enum RegType : int {
    R_STATIC = 0, R_TEMP, R_CONTROL, R_INPUT, R_OUTPUT, R_CONST, R_ITRAMSIZE, R_XTRAMSIZE,
    R_READ, R_WRITE, R_AT, R_CCR
};

// options (same values as FX_OPT_* in include/fx8010_amd.h and FXO_OPT_* in the oracle)
constexpr unsigned kOptTramDane = 1u << 0;       // DANE delay-line model: per-sample address counter, ring taps, &name tap registers
constexpr unsigned kOptTramAddrShift = 1u << 1;  // tap positions are DANE addresses (0x800 per sample)
constexpr unsigned kOptTramInterp = 1u << 2;     // (with the address shift) READ taps interpolate linearly with the address's low 11 bits
constexpr unsigned kOptAll = kOptTramDane | kOptTramAddrShift | kOptTramInterp;

constexpr int kMaxITram = 8192;     // reference MAX_IDELAY_SIZE, include/FX8010.h:41
constexpr int kMaxXTram = 1048576;  // reference MAX_XDELAY_SIZE, include/FX8010.h:42

struct Gpr {
    int type = R_STATIC;
    std::string name;
    float value = 0.0f;  // initial / host-side value
    int io = 0;          // channel for INPUT/OUTPUT registers
};

struct Instr {
    int op = 0, r = 0, a = 0, x = 0, y = 0;
    bool hasInput = false, hasOutput = false, hasNoise = false;
};

struct LoadError {
    std::string description;
    int row = 1;
};

// Front-end state of one program.  loadFile()/loadText() may be called more than once on
// the same object; like the reference (which never clears its vectors) state accumulates.
class Program {
public:
    explicit Program(int numChannels);

    bool loadFile(const std::string& path);  // reference loadFile, source/FX8010.cpp:777-875
    bool loadText(const std::string& text);  // same pipeline from memory

    int findRegister(const std::string& name) const;  // -1 if absent

    int numChannels;     // channels of the object as constructed: PCM layout, output latches, error texts
    int loaderChannels;  // what setChannels() changes: only the loader's I/O-index bound (reference FX8010.h:73, FX8010.cpp:447)
    std::vector<Gpr> regs;
    std::vector<Instr> instrs;
    std::vector<LoadError> errors;  // [0] is always {"Kein Fehler", 1}
    std::vector<std::string> controls;
    std::vector<std::pair<std::string, std::string>> meta;
    int iTramSize = 0, xTramSize = 0;
    // behaviour beyond the reference, off by default (FX_OPT_* of include/fx8010_amd.h): set before loading
    unsigned options = 0;
    bool ready = false;
    bool sawUnparsable = false;  // input on which the reference itself throws (stoi/stof)

private:
    bool finishLoad(const std::vector<std::string>& lines);
    void checkLine(const std::string& line);
    int resolveOperand(const std::string& token);
    void addError(const std::string& what);
    int lineNo_ = 1;  // reference errorCounter, include/FX8010.h:273
};

// LOG/EXP tables exactly as the reference builds them at construction
// (source/FX8010.cpp:63-105,129-199): 32 exponents x 64 doubles each, host libm pow.
// Entry [64] duplicates [63] so that the x == 1.0 read of [idx+1] stays finite.
struct Luts {
    double log_[32][65];
    double exp_[32][65];
    Luts();
};

// Device form of the LOG/EXP evaluation.  The reference computes, per call
// (linearInterpolate, source/FX8010.cpp:283-296):
//     idx = (int)((x - -1.0) / step);  x1 = -1.0 + idx*step;  x2 = -1.0 + (idx+1)*step;
//     y   = (y2 - y1) / (x2 - x1) * (x - x1) + y1
// Everything except the last multiply-add depends only on (table, idx), and idx is a monotone
// step function of t = x + 1.0.  So the two fp64 divisions are done here, once, with the very
// same IEEE operations, and the kernel only compares against thresholds and does one mul + add:
//   thr[k]   k = 0..64 : smallest double t with (int)(t / step) >= k   (thr[0] = 0, thr[64] = +inf)
//   x1[k]    k = 0..63 : -1.0 + k*step
//   seg[table][k]      : { (y2 - y1) / (x2 - x1), y1 }   table 0..31 LOG, 32..63 EXP
// The translated programs (fx_xlate.cpp) find idx from the fp32 operand itself: t = (double)x + 1.0 is monotone in
// x, so there are fp32 thresholds with the same meaning,
//   xthr[k]  k = 0..65 : smallest float x with (double)x + 1.0 >= thr[k]   (xthr[0] = -inf, xthr[64..65] = +inf)
//   xdom[2]            : smallest floats x with t >= -step and with t >= 64*step (outside: index out of range, flagged)
// Blob layout (doubles): thr[65] | pad to 66 | x1[64] | seg[64][64][2] | xthr[66] as floats | xdom[2] as floats
constexpr int kLutThrOff = 0;
constexpr int kLutX1Off = 66;
constexpr int kLutSegOff = 66 + 64;
constexpr int kLutXthrOff = kLutSegOff + 64 * 64 * 2;  // 66 floats = 33 doubles
constexpr int kLutXdomOff = kLutXthrOff + 33;          // 2 floats = 1 double
constexpr int kLutBlobDoubles = kLutXdomOff + 1;
// xdom[2] of the blob: the fp32 operand range [xdom[0], xdom[1]) inside which the segment index is 0..63
void lutDomainBounds(float out[2]);
uint32_t lutGuessWindowHi();  // see fx_frontend.cpp
struct LutDevice {
    std::vector<double> blob;
    double invStep;  // 63/2: only used to guess idx, the thresholds decide
    explicit LutDevice(const Luts& l);
};

}  // namespace fx
