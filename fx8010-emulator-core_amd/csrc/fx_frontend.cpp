// fx_frontend.cpp — clean-room loader for the reference's DANE-like ".da" dialect.
//
// What it must reproduce (reference: source/FX8010.cpp:365-875):
// the accept/reject set of the seven line patterns, the order in which registers get their
// indices, the error list (text + 1-based row) and the metadata/control lists.  The reference
// drives std::regex; this loader is a hand-written scanner with the same language, including
// the places where regex backtracking is observable (noted below).  It is exercised against
// the compiled reference on a 74-program corpus (tests/golden/parser_corpus.json) in tests/test_frontend.py.
#include "fx_model.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace fx {
namespace {

// character classes of ECMAScript \s \w \d in the "C" locale
inline bool isSpace(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
inline bool isDigit(char c) { return c >= '0' && c <= '9'; }
inline bool isWord(char c) { return isDigit(c) || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }
inline bool isOperandChar(char c) { return isWord(c) || c == '.' || c == '-'; }

// A cursor over one (already lower-cased, comment-stripped) line.
struct Scan {
    const std::string& s;
    size_t i = 0;
    explicit Scan(const std::string& str) : s(str) {}
    bool done() const { return i >= s.size(); }
    char peek() const { return s[i]; }
    size_t blanks() { size_t b = i; while (!done() && isSpace(s[i])) ++i; return i - b; }
    bool word(const char* w) {
        size_t n = std::strlen(w);
        if (s.compare(i, n, w) != 0) return false;
        i += n;
        return true;
    }
    template <class Pred> std::string run(Pred p) { size_t b = i; while (!done() && p(s[i])) ++i; return s.substr(b, i - b); }
    bool onlyBlanksLeft() const { size_t k = i; while (k < s.size() && isSpace(s[k])) ++k; return k == s.size(); }
};

struct Keyword { const char* text; int code; };
const Keyword kDeclWords[] = {{"static", R_STATIC}, {"temp", R_TEMP}, {"control", R_CONTROL},
                              {"input", R_INPUT}, {"output", R_OUTPUT}, {"const", R_CONST}};
const Keyword kOpWords[] = {{"macs", MACS}, {"macsn", MACSN}, {"macints", MACINTS}, {"macintw", MACINTW},
                            {"acc3", ACC3}, {"macmv", MACMV}, {"macw", MACW}, {"macwn", MACWN},
                            {"skip", SKIP}, {"andxor", ANDXOR}, {"tstneg", TSTNEG}, {"limit", LIMIT},
                            {"limitn", LIMITN}, {"log", LOG}, {"exp", EXP}, {"interp", INTERP},
                            {"idelay", IDELAY}, {"xdelay", XDELAY}};
const char* const kMetaWords[] = {"name", "copyright", "created", "engine", "comment", "guid"};

// the reference's error texts (source/FX8010.cpp:25-35)
const char kNoError[] = "Kein Fehler";
const char kRedeclared[] = "Mehrfache Variablendeklaration";
const char kUndeclared[] = "Variable nicht deklariert";
const char kInputAsResult[] = "Verwendung von Input fuer R ist nicht erlaubt";
const char kNoEnd[] = "Kein 'END' gefunden";
const char kBadSyntax[] = "Ungueltige Syntax";

// ^-?\d+(\.\d+)?$   (reference isNumber, source/helpers.cpp:21-27)
bool looksNumeric(const std::string& t) {
    size_t i = 0, n = t.size();
    if (i < n && t[i] == '-') ++i;
    size_t d = i;
    while (i < n && isDigit(t[i])) ++i;
    if (i == d) return false;
    if (i == n) return true;
    if (t[i++] != '.') return false;
    d = i;
    while (i < n && isDigit(t[i])) ++i;
    return i > d && i == n;
}

// Number after a declared name: optional separators [\s=,]* then \d+(\.\d+)? then blanks to end.
// Returns true if the remainder of the line is acceptable; `number` is empty when absent.
bool declRemainder(const std::string& s, size_t from, std::string& number) {
    size_t k = from;
    while (k < s.size() && (isSpace(s[k]) || s[k] == '=' || s[k] == ',')) ++k;
    if (k < s.size() && isDigit(s[k])) {
        size_t e = k;
        while (e < s.size() && isDigit(s[e])) ++e;
        if (e + 1 < s.size() && s[e] == '.' && isDigit(s[e + 1])) { e += 2; while (e < s.size() && isDigit(s[e])) ++e; }
        size_t t = e;
        while (t < s.size() && isSpace(s[t])) ++t;
        if (t == s.size()) { number = s.substr(k, e - k); return true; }
    }
    number.clear();
    size_t t = from;
    while (t < s.size() && isSpace(s[t])) ++t;
    return t == s.size();
}

}  // namespace

Program::Program(int channels) : numChannels(channels), loaderChannels(channels) {
    errors.push_back({kNoError, 1});          // source/FX8010.cpp:38-42
    regs.push_back({R_CCR, "ccr", 0.0f, 0});      // index 0, source/FX8010.cpp:50
    regs.push_back({R_READ, "read", 0.0f, 0});    // index 1, :53
    regs.push_back({R_WRITE, "write", 0.0f, 0});  // index 2, :54
    regs.push_back({R_AT, "at", 0.0f, 0});        // index 3, :55
}

int Program::findRegister(const std::string& name) const {
    for (size_t i = 0; i < regs.size(); ++i)
        if (regs[i].name == name) return (int)i;
    return -1;
}

void Program::addError(const std::string& what) { errors.push_back({what, lineNo_}); }

// reference mapRegisterToIndex (source/FX8010.cpp:745-774): an unknown numeric literal becomes a
// new STATIC register named by its own text ("0" and "0.0" are different registers).
int Program::resolveOperand(const std::string& token) {
    int idx = findRegister(token);
    if (idx >= 0) return idx;
    if (!looksNumeric(token)) return -1;
    float v = std::strtof(token.c_str(), nullptr);  // std::stof == strtof
    if (std::isinf(v)) sawUnparsable = true;        // the reference throws out_of_range here
    regs.push_back({R_STATIC, token, v, 0});
    return (int)regs.size() - 1;
}

// One line through the reference's pattern cascade (source/FX8010.cpp:395-739):
// declaration, blank, tramsize, instruction, metadata, end; anything else is a syntax error.
void Program::checkLine(const std::string& line) {
    Scan sc(line);
    sc.blanks();
    const size_t start = sc.i;

    // --- declaration: (static|temp|control|input|output|const) \s+ \w+ [number] ---
    for (const Keyword& kw : kDeclWords) {
        sc.i = start;
        if (!sc.word(kw.text) || sc.blanks() == 0) continue;
        const size_t nameAt = sc.i;
        const size_t nameMax = sc.run(isWord).size();
        // The name is \w+ and greedy, but the regex hands characters back when that makes
        // the rest match: "static a12.5" declares a1 with value 2.5.  Longest name first.
        std::string number;
        size_t len = nameMax;
        for (; len >= 1; --len)
            if (declRemainder(line, nameAt + len, number)) break;
        if (len == 0) break;  // no split works: not a declaration
        const std::string name = line.substr(nameAt, len);
        if (kw.code == R_CONTROL) controls.push_back(name);  // before the duplicate check (:408-411)
        if (findRegister(name) != -1) { addError(kRedeclared); return; }
        Gpr g;
        g.type = kw.code;
        g.name = name;
        if (!number.empty()) {
            if (kw.code == R_INPUT || kw.code == R_OUTPUT) {
                // the number is the channel; stoi semantics ("1.5" -> 1)
                long long ch = 0;
                for (char c : number) { if (!isDigit(c)) break; ch = ch * 10 + (c - '0'); if (ch > 2147483647LL) { ch = 2147483647LL; sawUnparsable = true; break; } }
                if (ch > loaderChannels - 1) {
                    addError("I/O Index ausserhalb des gueltigen Bereichs (max. " + std::to_string(numChannels) + ")");
                    return;
                }
                g.io = (int)ch;
            } else {
                g.value = std::strtof(number.c_str(), nullptr);
                if (std::isinf(g.value)) sawUnparsable = true;
            }
        }
        regs.push_back(g);
        return;
    }

    // --- blank ---
    if (start == line.size()) return;

    // --- (itramsize|xtramsize) \s+ (\d+)* \s  — exactly one blank must end the line ---
    for (int which = 0; which < 2; ++which) {
        sc.i = start;
        if (!sc.word(which == 0 ? "itramsize" : "xtramsize")) continue;
        const size_t gap = sc.blanks();
        if (gap == 0) break;
        const std::string digits = sc.run(isDigit);
        bool matched;
        if (!digits.empty()) matched = (sc.i + 1 == line.size() && isSpace(line[sc.i]));
        else matched = (sc.done() && gap >= 2);  // \s+ gives one blank back to the final \s
        if (!matched) break;
        if (digits.empty()) { sawUnparsable = true; addError(kBadSyntax); return; }  // reference: stoi("") throws
        long long v = 0;
        for (char c : digits) { v = v * 10 + (c - '0'); if (v > 2147483647LL) { v = 2147483647LL; sawUnparsable = true; break; } }
        int& size = which == 0 ? iTramSize : xTramSize;
        const int cap = which == 0 ? kMaxITram : kMaxXTram;
        // the reference tests the PREVIOUS size against the cap (source/FX8010.cpp:506,525)
        if (size > cap) {
            addError(which == 0 ? "iTRAM Size ausserhalb des gueltigen Bereichs (max. 8192)"
                                : "xRAM Size ausserhalb des gueltigen Bereichs (max. 1048576)");
            return;
        }
        size = (int)v;
        return;
    }

    // --- instruction: keyword \s+ R , A , X , Y ---
    for (const Keyword& kw : kOpWords) {
        sc.i = start;
        if (!sc.word(kw.text) || sc.blanks() == 0) continue;  // "macs" also prefixes "macsn": keep looking
        std::vector<std::string> ops;
        bool shape = true;
        for (;;) {
            sc.blanks();
            std::string tok;
            if ((options & kOptTramDane) && !sc.done() && sc.peek() == '&') { tok = "&"; ++sc.i; }  // opt-in: &name = position register of a delay tap
            tok += sc.run(isOperandChar);
            if (tok.empty()) { shape = false; break; }
            ops.push_back(tok);
            sc.blanks();
            if (sc.done()) break;
            if (sc.peek() != ',' || ops.size() == 4) { shape = false; break; }
            ++sc.i;
        }
        if (!shape || ops.size() != 4) break;
        Instr in;
        in.op = kw.code;
        int idx[4];
        // opt-in DANE taps (docs/TRAM Registermapping.pdf p.1): the position of "idelay read, rd, at, 17" lives in a register
        // of its own, "&rd" (created here with the literal as its value), which other instructions may write
        if ((options & kOptTramDane) && (kw.code == IDELAY || kw.code == XDELAY) && looksNumeric(ops[3]) && ops[1][0] != '&') {
            const std::string tap = "&" + ops[1];
            if (findRegister(tap) < 0) {
                float v = std::strtof(ops[3].c_str(), nullptr);
                if (std::isinf(v)) sawUnparsable = true;
                if (options & kOptTramAddrShift) v = v * 9.5367431640625e-07f;  // samples -> DANE address fraction: 0x800 / 2^31 per sample
                regs.push_back({R_STATIC, tap, v, 0});
            }
            ops[3] = tap;
        }
        for (int k = 0; k < 4; ++k) {  // R, A, X, Y in that order; stop at the first unknown name
            idx[k] = resolveOperand(ops[k]);
            if (idx[k] < 0) { addError(kUndeclared); return; }
            const Gpr& g = regs[idx[k]];
            if (k == 0) {
                if (g.type == R_INPUT) { addError(kInputAsResult); return; }
                if (g.type == R_OUTPUT) in.hasOutput = true;
            } else if (g.type == R_INPUT) in.hasInput = true;
            else if (g.name == "noise") in.hasNoise = true;
        }
        in.r = idx[0]; in.a = idx[1]; in.x = idx[2]; in.y = idx[3];
        instrs.push_back(in);
        return;
    }

    // --- metadata: key \s+ "value" and nothing after the closing quote ---
    for (const char* key : kMetaWords) {
        sc.i = start;
        if (!sc.word(key) || sc.blanks() == 0 || sc.done() || sc.peek() != '"') continue;
        const size_t open = sc.i;
        const size_t close = line.find('"', open + 1);
        if (close == std::string::npos || close == open + 1 || close + 1 != line.size()) continue;
        const std::string value = line.substr(open + 1, close - open - 1);
        bool replaced = false;
        for (auto& kv : meta)
            if (kv.first == key) { kv.second = value; replaced = true; }
        if (!replaced) meta.emplace_back(key, value);
        return;
    }

    // --- end ---
    sc.i = start;
    if (sc.word("end") && sc.onlyBlanksLeft()) {
        Instr in;
        in.op = END;  // all operands 0 (source/FX8010.cpp:716)
        instrs.push_back(in);
        return;
    }
    addError(kBadSyntax);
}

bool Program::finishLoad(const std::vector<std::string>& lines) {
    for (const std::string& l : lines) {
        checkLine(l);
        ++lineNo_;
    }
    // the last physical line must be exactly "end" (source/FX8010.cpp:829-838)
    if (lines.empty() || lines.back() != "end") addError(kNoEnd);
    if (errors.size() > 1) return false;
    ready = true;
    return true;
}

namespace {
// pre-pass of the reference loader (source/FX8010.cpp:790-812): cut at the first ';', lower-case.
std::vector<std::string> splitAndClean(std::istream& is) {
    std::vector<std::string> lines;
    std::string line;
    while (std::getline(is, line)) {
        size_t semi = line.find(';');
        if (semi != std::string::npos) line.resize(semi);
        for (char& c : line)
            if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
        lines.push_back(line);
    }
    return lines;
}
}  // namespace

bool Program::loadFile(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;  // open failure: false, no error entry (source/FX8010.cpp:868-873)
    return finishLoad(splitAndClean(f));
}

bool Program::loadText(const std::string& text) {
    std::istringstream is(text);
    return finishLoad(splitAndClean(is));
}

Luts::Luts() {
    const int n = 32;
    for (int e = 0; e < 32; ++e) {
        double rootTab[32], powTab[32];
        const double step = (1.0 - 0.0) / (n - 1);
        const float ef = static_cast<float>(e);
        for (int i = 0; i < n; ++i) {
            const double x = 0.0 + i * step;
            rootTab[i] = std::pow(x, 1.0 / ef);  // LOG: e-th root, source/FX8010.cpp:142
            powTab[i] = std::pow(x, ef);          // EXP: e-th power, :159
        }
        for (int k = 0; k < n; ++k) {
            // negative half = mirrored and negated; the reference's negation loop iterates
            // with a float variable, so these 32 entries pass through float (:190-199)
            log_[e][k] = -static_cast<double>(static_cast<float>(rootTab[n - 1 - k]));
            exp_[e][k] = -static_cast<double>(static_cast<float>(powTab[n - 1 - k]));
            log_[e][n + k] = rootTab[k];
            exp_[e][n + k] = powTab[k];
        }
        log_[e][64] = log_[e][63];
        exp_[e][64] = exp_[e][63];
    }
}

namespace {
// smallest float x whose t = (double)x - -1.0 reaches the bound
float firstFloatReaching(double bound) {
    float x = (float)(bound - 1.0);
    while ((double)x - -1.0 >= bound) x = std::nextafterf(x, -4.0f);
    while ((double)x - -1.0 < bound) x = std::nextafterf(x, 4.0f);
    return x;
}
}  // namespace

void lutDomainBounds(float out[2]) {
    const double step = (1.0 - -1.0) / 63.0;
    out[0] = firstFloatReaching(-step);
    out[1] = firstFloatReaching(64.0 * step);
}

// The generated code's cheap test of a guessed segment g: with d = (double)x - x1[g], "0 <= d < W" must imply that the
// reference's index for x is g.  d is monotone in x, so it is enough that the smallest float at or above x1[g] has
// reached threshold g, and that d at threshold g+1 is at least W.  Returns the high word of the largest such W with a
// zero low word (the test is one unsigned compare of d's high word: sign bit and NaN read as "large"), 0 if the grid
// does not allow the test.  Only the grid enters (step, x1, thresholds), not the table contents.
static uint32_t computeGuessWindowHi();
uint32_t lutGuessWindowHi() {
    static const uint32_t hi = computeGuessWindowHi();
    return hi;
}
static uint32_t computeGuessWindowHi() {
    const double step = (1.0 - -1.0) / 63.0;
    auto indexOf = [&](double t) { return (int)(t / step); };
    double w = HUGE_VAL;
    float next = 0.0f;
    for (int k = 0; k <= 63; ++k) {
        const double x1 = -1.0 + k * step;
        float lo = (float)x1;
        if ((double)lo < x1) lo = std::nextafterf(lo, 4.0f);
        if (indexOf((double)lo - -1.0) < k) return 0;          // a float in [x1[k], threshold k) exists
        if (k > 0) {
            // next = threshold k as a float, first x of segment k: d of segment k-1 there bounds its window
            double t = k * step;
            while (indexOf(t) >= k) t = std::nextafter(t, 0.0);
            while (indexOf(t) < k) t = std::nextafter(t, 4.0);
            next = firstFloatReaching(t);
            w = std::min(w, (double)next - (-1.0 + (k - 1) * step));
        }
    }
    uint64_t bits;
    std::memcpy(&bits, &w, 8);
    return (uint32_t)(bits >> 32);
}

LutDevice::LutDevice(const Luts& l) : blob(kLutBlobDoubles, 0.0), invStep(63.0 / 2.0) {
    const double step = (1.0 - -1.0) / 63.0;
    auto indexOf = [&](double t) { return (int)(t / step); };  // t in [0, 2]: in range, plain truncation
    blob[kLutThrOff + 0] = 0.0;
    for (int k = 1; k <= 63; ++k) {
        double t = k * step;
        // walk to the exact boundary: the smallest t whose quotient truncates to >= k
        while (indexOf(t) >= k) t = std::nextafter(t, 0.0);
        while (indexOf(t) < k) t = std::nextafter(t, 4.0);
        blob[kLutThrOff + k] = t;
    }
    blob[kLutThrOff + 64] = HUGE_VAL;
    for (int k = 0; k < 64; ++k) blob[kLutX1Off + k] = -1.0 + k * step;
    for (int tsel = 0; tsel < 64; ++tsel) {
        const double* tbl = tsel < 32 ? l.log_[tsel] : l.exp_[tsel - 32];
        for (int k = 0; k < 64; ++k) {
            const double x1 = -1.0 + k * step;
            const double x2 = -1.0 + (k + 1) * step;
            const double y1 = tbl[k];
            const double y2 = tbl[k + 1];
            blob[kLutSegOff + (tsel * 64 + k) * 2 + 0] = (y2 - y1) / (x2 - x1);
            blob[kLutSegOff + (tsel * 64 + k) * 2 + 1] = y1;
        }
    }
    // fp32 thresholds
    float* xthr = reinterpret_cast<float*>(&blob[kLutXthrOff]);
    xthr[0] = -HUGE_VALF;
    for (int k = 1; k <= 63; ++k) xthr[k] = firstFloatReaching(blob[kLutThrOff + k]);
    xthr[64] = xthr[65] = HUGE_VALF;
    float* xdom = reinterpret_cast<float*>(&blob[kLutXdomOff]);
    lutDomainBounds(xdom);
}

}  // namespace fx
