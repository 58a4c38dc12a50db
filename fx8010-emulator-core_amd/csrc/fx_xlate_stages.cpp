// fx_xlate_stages.cpp — a program pipelined over the wavefronts of a workgroup (fx_xlate.hpp StageInfo): which rows a record
// touches, where cuts are legal and which rows cross them (planStages), the proof that a plan computes what the program
// computes (verifyStagePlan), where packets, flags and tables lie in LDS (stageLdsLayout), and the staged code object
// (buildStagedImage: every stage translated as a program of its own by planXlate).
#include <algorithm>
#include <cstdio>
#include <cstring>

#include "fx_knobs.hpp"
#include "fx_xlate_emit.hpp"
#include "fx_xlate_internal.hpp"

namespace fx {
using namespace xl;

// ---- stage pipelining (fx_xlate.hpp StageInfo) ---------------------------------------------------------------------
// register-file rows a record reads and writes (uniform operands are not rows; the X word of LOG / EXP is a table)
Access xl::accessOf(const MicroOp& r) {
    Access a;
    const uint32_t slot = r.w[0];
    if (slot == AS_ENDSAMPLE || slot == AS_NOP || slot == AS_PRED || slot == AS_UNPRED) return a;
    const bool hot = slot >= AS_MACS && slot < (uint32_t)kAsmSlots;
    const uint32_t kind = hot ? ((slot - AS_MACS) % 16) / 2 : (r.w[6] & 7u);
    auto read = [&](uint32_t word, bool uniform) { if (!uniform) a.reads[a.nReads++] = word; };
    if (slot == AS_SKIP) {
        a.reads[a.nReads++] = 0;  // the CCR row
        read(r.w[3], kind & 2u);
        read(r.w[4], kind & 4u);
        return a;
    }
    if (slot == AS_NOISE) { a.noise = true; a.write = (int)r.w[5]; return a; }
    if (slot >= AS_TRAM_IR && slot <= AS_TRAM_XW) {
        a.tram = true;
        read(r.w[4], kind & 4u);
        if (slot == AS_TRAM_IR || slot == AS_TRAM_XR) a.write = (int)r.w[5];
        else read(r.w[2], kind & 1u);
        return a;
    }
    if (hot && kind == 7u) { a.write = (int)r.w[5]; a.ccr = (slot - AS_MACS) & 1u; return a; }  // folded on the host
    read(r.w[2], kind & 1u);
    if (slot == AS_LUT) read(r.w[3], kind & 2u);   // (the table number: a row when it is a per-instance value)
    else if (slot != AS_MOV) {
        read(r.w[3], kind & 2u);
        read(r.w[4], kind & 4u);
    }
    a.write = (int)r.w[5];
    a.ccr = hot ? ((slot - AS_MACS) & 1u) != 0 : ((r.w[6] >> 3) & 1u) != 0;
    return a;
}

namespace {
// a rough count of the vector instructions a record costs (for balancing the stages only)
int costOf(const MicroOp& r) {
    const uint32_t slot = r.w[0];
    if (slot == AS_ENDSAMPLE || slot == AS_NOP || slot == AS_UNPRED) return 0;
    if (slot == AS_PRED) return 6;
    if (slot == AS_SKIP) return 3;
    if (slot == AS_MOV) return 1;
    if (slot == AS_LUT) return 16;
    if (slot == AS_NOISE || slot == AS_LIMIT || slot == AS_LIMITN) return 4;
    if (slot >= AS_TRAM_IR && slot <= AS_TRAM_XW) return 2;
    if (slot < AS_MACS) return 12;  // wrap-around family, logic, TSTNEG
    const uint32_t rel = slot - AS_MACS, family = rel / 16;
    return (family == 3 ? 7 : 3) + ((rel & 1u) ? 12 : 0);
}
}  // namespace

namespace {
bool verifyStagePlan(const std::vector<MicroOp>& steady, const std::vector<MicroOp>& last, size_t n, const XlateProgram& prog, int rows, const StagePlan& P, std::string* why);
}

StagePlan planStages(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords, const XlateProgram& prog, int nRows, int wanted) {
    StagePlan P;
    auto no = [&](const std::string& why) { P.why = why; P.cuts.clear(); P.live.clear(); return P; };
    if (wanted < 2) return no("one stage asked for");
    if (!prog.trackRows.empty()) return no("control tracks");
    if (prog.tramDane) return no("DANE delay-line model");
    if (steadyRecords.size() != lastRecords.size()) return no("streams of different length");
    size_t n = 0;
    while (n < steadyRecords.size() && steadyRecords[n].w[0] != AS_ENDSAMPLE) ++n;
    if (n < 4 || n >= steadyRecords.size() || lastRecords[n].w[0] != AS_ENDSAMPLE) return no("too short");
    auto shape = [](const MicroOp& r) { return r.w[0] >= AS_MACS && r.w[0] < (uint32_t)kAsmSlots ? (r.w[0] & ~1u) : r.w[0]; };  // (hot slots: without the CCR bit)
    for (size_t i = 0; i < n; ++i)
        if (shape(steadyRecords[i]) != shape(lastRecords[i]) || steadyRecords[i].w[5] != lastRecords[i].w[5]) return no("streams differ in structure");
    const int rows = std::max(nRows, 1);
    // boundary b (1 .. n-1) = a cut between record b-1 and record b
    std::vector<uint8_t> allowed(n + 1, 1);
    allowed[0] = 0;
    allowed[n] = 0;
    std::vector<std::vector<int>> liveAt(n + 1);
    auto forbid = [&](size_t lo, size_t hi) {  // no cut b with lo < b <= hi
        for (size_t b = lo + 1; b <= hi && b <= n; ++b) allowed[b] = 0;
    };
    auto addLive = [&](size_t lo, size_t hi, int row) {  // row is live at every cut b with lo < b <= hi
        for (size_t b = lo + 1; b <= hi && b <= n; ++b)
            if (std::find(liveAt[b].begin(), liveAt[b].end(), row) == liveAt[b].end()) liveAt[b].push_back(row);
    };
    // structure: a SKIP, the instruction in front of it (fused predicate) and its shadow up to the UNPRED stay together;
    // delay-line and noise instructions belong to stage 0
    {
        bool open = false;
        size_t from = 0, lastOwned = 0;
        bool anyOwned = false;
        for (size_t i = 0; i < n; ++i) {
            const uint32_t slot = steadyRecords[i].w[0];
            if (slot == AS_SKIP && !open) { open = true; from = i > 0 ? i - 1 : 0; }
            if (slot == AS_PRED && !open) { open = true; from = i > 0 ? i - 1 : 0; }
            if (slot == AS_UNPRED && open) { forbid(from, i); open = false; }
            const Access a = accessOf(steadyRecords[i]);
            if (a.tram || a.noise) { lastOwned = i; anyOwned = true; }
        }
        if (open) forbid(from, n);
        if (anyOwned) forbid(0, lastOwned + 1 > n ? n : lastOwned + 1);
    }
    std::vector<uint8_t> isInput((size_t)rows, 0);
    for (int r : prog.inRows)
        if (r >= 0 && r < rows) isInput[(size_t)r] = 1;
    for (const std::vector<MicroOp>* recs : {&steadyRecords, &lastRecords}) {
        // per row: its writes (position, conditional?) in program order
        std::vector<std::vector<std::pair<size_t, bool>>> writes((size_t)rows);
        bool shadow = false;
        std::vector<uint8_t> shadowed(n, 0);
        for (size_t i = 0; i < n; ++i) {
            const uint32_t slot = (*recs)[i].w[0];
            if (slot == AS_PRED) shadow = true;
            else if (slot == AS_UNPRED) shadow = false;
            shadowed[i] = shadow;
            const Access a = accessOf((*recs)[i]);
            if (a.write >= 0 && a.write < rows) writes[(size_t)a.write].emplace_back(i, shadow);
            if (a.ccr) writes[0].emplace_back(i, shadow);
        }
        for (size_t i = 0; i < n; ++i) {
            const Access a = accessOf((*recs)[i]);
            for (int k = 0; k < a.nReads; ++k) {
                const uint32_t R = a.reads[k];
                if (R >= (uint32_t)rows) return no("operand row out of range");
                if (isInput[R]) { addLive(0, i, (int)R); continue; }  // stage 0 loads the PCM input; it travels with the packets
                const auto& W = writes[R];
                // walk back from the read: conditional writes, down to the nearest unconditional one
                size_t lowest = i;  // the earliest position the value can come from (same sample)
                bool found = false;
                for (size_t q = W.size(); q-- > 0;) {
                    if (W[q].first >= i) continue;
                    lowest = W[q].first;
                    if (!W[q].second) { found = true; break; }
                }
                if (found) { addLive(lowest, i, (int)R); continue; }
                // ... the value (also) comes from the previous sample: every candidate definition - the conditional ones of
                // this sample, and the previous sample's from the end of the program back to its last unconditional one -
                // must be in the reader's stage
                size_t highest = i;
                for (size_t q = W.size(); q-- > 0;) {
                    if (W[q].first < i) break;
                    highest = std::max(highest, W[q].first);
                    if (!W[q].second) break;
                }
                if (!W.empty()) {
                    size_t top = i;
                    for (const auto& w : W)
                        if (w.first >= i) top = std::max(top, w.first);
                    highest = top;  // (conservative: up to the last write of the row)
                }
                forbid(lowest, i);
                forbid(i, highest);
            }
        }
        // the stage of a row's LAST write stores it (state rows at the end of a block, PCM latch rows every sample): if that
        // write is conditional the stage needs the value it replaces - from the nearest unconditional write in front of it, or,
        // when there is none in the sample, from the previous sample: then all writes of the row stay in one stage
        for (int R = 0; R < rows; ++R) {
            const auto& W = writes[(size_t)R];
            if (W.size() < 2 || !W.back().second) continue;
            size_t q = W.size() - 1;
            while (q > 0 && W[q].second) --q;
            if (W[q].second) forbid(W.front().first, W.back().first);
            else addLive(W[q].first, W.back().first, R);
        }
    }
    // balance: cumulative cost, cuts at allowed boundaries nearest to the ideal positions
    std::vector<int> cum(n + 1, 0);
    for (size_t i = 0; i < n; ++i) cum[i + 1] = cum[i] + costOf(steadyRecords[i]);
    const int total = cum[n];
    if (total < 16 * wanted) wanted = std::max(1, total / 16);
    if (wanted < 2) return no("too little work per stage");
    // The slowest stage sets the pace of the whole workgroup (everybody meets at the step's barrier), and a stage's time is its
    // share of the program PLUS what the pipeline costs it: stage 0 fetches the PCM input (ring of bursts, index mode), the last
    // writer of an output latch stores PCM, every row received costs a move and a request, every row sent a write.  Cuts = the allowed boundaries that minimise the largest such sum (dynamic programme over boundaries; a
    // stage with less than a quarter of an even share of the program is not worth a barrier: fewer stages then).
    bool anyInputRow = false;
    for (int r : prog.inRows) anyInputRow = anyInputRow || r >= 0;
    static const bool flat = knobInt(FX_DIAG_KNOB("FX_STAGES_BALANCE"), 1) == 0;   // diagnostics: the program's share only
    const int kInputCost = anyInputRow && !flat ? 14 : 0, kOutputCost = flat ? 0 : 7, kRecvCost = flat ? 0 : 3, kSendCost = flat ? 0 : 1, kFixedCost = 8;
    std::vector<size_t> bounds{0};
    for (size_t b = 1; b < n; ++b)
        if (allowed[b]) bounds.push_back(b);
    bounds.push_back(n);
    const size_t nb = bounds.size();
    auto stageCost = [&](size_t lo, size_t hi) {   // indices into bounds
        const size_t from = bounds[lo], to = bounds[hi];
        int c = cum[to] - cum[from] + kFixedCost;
        if (from == 0) c += kInputCost; else c += kRecvCost * (int)liveAt[from].size();
        if (to == n) c += kOutputCost; else c += kSendCost * (int)liveAt[to].size();
        return c;
    };
    std::vector<int> cuts;
    for (int K = std::min<int>(wanted, (int)nb - 1); K >= 2 && cuts.empty(); --K) {
        const int kInf = 1 << 30;
        // best[k][j]: the smallest possible largest-stage cost of records [0, bounds[j]) in k stages
        std::vector<std::vector<int>> best((size_t)K + 1, std::vector<int>(nb, kInf)), from((size_t)K + 1, std::vector<int>(nb, -1));
        best[0][0] = 0;
        for (int k = 1; k <= K; ++k)
            for (size_t j = 1; j < nb; ++j)
                for (size_t i = 0; i < j; ++i) {
                    if (best[(size_t)k - 1][i] == kInf) continue;
                    if ((cum[bounds[j]] - cum[bounds[i]]) * 4 * wanted < total) continue;
                    const int c = std::max(best[(size_t)k - 1][i], stageCost(i, j));
                    if (c < best[(size_t)k][j]) { best[(size_t)k][j] = c; from[(size_t)k][j] = (int)i; }
                }
        if (best[(size_t)K][nb - 1] == kInf) continue;
        size_t j = nb - 1;
        for (int k = K; k >= 1; --k) {
            j = (size_t)from[(size_t)k][j];
            if (k > 1) cuts.push_back((int)bounds[j]);
        }
        std::reverse(cuts.begin(), cuts.end());
    }
    P.totalCost = total;
    for (size_t i = 0; i < n; ++i) P.totalLuts += steadyRecords[i].w[0] == AS_LUT ? 1 : 0;
    if (FX_DIAG_KNOB("FX_STAGES_DEBUG")) {
        std::string line;
        for (size_t b = 0; b <= n; ++b) line += allowed[b] ? '+' : '.';
        std::fprintf(stderr, "planStages: %zu records, total cost %d, boundaries %s\n", n, total, line.c_str());
    }
    P.totalCost = total;
    if (cuts.empty()) return no("no legal cut");
    P.cuts = cuts;
    {   // what the plan expects each stage to cost (the units of costOf: roughly vector instructions), pipeline overhead included
        size_t lo = 0;
        std::vector<size_t> edges{0};
        for (int b : cuts) edges.push_back((size_t)b);
        edges.push_back(n);
        for (size_t k = 0; k + 1 < edges.size(); ++k) {
            int c = cum[edges[k + 1]] - cum[edges[k]] + kFixedCost;
            if (edges[k] == 0) c += kInputCost; else c += kRecvCost * (int)liveAt[edges[k]].size();
            if (edges[k + 1] == n) c += kOutputCost; else c += kSendCost * (int)liveAt[edges[k + 1]].size();
            int luts = 0;
            for (size_t i = edges[k]; i < edges[k + 1]; ++i) luts += steadyRecords[i].w[0] == AS_LUT ? 1 : 0;
            P.stageCost.push_back(c);
            P.stageLuts.push_back(luts);
        }
        (void)lo;
    }
    if (FX_DIAG_KNOB("FX_STAGES_DEBUG")) {
        std::string line;
        for (size_t k = 0; k < P.stageCost.size(); ++k) line += " " + std::to_string(P.stageCost[k]) + "(" + std::to_string(P.stageLuts[k]) + ")";
        std::fprintf(stderr, "planStages: wanted %d -> %zu stages, cost per stage (LOG/EXP):%s\n", wanted, P.stageCost.size(), line.c_str());
    }
    for (int b : cuts) {
        std::vector<int> l = liveAt[(size_t)b];
        std::sort(l.begin(), l.end());
        P.live.push_back(l);
    }
    const int K = (int)cuts.size() + 1;
    auto stageOf = [&](size_t pos) { int st = 0; for (int b : cuts) if ((int)pos >= b) ++st; return st; };
    // who stores what at the end of a block: the stage of the row's last write (the last stream decides: it makes every CCR
    // write live); rows no record writes - PCM input rows, untouched state - stay with stage 0, which loads every input channel
    P.storeStage.assign((size_t)rows, 0);
    for (size_t i = 0; i < n; ++i) {
        const Access a = accessOf(lastRecords[i]);
        if (a.write >= 0 && a.write < rows) P.storeStage[(size_t)a.write] = stageOf(i);
        if (a.ccr) P.storeStage[0] = stageOf(i);
    }
    P.pcmStage.assign(prog.latchRows.size(), 0);
    for (size_t c = 0; c < prog.latchRows.size(); ++c)
        if (prog.latchRows[c] >= 0 && prog.latchRows[c] < rows) P.pcmStage[c] = P.storeStage[(size_t)prog.latchRows[c]];
    P.inMask.assign((size_t)K, 0);
    for (size_t c = 0; c < prog.inRows.size(); ++c)
        if (prog.inRows[c] >= 0) P.inMask[0] |= 1u << c;
    {
        std::string why;
        if (!verifyStagePlan(steadyRecords, lastRecords, n, prog, rows, P, &why)) return no(why);
    }
    return P;
}

namespace {
// A plan is checked before it is used, by running the program's DATA FLOW twice over a few samples and two launches - as the
// reference runs it (one register file, records in order) and as the stages would (one file per stage, the rows of `live`
// copied at each cut, inputs loaded by stage 0, rows stored at the end of a launch by their owners) - on symbolic values: every
// write makes a value that is a hash of the record, the sample and the values it read (a conditional write: also of the value
// it may leave in place).  Every read, every PCM output and every stored row must see the same value in both runs.
uint64_t mix64(uint64_t h, uint64_t v) {
    h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdull;
    return h ^ (h >> 33);
}
bool verifyStagePlan(const std::vector<MicroOp>& steady, const std::vector<MicroOp>& last, size_t n, const XlateProgram& prog, int rows, const StagePlan& P, std::string* why) {
    const int K = (int)P.cuts.size() + 1;
    const std::vector<MicroOp>* current = &steady;   // (the last sample of a launch runs the last-sample stream: every CCR write live)
    auto stageOf = [&](size_t pos) { int st = 0; for (int b : P.cuts) if ((int)pos >= b) ++st; return st; };
    std::vector<uint64_t> state((size_t)rows);
    for (int r = 0; r < rows; ++r) state[(size_t)r] = mix64(0x1234, (uint64_t)r);
    std::vector<uint8_t> shadowed(n, 0);
    {
        bool sh = false;
        for (size_t i = 0; i < n; ++i) {
            if (steady[i].w[0] == AS_PRED) sh = true;
            else if (steady[i].w[0] == AS_UNPRED) sh = false;
            shadowed[i] = sh;
        }
    }
    auto step = [&](std::vector<uint64_t>& file, size_t i, int launch, int t, std::vector<uint64_t>* trace) {
        const Access a = accessOf((*current)[i]);
        uint64_t h = mix64(mix64((uint64_t)i * 977 + 13, (uint64_t)t), (uint64_t)launch);
        for (int k = 0; k < a.nReads; ++k) {
            const uint64_t v = file[a.reads[k]];
            if (trace) trace->push_back(v);
            h = mix64(h, v);
        }
        auto write = [&](int R) {
            uint64_t v = mix64(h, (uint64_t)R);
            if (shadowed[i]) v = mix64(v, file[(size_t)R]);  // (may leave the old value in place)
            file[(size_t)R] = v;
        };
        if (a.write >= 0 && a.write < rows) write(a.write);
        if (a.ccr) write(0);
    };
    for (int launch = 0; launch < 2; ++launch) {
        std::vector<uint64_t> seq = state;
        std::vector<std::vector<uint64_t>> file((size_t)K, state);
        for (int t = 0; t < 3; ++t) {
            current = t == 2 ? &last : &steady;
            std::vector<uint64_t> want, got;
            for (size_t c = 0; c < prog.inRows.size(); ++c)
                if (prog.inRows[c] >= 0 && prog.inRows[c] < rows) {
                    seq[(size_t)prog.inRows[c]] = mix64(mix64(0x77, c), (uint64_t)(launch * 16 + t));
                    file[0][(size_t)prog.inRows[c]] = seq[(size_t)prog.inRows[c]];
                }
            for (size_t i = 0; i < n; ++i) step(seq, i, launch, t, &want);
            for (int k = 0; k < K; ++k) {
                if (k > 0)
                    for (int R : P.live[(size_t)k - 1]) file[(size_t)k][(size_t)R] = file[(size_t)k - 1][(size_t)R];
                const size_t from = k == 0 ? 0 : (size_t)P.cuts[(size_t)k - 1], to = k + 1 == K ? n : (size_t)P.cuts[(size_t)k];
                for (size_t i = from; i < to; ++i) step(file[(size_t)k], i, launch, t, &got);
            }
            if (want != got) { if (why) *why = "plan check: a read would see another value"; return false; }
            for (size_t c = 0; c < prog.latchRows.size(); ++c) {
                const int R = prog.latchRows[c];
                if (R < 0 || R >= rows) continue;
                if (file[(size_t)P.pcmStage[c]][(size_t)R] != seq[(size_t)R]) { if (why) *why = "plan check: PCM output of another stage's copy"; return false; }
            }
        }
        for (int R = 0; R < rows; ++R) {
            if (file[(size_t)P.storeStage[(size_t)R]][(size_t)R] != seq[(size_t)R]) { if (why) *why = "plan check: a row would be stored by the wrong stage"; return false; }
            state[(size_t)R] = seq[(size_t)R];
        }
        (void)stageOf;
    }
    return true;
}

// the records of one stage as a stream of its own (ENDSAMPLE and the fetch pad behind it)
std::vector<MicroOp> stageRecords(const std::vector<MicroOp>& all, size_t from, size_t to) {
    std::vector<MicroOp> out(all.begin() + (long)from, all.begin() + (long)to);
    MicroOp end{};
    end.w[0] = AS_ENDSAMPLE;
    out.push_back(end);
    MicroOp nop{};
    nop.w[0] = AS_NOP;
    for (int k = 0; k < 4; ++k) out.push_back(nop);
    return out;
}
}  // namespace

// LDS of a staged program: the LOG/EXP tables (shared by all stages: every wavefront stages the same bytes), one flag row per
// stage ("my packets may hold non-finite values", Translator::stageFlagCheck), the ring of 4 * group packet buffers (the
// generated code steps through them with an add and an AND: the stride is a power of two), the epilogue's scratch
bool stageLdsLayout(const XlateProgram& program, const StagePlan& plan, uint32_t ldsBudget, int maxGroup, StageLds* L, int pinGroup) {
    const int K = (int)plan.cuts.size() + 1;
    const uint32_t tableBytes = program.lutTables.empty() ? 0u : lutLds().bytes(program.lutTables.size());
    L->cutOff.clear();
    uint32_t bufStride = 0;
    for (const auto& l : plan.live) { L->cutOff.push_back(bufStride); bufStride += 256u * (uint32_t)l.size(); }
    uint32_t pow2 = 256u;
    while (pow2 < bufStride) pow2 <<= 1;
    L->bufStride = pow2;
    L->scratchBytes = (uint32_t)K * 512u;   // the template's epilogue (counts and flags of the stages -> stage 0)
    int group = kStageGroupMax;
    while (group > 1 && group > maxGroup) group /= 2;
    if (pinGroup == 1 || pinGroup == 2 || pinGroup == 4) group = pinGroup;   // tests: a shorter ring than the LDS would allow
    // without tables the ring lies at address 0 (the pointer's and-mask needs no base: StageInfo::ptrBias), flag rows and scratch
    // behind it; with tables: [tables][flags][ring][scratch]
    static const bool pairOff = knobInt(FX_DIAG_KNOB("FX_XLATE_LDS2"), 1) == 0;   // diagnostics
    L->ringFirst = tableBytes == 0 && !pairOff;
    const uint32_t fixed = ((tableBytes + 255u) & ~255u) + 256u * (uint32_t)K + L->scratchBytes;
    while (group > 1 && fixed + 4u * (uint32_t)group * L->bufStride > ldsBudget) group /= 2;
    L->group = group;
    const uint32_t ring = 4u * (uint32_t)group * L->bufStride;
    if (L->ringFirst) {
        L->bufBase = 0;
        L->flagBase = ring;
        L->scratchOff = ring + 256u * (uint32_t)K;
    } else {
        L->flagBase = (tableBytes + 255u) & ~255u;
        L->bufBase = L->flagBase + 256u * (uint32_t)K;
        L->scratchOff = L->bufBase + ring;
    }
    L->bytes = fixed + ring;
    return !(L->bufBase + L->bufStride > 0xff00u || L->bytes > std::min(ldsBudget, 160u * 1024u));
}

bool buildStagedImage(const std::vector<MicroOp>& steadyRecords, const std::vector<MicroOp>& lastRecords, const XlateTemplate& tmpl,
                      const XlateProgram& program, const StagePlan& plan, XlateImage* out, std::vector<std::vector<uint32_t>>* codeOut,
                      std::vector<std::string>* listingOut, std::string* err, uint32_t ldsBudget, int maxGroup, int pinGroup) {
    const int K = (int)plan.cuts.size() + 1;
    if (K < 2) { if (err) *err = "not a staged plan"; return false; }
    size_t n = 0;
    while (n < steadyRecords.size() && steadyRecords[n].w[0] != AS_ENDSAMPLE) ++n;
    StageLds L;
    if (!stageLdsLayout(program, plan, ldsBudget, maxGroup, &L, pinGroup)) { if (err) *err = "staged program: packets beyond the LDS"; return false; }
    const std::vector<uint32_t>& cutOff = L.cutOff;
    const uint32_t bufStride = L.bufStride, flagBase = L.flagBase, bufBase = L.bufBase;
    const int group = L.group;
    std::vector<std::vector<uint32_t>> code((size_t)K * 4 + 1);
    std::vector<std::string> listing((size_t)K * 4 + 1);
    uint32_t at = tmpl.holeOff;
    out->stages = K;
    out->stageDesc.assign((size_t)K, StageDescriptor{});
    out->stageStoreRows.assign((size_t)K, {});
    out->plan = plan;
    out->wildRow = program.wildRow;
    out->steady = XlateStats();
    out->last = XlateStats();
    for (size_t r = 0; r < plan.storeStage.size(); ++r) out->stageStoreRows[(size_t)plan.storeStage[r]].push_back((int)r);
    int worstValu = -1;
    HoistPlan stage0Hoist;
    for (int k = 0; k < K; ++k) {
        const size_t from = k == 0 ? 0 : (size_t)plan.cuts[(size_t)k - 1], to = k + 1 == K ? n : (size_t)plan.cuts[(size_t)k];
        const std::vector<MicroOp> steady = stageRecords(steadyRecords, from, to), last = stageRecords(lastRecords, from, to);
        // the stage as a program of its own: its delay-line reads may lead (stage 0 owns all of them), its PCM channels; tables,
        // row classes and the LDS layout are the whole program's
        std::vector<int> inRows = program.inRows;
        for (size_t c = 0; c < inRows.size(); ++c)
            if (!((plan.inMask[(size_t)k] >> c) & 1u)) inRows[c] = -1;
        XlateProgram p = xlateProgramOf(steady, last, program.iSize, program.xSize, (int)program.wildRow.size(), inRows, program.latchRows);
        p.lutTables = program.lutTables;
        p.wildRow = program.wildRow;
        p.tramStreaming = program.tramStreaming;
        p.stage.index = k;
        p.stage.count = K;
        p.stage.bufBase = bufBase;
        p.stage.flagBase = flagBase;
        p.stage.bufStride = bufStride;
        p.stage.group = group;
        if (k > 0) { p.stage.recvRows = plan.live[(size_t)k - 1]; p.stage.recvOff = cutOff[(size_t)k - 1]; }
        if (k + 1 < K) { p.stage.sendRows = plan.live[(size_t)k]; p.stage.sendOff = cutOff[(size_t)k]; }
        p.stage.ptrBias = L.ringFirst ? (k > 0 ? p.stage.recvOff : p.stage.sendOff) : 0u;
        p.stage.storeMask = 0;
        for (size_t c = 0; c < plan.pcmStage.size(); ++c)
            if (plan.pcmStage[c] == k) p.stage.storeMask |= 1u << c;
        // a leading delay-line read issued a sample ahead lands in its row while this sample's tail is still to come: a row the
        // tail hands to the next stage must not be one of those (the hoist point only knows the stage's own records)
        for (int q = 0; q < p.hoist.leadCount; ++q)
            if (std::find(p.stage.sendRows.begin(), p.stage.sendRows.end(), (int)steady[(size_t)q].w[5]) != p.stage.sendRows.end()) {
                p.hoist = HoistPlan();
                break;
            }
        // steps far shorter than a trip to memory: PCM input in bursts (a stage with delay lines keeps the loop's own prefetch)
        p.stage.inRing = (p.tramOpsInline == 0 && p.hoist.leadCount == 0 && !FX_DIAG_KNOB("FX_STAGES_NO_RING")) ? -2 : -1;
        if (k == 0) stage0Hoist = p.hoist;
        XlateImage one;
        std::vector<uint32_t> c5[5];
        std::string t5[5];
        // planXlate lays its streams out from the hole's start; here they follow the previous stage's
        XlateTemplate shifted = tmpl;
        shifted.holeOff = at;
        shifted.holeBytes = tmpl.holeBytes - (at - tmpl.holeOff);
        if (!planXlate(steady, last, shifted, p, &one, c5, listingOut ? t5 : nullptr, err)) return false;
        // (its run-once code is dropped: one copy for the whole program follows the last stage)
        uint32_t end = at;
        for (int q = 0; q < 4; ++q) {
            code[(size_t)k * 4 + (size_t)q] = c5[q];
            listing[(size_t)k * 4 + (size_t)q] = t5[q];
            if (!c5[q].empty()) end = std::max(end, one.base[q] + align64((uint32_t)c5[q].size() * 4));
        }
        StageDescriptor& d = out->stageDesc[(size_t)k];
        d.steadyFast = one.steadyFastOff;
        d.steadyExact = one.steadyOff;
        d.lastFast = one.lastFastOff;
        d.lastExact = one.lastOff;
        for (int q = 0; q < 4; ++q) out->base[q] = one.base[q];  // (of the last stage: diagnostics only)
        out->stageBases.push_back({one.base[0], one.base[1], one.base[2], one.base[3]});
        // statistics: the stages' vector instructions ADD UP to the work of one sample of one instance group (the slowest
        // stage sets the pace: worstValu)
        worstValu = std::max(worstValu, one.steady.valu);
        for (auto pr : {std::make_pair(&out->steady, &one.steady), std::make_pair(&out->last, &one.last)}) {
            XlateStats& a = *pr.first;
            const XlateStats& b = *pr.second;
            a.inlined += b.inlined; a.called += b.called; a.instructions += b.instructions; a.valu += b.valu; a.valuSlow += b.valuSlow;
            a.valuClocks += b.valuClocks; a.fusedSkips += b.fusedSkips; a.regions += b.regions; a.unitMultipliers += b.unitMultipliers;
            a.reusedProducts += b.reusedProducts; a.fusedZeroAdds += b.fusedZeroAdds; a.unsaturated += b.unsaturated;
        }
        out->vgprConstants = one.vgprConstants;
        at = end;
    }
    // the run-once code: tables of the whole program, the hoist decision of stage 0
    out->initOff = 0;
    out->ldsBytes = L.bytes;
    for (StageDescriptor& d : out->stageDesc) d.scratchOff = L.scratchOff;
    {
        XlateProgram initProg = program;
        initProg.hoist = stage0Hoist;
        if (!initProg.lutTables.empty() || initProg.hoist.leadCount > 0) {
            emitInit(initProg, &code[(size_t)K * 4], listingOut ? &listing[(size_t)K * 4] : nullptr);
            out->initOff = at;
            at += align64((uint32_t)code[(size_t)K * 4].size() * 4);
        }
    }
    out->codeBytes = at - tmpl.holeOff;
    out->slowestStageValu = worstValu;
    if (out->codeBytes + 4 > tmpl.holeBytes) { if (err) *err = "translated program larger than the code hole of the template"; return false; }
    out->elf.assign(tmpl.image, tmpl.image + tmpl.imageBytes);
    for (int k = 0; k < K; ++k)
        for (int q = 0; q < 4; ++q) {
            const std::vector<uint32_t>& c = code[(size_t)k * 4 + (size_t)q];
            placeCode(tmpl, &out->elf, out->stageBases[(size_t)k][(size_t)q], c);
        }
    if (out->initOff) placeCode(tmpl, &out->elf, out->initOff, code[(size_t)K * 4]);
    if (codeOut) *codeOut = code;
    if (listingOut) *listingOut = listing;
    return true;
}

}  // namespace fx
