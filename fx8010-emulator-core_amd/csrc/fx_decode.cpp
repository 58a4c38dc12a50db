// fx_decode.cpp — static analysis + lowering (see fx_decode.hpp for the idea).
#include "fx_decode.hpp"

#include <algorithm>
#include <cstring>

namespace fx {
namespace {

inline uint32_t bitsOf(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// what `static_cast<int>(float)` means in the reference build (x86 cvttss2si)
inline int32_t x86Trunc(float v) {
    if (!(v < 2147483648.0f) || v < -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}

bool writesResult(int op) { return op != SKIP && op != IDELAY && op != XDELAY && op != END; }
bool setsCcr(int op) { return writesResult(op); }  // every ALU op ends in setCCR(R)

}  // namespace

StateLayout makeLayout(int nRegs, int channels) {
    StateLayout L;
    L.nRegs = nRegs;
    L.outBase = nRegs;
    L.cursorBase = L.outBase + channels;
    L.noiseBase = L.cursorBase + 4;
    L.oodRow = L.noiseBase + 2;
    L.countLo = L.oodRow + 1;
    L.countHi = L.countLo + 1;
    L.totalRows = L.countHi + 1;
    return L;
}

Lowered lowerProgram(const Program& prog, const std::vector<float>& hostValue, const std::vector<uint8_t>& forcedLane, int instPerLane,
                     bool ldsBookkeeping, uint32_t rowPitch) {
    Lowered out;
    out.instPerLane = instPerLane;
    const uint32_t rowBytes = rowPitch ? rowPitch : 256u * (uint32_t)instPerLane;
    out.rowPitch = rowBytes;
    const int nRegs = (int)prog.regs.size();
    const int P = (int)prog.instrs.size();
    const int CH = prog.numChannels;
    out.layout = makeLayout(nRegs, CH);
    if (P == 0) { out.error = "program has no instructions"; return out; }
    if (CH < 1 || CH > 4) { out.error = "1..4 channels supported by the device path"; return out; }
    // setChannels() can raise the loader's bound after construction (reference FX8010.h:73); the reference would then index
    // its one-sample buffers out of bounds (FX8010.cpp:1056,1231): outside the parity domain, refused here
    for (const Gpr& g : prog.regs)
        if ((g.type == R_INPUT || g.type == R_OUTPUT) && (g.io < 0 || g.io >= CH)) {
            out.error = "register '" + g.name + "' uses I/O index " + std::to_string(g.io) + " but the object was constructed with " + std::to_string(CH) + " channel(s)";
            return out;
        }

    auto isReadDelay = [&](const Instr& I) { return (I.op == IDELAY || I.op == XDELAY) && prog.regs[I.r].type == R_READ; };
    auto isWriteDelay = [&](const Instr& I) { return (I.op == IDELAY || I.op == XDELAY) && prog.regs[I.r].type == R_WRITE; };
    auto chanOf = [&](const Instr& I) { return prog.regs[I.a].io; };  // the reference indexes the input with A's IOIndex for A, X and Y

    // ---- 1. which registers must live per instance --------------------------------------
    std::vector<uint8_t> lane(nRegs, 0);
    lane[0] = 1;  // ccr
    for (int r = 0; r < nRegs; ++r) {
        if (r < (int)forcedLane.size() && forcedLane[r]) lane[r] = 1;
        if (prog.regs[r].type == R_OUTPUT) lane[r] = 1;
    }
    for (const Instr& I : prog.instrs) {
        if (writesResult(I.op)) lane[I.r] = 1;
        if (isReadDelay(I)) lane[I.a] = 1;
        if (I.hasNoise) {
            int t = prog.regs[I.a].name == "noise" ? I.a : (prog.regs[I.x].name == "noise" ? I.x : I.y);
            lane[t] = 1;
            out.usesNoise = true;
        }
        if (I.op == IDELAY && (isReadDelay(I) || isWriteDelay(I))) out.usesITram = true;
        if (I.op == XDELAY && (isReadDelay(I) || isWriteDelay(I))) out.usesXTram = true;
        if (I.op == LOG || I.op == EXP) out.usesLut = true;
    }

    // ---- 2. SKIP shadows ------------------------------------------------------------------
    std::vector<uint8_t> shadow(P, 0);
    bool shadowAll = false;
    for (int k = 0; k < P && !shadowAll; ++k) {
        const Instr& I = prog.instrs[k];
        if (I.op != SKIP) continue;
        if (lane[I.y]) { shadowAll = true; break; }  // per-instance skip distance
        int32_t cnt = x86Trunc(hostValue[I.y]);
        if (cnt == 0) continue;
        if (cnt < 0) cnt = 1;  // a negative count skips exactly one instruction (FX8010.cpp:1238)
        if (cnt >= P) { shadowAll = true; break; }
        for (int j = 1; j <= cnt; ++j) shadow[(k + j) % P] = 1;
    }
    if (shadowAll) std::fill(shadow.begin(), shadow.end(), 1);
    // END inside a shadow: a lane may need further passes over the program (FX8010.cpp:1033,1243)
    bool anyEndUnshadowed = false;
    for (int k = 0; k < P; ++k)
        if (prog.instrs[k].op == END && !shadow[k]) anyEndUnshadowed = true;
    out.multipass = !anyEndUnshadowed;
    if (out.multipass) std::fill(shadow.begin(), shadow.end(), 1);

    // ---- 3. INPUT registers: alias onto the channel's input row where exact ---------------
    out.inRow.assign(CH, -1);
    std::vector<int> aliasChan(nRegs, -1);  // >=0: register is an alias of that channel's input row
    {
        std::vector<int> chanSeen(nRegs, -2);      // -2 none yet, -1 conflicting
        std::vector<uint8_t> uncond(nRegs, 0), delayTarget(nRegs, 0);
        for (int k = 0; k < P; ++k) {
            const Instr& I = prog.instrs[k];
            if (isReadDelay(I)) delayTarget[I.a] = 1;
            if (!I.hasInput) continue;
            const int ops[3] = {I.a, I.x, I.y};
            for (int o : ops) {
                if (prog.regs[o].type != R_INPUT) continue;
                int c = chanOf(I);
                if (chanSeen[o] == -2) chanSeen[o] = c;
                else if (chanSeen[o] != c) chanSeen[o] = -1;
                if (!shadow[k]) uncond[o] = 1;
            }
        }
        for (int r = 0; r < nRegs; ++r) {
            if (prog.regs[r].type != R_INPUT || chanSeen[r] == -2) continue;  // never read: stays uniform
            if (chanSeen[r] >= 0 && uncond[r] && !delayTarget[r])
                aliasChan[r] = chanSeen[r];
            lane[r] = 1;
        }
    }

    // ---- 4. rows ------------------------------------------------------------------------------
    out.rowOfReg.assign(nRegs, -1);
    int nextRow = 0;
    out.rowOfReg[0] = nextRow++;  // ccr is row 0
    for (int c = 0; c < CH; ++c) {
        bool used = false;
        for (const Instr& I : prog.instrs)
            if (I.hasInput && chanOf(I) == c) used = true;
        if (used) out.inRow[c] = nextRow++;
    }
    out.latchRow.assign(CH, -1);
    for (int c = 0; c < CH; ++c) out.latchRow[c] = nextRow++;
    for (int r = 1; r < nRegs; ++r) {
        if (!lane[r]) continue;
        if (aliasChan[r] >= 0) out.rowOfReg[r] = out.inRow[aliasChan[r]];
        else out.rowOfReg[r] = nextRow++;
    }
    // bookkeeping rows, only what this program needs
    bool anyShadow = false;  // a SKIP writes numSkip even when its count is statically 0
    for (int k = 0; k < P; ++k) anyShadow = anyShadow || shadow[k] || prog.instrs[k].op == SKIP;
    if (ldsBookkeeping) {
        out.oodRow = nextRow++;
        out.zeroRows.push_back(out.oodRow);
        if (anyShadow) { out.skipRow = nextRow; nextRow += 3; for (int q = 0; q < 3; ++q) out.zeroRows.push_back(out.skipRow + q); }
        if (out.usesITram || out.usesXTram) { out.cursorRow = nextRow; nextRow += 4; }
        if (out.usesNoise) { out.noiseRow = nextRow; nextRow += 2; }
        if (out.multipass) { out.aliveRow = nextRow; nextRow += 2; out.zeroRows.push_back(out.aliveRow); out.zeroRows.push_back(out.aliveRow + 1); }
    }
    out.nRows = nextRow;
    for (int r = 0; r < nRegs; ++r) (lane[r] ? out.nLaneRegs : out.nUniformRegs)++;
    if (rowPitch == 0 && (long)out.nRows * rowBytes > 160 * 1024) {
        out.error = "program needs " + std::to_string(out.nRows) + " per-instance rows: more than the 160 KiB LDS register file holds at " +
                    std::to_string(instPerLane) + " instance(s) per lane";
        return out;
    }
    for (int r = 0; r < nRegs; ++r) {
        if (!lane[r]) continue;
        if (aliasChan[r] < 0) out.loadRows.push_back({(uint16_t)out.rowOfReg[r], (uint16_t)r});
        out.storeRows.push_back({(uint16_t)out.rowOfReg[r], (uint16_t)r});
    }
    for (int c = 0; c < CH; ++c) {
        out.loadRows.push_back({(uint16_t)out.latchRow[c], (uint16_t)(out.layout.outBase + c)});
        out.storeRows.push_back({(uint16_t)out.latchRow[c], (uint16_t)(out.layout.outBase + c)});
    }
    if (out.cursorRow >= 0)
        for (int q = 0; q < 4; ++q) {
            out.loadRows.push_back({(uint16_t)(out.cursorRow + q), (uint16_t)(out.layout.cursorBase + q)});
            out.storeRows.push_back({(uint16_t)(out.cursorRow + q), (uint16_t)(out.layout.cursorBase + q)});
        }
    if (out.noiseRow >= 0)
        for (int q = 0; q < 2; ++q) {
            out.loadRows.push_back({(uint16_t)(out.noiseRow + q), (uint16_t)(out.layout.noiseBase + q)});
            out.storeRows.push_back({(uint16_t)(out.noiseRow + q), (uint16_t)(out.layout.noiseBase + q)});
        }

    // ---- 5. CCR liveness ------------------------------------------------------------------------
    // A CCR write is observable if, walking forward (wrapping into the next sample), a reader of
    // register 0 comes before an instruction that overwrites CCR unconditionally.
    auto readsCcr = [&](const Instr& I) {
        if (I.op == SKIP) return true;
        if (I.op == END) return false;
        return I.a == 0 || I.x == 0 || I.y == 0;
    };
    std::vector<uint8_t> ccrLive(P, 0);
    for (int i = 0; i < P; ++i) {
        if (!setsCcr(prog.instrs[i].op)) continue;
        if (out.multipass) { ccrLive[i] = 1; continue; }
        for (int d = 1; d <= P; ++d) {
            int j = (i + d) % P;
            const Instr& J = prog.instrs[j];
            if (readsCcr(J)) { ccrLive[i] = 1; break; }
            if (setsCcr(J.op) && !shadow[j]) break;  // killed (also covers j == i after a full lap)
        }
    }

    // ---- 6. TRAM sizing ---------------------------------------------------------------------------
    out.tramDane = (prog.options & kOptTramDane) != 0;
    auto tramSlots = [&](int op, int size, int cap) {
        if (size <= 0) return 0;
        if (out.tramDane) return std::min(size, cap);  // ring addressing: every slot index is below size
        int maxOff = 0;
        bool dynamic = false;
        for (const Instr& I : prog.instrs) {
            if (I.op != op || !isWriteDelay(I)) continue;
            if (lane[I.y]) dynamic = true;
            else maxOff = std::max(maxOff, std::min(std::max(x86Trunc(hostValue[I.y]), 0), size - 1));
        }
        long want = dynamic ? 2L * size - 1 : (long)size + maxOff;
        return (int)std::min<long>(want, cap);
    };
    out.iSlots = out.usesITram ? tramSlots(IDELAY, std::min(prog.iTramSize, kMaxITram), kMaxITram) : 0;
    out.xSlots = out.usesXTram ? tramSlots(XDELAY, std::min(prog.xTramSize, kMaxXTram), kMaxXTram) : 0;
    if (prog.iTramSize > kMaxITram || prog.xTramSize > kMaxXTram) {
        out.error = "TRAM size beyond the reference's arrays (8192 / 1048576): outside the parity domain";
        return out;
    }

    // ---- 7. emission ------------------------------------------------------------------------------
    auto rowOff = [&](int reg) { return (uint32_t)out.rowOfReg[reg] * rowBytes; };

    for (int pass = 0; pass < 2; ++pass) {
        std::vector<MicroOp>& dst = pass == 0 ? out.steady : out.last;
        for (int k = 0; k < P; ++k) {
            const Instr& I = prog.instrs[k];
            const uint32_t sh = shadow[k] ? F_SHADOW : 0;

            // operand -> row of this instruction (an aliased INPUT always resolves to chanOf(I)'s row)
            auto operandRow = [&](int reg) -> int {
                if (prog.regs[reg].type == R_INPUT && aliasChan[reg] >= 0) return out.inRow[chanOf(I)];
                return out.rowOfReg[reg];
            };
            auto mov = [&](uint32_t flags, uint32_t dstOff, uint32_t srcOff) {
                MicroOp m{};
                m.w[0] = H_MOV | flags | F_WRITE_R | F_UX | F_UY;
                m.w[1] = dstOff;
                m.w[2] = srcOff;
                dst.push_back(m);
            };

            // prefix: refresh of INPUT operands that could not be aliased (FX8010.cpp:1053-1061)
            if (I.hasInput) {
                int seen[3] = {-1, -1, -1};
                const int ops[3] = {I.a, I.x, I.y};
                for (int q = 0; q < 3; ++q) {
                    int o = ops[q];
                    if (prog.regs[o].type != R_INPUT || aliasChan[o] >= 0) continue;
                    if (o == seen[0] || o == seen[1]) continue;
                    seen[q] = o;
                    mov(sh | F_PREFIX, rowOff(o), (uint32_t)out.inRow[chanOf(I)] * rowBytes);
                }
            }
            if (I.hasNoise) {
                int t = prog.regs[I.a].name == "noise" ? I.a : (prog.regs[I.x].name == "noise" ? I.x : I.y);
                MicroOp m{};
                m.w[0] = H_NOISE | sh | F_PREFIX | F_WRITE_R | F_UA | F_UX | F_UY;
                m.w[1] = rowOff(t);
                dst.push_back(m);
            }

            MicroOp m{};
            uint32_t h = H_NOP;
            switch (I.op) {
                case MACS: case MACINTS: h = H_MACS; break;  // identical expressions (FX8010.cpp:1077-1085,1095-1103)
                case MACSN: h = H_MACSN; break;
                case ACC3: h = H_ACC3; break;
                case INTERP: h = H_INTERP; break;
                case MACW: h = H_MACW; break;
                case MACWN: h = H_MACWN; break;
                case MACINTW: h = H_MACINTW; break;
                case MACMV: h = H_MOV; break;
                case ANDXOR: h = H_ANDXOR; break;
                case TSTNEG: h = H_TSTNEG; break;
                case LIMIT: h = H_LIMIT; break;
                case LIMITN: h = H_LIMITN; break;
                case LOG: h = H_LOG; break;
                case EXP: h = H_EXP; break;
                case SKIP: h = H_SKIP; break;
                case IDELAY: h = isReadDelay(I) ? H_TRAM_IR : (isWriteDelay(I) ? H_TRAM_IW : H_NOP); break;
                case XDELAY: h = isReadDelay(I) ? H_TRAM_XR : (isWriteDelay(I) ? H_TRAM_XW : H_NOP); break;
                case END: h = H_END; break;
                default: h = H_NOP; break;
            }
            uint32_t flags = sh | F_COUNT;
            if (setsCcr(I.op) && (pass == 1 || ccrLive[k])) flags |= F_CCR;
            if (h >= H_TRAM_IR && h <= H_TRAM_XW && out.tramDane) {
                flags |= F_TRAM_DANE | ((prog.options & kOptTramAddrShift) ? F_TRAM_SHIFT : 0u);
                if ((h == H_TRAM_IR || h == H_TRAM_XR) && (prog.options & kOptTramAddrShift) && (prog.options & kOptTramInterp)) flags |= F_TRAM_INTERP;
            }

            // operand slots: LDS byte offset of a per-instance row, or the immediate of a uniform register;
            // slots an opcode does not read are immediates too, so the kernel issues no LDS read for them
            auto src = [&](int reg, uint32_t uflag, int slot) {
                if (lane[reg]) m.w[slot] = (uint32_t)operandRow(reg) * rowBytes;
                else { flags |= uflag; m.w[slot] = bitsOf(hostValue[reg]); }
            };
            auto unused = [&](uint32_t uflag, int slot) { flags |= uflag; m.w[slot] = 0; };
            if (writesResult(I.op)) { m.w[1] = rowOff(I.r); flags |= F_WRITE_R; }
            if (h == H_TRAM_IR || h == H_TRAM_XR) {
                m.w[1] = rowOff(I.a);  // the read lands in A (FX8010.cpp:1192,1204)
                flags |= F_WRITE_R;
                unused(F_UA, 2); unused(F_UX, 3); src(I.y, F_UY, 4);
            } else if (h == H_TRAM_IW || h == H_TRAM_XW) {
                src(I.a, F_UA, 2); unused(F_UX, 3); src(I.y, F_UY, 4);
            } else if (h == H_SKIP) {
                unused(F_UA, 2); src(I.x, F_UX, 3); src(I.y, F_UY, 4);
            } else if (h == H_LOG || h == H_EXP) {
                src(I.a, F_UA, 2);
                unused(F_UY, 4);  // Y (sign) is ignored by the reference (FX8010.cpp:1114)
                if (lane[I.x]) m.w[3] = (uint32_t)operandRow(I.x) * rowBytes;
                else {
                    flags |= F_UX;
                    int32_t t = x86Trunc(hostValue[I.x]);
                    if (t < 0 || t > 31) { flags |= F_STATIC_OOD; t = t < 0 ? 0 : 31; }
                    m.w[5] = (uint32_t)((h == H_EXP ? 32 : 0) + t);
                    m.w[3] = bitsOf(hostValue[I.x]);
                }
            } else if (h == H_MOV) {  // MACMV: R = A
                src(I.a, F_UA, 2); unused(F_UX, 3); unused(F_UY, 4);
            } else if (h == H_END || h == H_NOP) {
                unused(F_UA, 2); unused(F_UX, 3); unused(F_UY, 4);
            } else {
                src(I.a, F_UA, 2); src(I.x, F_UX, 3); src(I.y, F_UY, 4);
            }
            m.w[0] = h | flags;
            dst.push_back(m);

            // postfix: output latch (FX8010.cpp:1229-1233) after ANY executed instruction whose R is an OUTPUT
            if (prog.regs[I.r].type == R_OUTPUT)
                mov(sh | F_POSTFIX, (uint32_t)out.latchRow[prog.regs[I.r].io] * rowBytes, rowOff(I.r));
        }
    }

    for (int k = 0; k < P; ++k) {
        if (shadow[k]) out.nShadowed++; else out.staticCount++;
        if (ccrLive[k]) out.nCcrLive++;
        const Instr& I = prog.instrs[k];
        if (isReadDelay(I) || isWriteDelay(I)) out.tramOpsPerSample++;
    }
    return out;
}

}  // namespace fx
