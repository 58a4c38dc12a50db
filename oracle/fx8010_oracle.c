/* oracle/fx8010_oracle.c — TEST INFRASTRUCTURE ONLY (parity checker, never shipped).
 *
 * Scalar C restatement of the reference interpreter easypx/FX8010-Emulator-Core
 * (class Klangraum::FX8010).  It follows the reference's algorithm operation by
 * operation — same IEEE-754 types, same order, no fused multiply-add — so that it
 * can be compiled anywhere (the reference sources do not travel to the GPU box)
 * and still produce the reference's bits.  Build: gcc -O2 -ffp-contract=off
 * (oracle/Makefile).  Citations are path:line inside /root/reference.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 * golden vectors generated from the compiled, unmodified reference
 * (oracle/_ref, tests/golden/make_golden.py), and, when oracle/_ref is present,
 * tests/test_oracle_vs_ref.py diff it live against the reference itself.
 *
 * Where the reference has undefined behaviour (out-of-bounds reads, division by
 * zero, an endless loop) this file defines a behaviour, raises a sticky
 * FXO_OOD_* flag, and the case is OUTSIDE the parity domain.
 */
#define _GNU_SOURCE
#include "fx8010_oracle.h"

#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- model: include/FX8010.h:79-100 (Opcode), :127-142 (RegisterType) ---- */
enum { OP_MACS = 0, OP_MACSN, OP_MACW, OP_MACWN, OP_MACINTS, OP_MACINTW, OP_ACC3, OP_MACMV,
       OP_ANDXOR, OP_TSTNEG, OP_LIMIT, OP_LIMITN, OP_LOG, OP_EXP, OP_INTERP, OP_SKIP,
       OP_IDELAY, OP_XDELAY, OP_END };
enum { RT_STATIC = 0, RT_TEMP, RT_CONTROL, RT_INPUT, RT_OUTPUT, RT_CONST, RT_ITRAMSIZE,
       RT_XTRAMSIZE, RT_READ, RT_WRITE, RT_AT, RT_CCR };

#define MAX_IDELAY_SIZE 8192     /* include/FX8010.h:41 */
#define MAX_XDELAY_SIZE 1048576  /* include/FX8010.h:42 */
#define PASS_CAP 64              /* ours: the reference would spin forever */

typedef struct { int type; char* name; float value; int io; } gpr_t;          /* FX8010.h:167-174 */
typedef struct { int op, r, a, x, y; int has_in, has_out, has_noise; } ins_t; /* FX8010.h:180-191 */
typedef struct { char* desc; int row; } err_t;                                /* FX8010.h:63-67 */

struct fxo {
    int channels;
    gpr_t* regs; int nregs, capregs;
    ins_t* ins; int nins, capins;
    err_t* errs; int nerrs, caperrs;
    char** controls; int nctl, capctl;
    char* meta_key[6]; char* meta_val[6];
    int error_counter;            /* FX8010.h:273 */
    int itram_size, xtram_size;   /* FX8010.h:204-205 */
    float* itram; float* xtram;   /* FX8010.h:210-211 (zero-initialised here) */
    int iw, ir, xw, xr;           /* FX8010.h:214-217 */
    double acc;                   /* FX8010.h:162 */
    int64_t icount;               /* FX8010.h:163 (int there) */
    float* outbuf;                /* FX8010.h:164 */
    int32_t g_x1, g_x2;           /* FX8010.h:290-291 */
    double lut_log[32][65];       /* FX8010.h:197-198; entry 64 is our finite pad */
    double lut_exp[32][65];
    int ready;
    unsigned ood;
    unsigned opts;                /* FXO_OPT_*: behaviour beyond the reference, off by default (see fx8010_oracle.h) */
};

/* ------------------------------------------------------------------ helpers */
static char* xstrdup_n(const char* s, size_t n) {
    char* p = (char*)malloc(n + 1);
    memcpy(p, s, n); p[n] = 0; return p;
}
static int is_s(int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }
static int is_w(int c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_'; }
static int is_d(int c) { return c >= '0' && c <= '9'; }

static void push_error(fxo_t* f, const char* desc, int row) {
    if (f->nerrs == f->caperrs) { f->caperrs = f->caperrs ? 2 * f->caperrs : 8; f->errs = (err_t*)realloc(f->errs, sizeof(err_t) * f->caperrs); }
    f->errs[f->nerrs].desc = xstrdup_n(desc, strlen(desc));
    f->errs[f->nerrs].row = row;
    f->nerrs++;
}
static int push_reg(fxo_t* f, int type, const char* name, size_t nlen, float v, int io) {
    if (f->nregs == f->capregs) { f->capregs = f->capregs ? 2 * f->capregs : 32; f->regs = (gpr_t*)realloc(f->regs, sizeof(gpr_t) * f->capregs); }
    gpr_t* g = &f->regs[f->nregs];
    g->type = type; g->name = xstrdup_n(name, nlen); g->value = v; g->io = io;
    return f->nregs++;
}
static void push_ins(fxo_t* f, ins_t in) {
    if (f->nins == f->capins) { f->capins = f->capins ? 2 * f->capins : 64; f->ins = (ins_t*)realloc(f->ins, sizeof(ins_t) * f->capins); }
    f->ins[f->nins++] = in;
}
/* findRegisterIndexByName, source/FX8010.cpp:880-890 */
static int find_reg(fxo_t* f, const char* name, size_t nlen) {
    for (int i = 0; i < f->nregs; ++i)
        if (strlen(f->regs[i].name) == nlen && memcmp(f->regs[i].name, name, nlen) == 0) return i;
    return -1;
}

/* error strings: source/FX8010.cpp:25-35 */
static void err_io_range(fxo_t* f, char* buf, size_t n) { snprintf(buf, n, "I/O Index ausserhalb des gueltigen Bereichs (max. %d)", f->channels); }
#define ERR_NONE "Kein Fehler"
#define ERR_MULTI "Mehrfache Variablendeklaration"
#define ERR_UNDECL "Variable nicht deklariert"
#define ERR_R_INPUT "Verwendung von Input fuer R ist nicht erlaubt"
#define ERR_NO_END "Kein 'END' gefunden"
#define ERR_SYNTAX "Ungueltige Syntax"
#define ERR_ITRAM "iTRAM Size ausserhalb des gueltigen Bereichs (max. 8192)"
#define ERR_XTRAM "xRAM Size ausserhalb des gueltigen Bereichs (max. 1048576)"

/* ---- LUT build: source/FX8010.cpp:63-105, 129-199 ---- */
static void build_luts(fxo_t* f) {
    const int n = 32;
    for (int e = 0; e < 32; ++e) {
        double tl[32], te[32];
        double step = (1.0 - 0.0) / (n - 1);                    /* :132, :153 */
        for (int i = 0; i < n; ++i) {
            double x = 0.0 + (i * step);                        /* :137, :157 */
            tl[i] = pow(x, 1.0 / (double)(float)e);             /* :142  pow(x, 1.0/static_cast<float>(e)) */
            te[i] = pow(x, (double)(float)e);                   /* :159 */
        }
        for (int k = 0; k < n; ++k) {
            /* mirrorYVector :167-175 then negateVector :190-199 — the negation loop
             * iterates with `float value`, so the negative half is rounded through float */
            float ml = (float)tl[n - 1 - k];
            float me = (float)te[n - 1 - k];
            f->lut_log[e][k] = (double)(-ml);
            f->lut_exp[e][k] = (double)(-me);
            f->lut_log[e][n + k] = tl[k];                       /* concatenateVectors :179-186 */
            f->lut_exp[e][n + k] = te[k];
        }
        f->lut_log[e][64] = f->lut_log[e][63];                  /* ours: finite pad for the x==1.0 read of [idx+1] */
        f->lut_exp[e][64] = f->lut_exp[e][63];
    }
}

/* ---- construction: source/FX8010.cpp:9-125 ---- */
fxo_t* fxo_create(int channels) {
    fxo_t* f = (fxo_t*)calloc(1, sizeof(fxo_t));
    f->channels = channels;
    f->error_counter = 1;
    push_error(f, ERR_NONE, 1);                                  /* :38-42 */
    push_reg(f, RT_CCR, "ccr", 3, 0.0f, 0);                      /* :50 index 0 */
    push_reg(f, RT_READ, "read", 4, 0.0f, 0);                    /* :53 index 1 */
    push_reg(f, RT_WRITE, "write", 5, 0.0f, 0);                  /* :54 index 2 */
    push_reg(f, RT_AT, "at", 2, 0.0f, 0);                        /* :55 index 3 */
    build_luts(f);
    f->outbuf = (float*)calloc((size_t)(channels > 0 ? channels : 1), sizeof(float)); /* :122 */
    f->g_x1 = (int32_t)0x70f4f854;                               /* FX8010.h:290 */
    f->g_x2 = (int32_t)0xe1e9f0a7;                               /* FX8010.h:291 */
    f->itram = (float*)calloc(MAX_IDELAY_SIZE, sizeof(float));
    f->xtram = NULL;                                             /* allocated on first use (4 MiB) */
    return f;
}

void fxo_destroy(fxo_t* f) {
    if (!f) return;
    for (int i = 0; i < f->nregs; ++i) free(f->regs[i].name);
    for (int i = 0; i < f->nerrs; ++i) free(f->errs[i].desc);
    for (int i = 0; i < f->nctl; ++i) free(f->controls[i]);
    for (int i = 0; i < 6; ++i) { free(f->meta_key[i]); free(f->meta_val[i]); }
    free(f->regs); free(f->ins); free(f->errs); free(f->controls);
    free(f->itram); free(f->xtram); free(f->outbuf); free(f);
}

/* ---- isNumber: source/helpers.cpp:21-27  ^-?\d+(\.\d+)?$ ---- */
static int is_number(const char* s, size_t n) {
    size_t i = 0;
    if (i < n && s[i] == '-') ++i;
    size_t d0 = i;
    while (i < n && is_d(s[i])) ++i;
    if (i == d0) return 0;
    if (i == n) return 1;
    if (s[i] != '.') return 0;
    ++i;
    size_t f0 = i;
    while (i < n && is_d(s[i])) ++i;
    return i > f0 && i == n;
}

/* stof: strtof of the whole token; the reference throws on overflow → OOD here */
static float parse_float(fxo_t* f, const char* s, size_t n) {
    char* tmp = xstrdup_n(s, n);
    float v = strtof(tmp, NULL);
    if (isinf(v)) f->ood |= FXO_OOD_PARSE;
    free(tmp);
    return v;
}
/* stoi of a digit string (possibly "12.5" → 12); overflow → OOD */
static int parse_int(fxo_t* f, const char* s, size_t n) {
    long long v = 0; size_t i = 0;
    if (n == 0) { f->ood |= FXO_OOD_PARSE; return 0; }
    while (i < n && is_d(s[i])) { v = v * 10 + (s[i] - '0'); if (v > 2147483647LL) { f->ood |= FXO_OOD_PARSE; return 2147483647; } ++i; }
    return (int)v;
}

/* mapRegisterToIndex: source/FX8010.cpp:745-774 */
static int map_register(fxo_t* f, const char* s, size_t n) {
    int idx = find_reg(f, s, n);
    if (idx >= 0) return idx;
    if (is_number(s, n)) return push_reg(f, RT_STATIC, s, n, parse_float(f, s, n), 0);
    return -1;
}

static const char* const KW_DECL[] = { "static", "temp", "control", "input", "output", "const" };
static const int KW_DECL_T[] = { RT_STATIC, RT_TEMP, RT_CONTROL, RT_INPUT, RT_OUTPUT, RT_CONST };
static const char* const KW_OPS[] = { "macs", "macsn", "macints", "macintw", "acc3", "macmv", "macw", "macwn", "skip", "andxor",
                                      "tstneg", "limit", "limitn", "log", "exp", "interp", "idelay", "xdelay" };
static const int KW_OPS_C[] = { OP_MACS, OP_MACSN, OP_MACINTS, OP_MACINTW, OP_ACC3, OP_MACMV, OP_MACW, OP_MACWN, OP_SKIP, OP_ANDXOR,
                                OP_TSTNEG, OP_LIMIT, OP_LIMITN, OP_LOG, OP_EXP, OP_INTERP, OP_IDELAY, OP_XDELAY };
static const char* const KW_META[] = { "name", "copyright", "created", "engine", "comment", "guid" };

static int starts_with(const char* s, size_t n, size_t at, const char* kw) {
    size_t k = strlen(kw);
    return at + k <= n && memcmp(s + at, kw, k) == 0;
}
static size_t skip_s(const char* s, size_t n, size_t i) { while (i < n && is_s(s[i])) ++i; return i; }

/* tail of pattern1 after the name: (?:[\s=,]*\s*(\d+(?:\.\d+)?))?\s*$  (source/FX8010.cpp:371) */
static int decl_tail(const char* s, size_t n, size_t p, size_t* v0, size_t* v1) {
    size_t q = p;
    while (q < n && (is_s(s[q]) || s[q] == '=' || s[q] == ',')) ++q;
    if (q < n && is_d(s[q])) {
        size_t e = q;
        while (e < n && is_d(s[e])) ++e;
        if (e + 1 < n && s[e] == '.' && is_d(s[e + 1])) { e += 2; while (e < n && is_d(s[e])) ++e; }
        if (skip_s(s, n, e) == n) { *v0 = q; *v1 = e; return 1; }
    }
    *v0 = *v1 = 0;
    return skip_s(s, n, p) == n;
}

/* syntaxCheck: source/FX8010.cpp:365-741.  The seven std::regex patterns (:371-389)
 * are matched by hand; backtracking cases are noted inline. */
static void syntax_check(fxo_t* f, const char* s, size_t n) {
    const int row = f->error_counter;
    size_t i0 = skip_s(s, n, 0);

    /* pattern1 declaration (:371, handled :395-485) */
    for (int k = 0; k < 6; ++k) {
        if (!starts_with(s, n, i0, KW_DECL[k])) continue;
        size_t p = i0 + strlen(KW_DECL[k]);
        size_t j = skip_s(s, n, p);
        if (j == p) break;                       /* needs \s+ */
        size_t e = j;
        while (e < n && is_w(s[e])) ++e;
        if (e == j) break;                       /* needs \w+ */
        /* \w+ is greedy but gives characters back: "static a12.5" declares a1 = 2.5 */
        size_t v0 = 0, v1 = 0, L;
        int ok = 0;
        for (L = e - j; L >= 1; --L) { if (decl_tail(s, n, j + L, &v0, &v1)) { ok = 1; break; } }
        if (!ok) break;
        const char* name = s + j; size_t nlen = L;
        if (k == 2) {                            /* control list first (:408-411) */
            if (f->nctl == f->capctl) { f->capctl = f->capctl ? 2 * f->capctl : 8; f->controls = (char**)realloc(f->controls, sizeof(char*) * f->capctl); }
            f->controls[f->nctl++] = xstrdup_n(name, nlen);
        }
        if (find_reg(f, name, nlen) == -1) {     /* :414 */
            float value = 0.0f; int io = 0;
            if (v1 > v0) {
                if (k == 3 || k == 4) {          /* input/output: number is the channel (:443-460) */
                    int ch = parse_int(f, s + v0, v1 - v0);
                    if (ch > f->channels - 1) { char b[128]; err_io_range(f, b, sizeof b); push_error(f, b, row); return; }
                    io = ch;
                } else value = parse_float(f, s + v0, v1 - v0); /* :463 */
            }
            push_reg(f, KW_DECL_T[k], name, nlen, value, io);
        } else push_error(f, ERR_MULTI, row);    /* :475-483 */
        return;
    }

    /* pattern2 blank (:374) */
    if (i0 == n) return;

    /* pattern3 ^\s*(itramsize|xtramsize)\s+(\d+)*\s$ (:377, handled :498-543) */
    for (int k = 0; k < 2; ++k) {
        const char* kw = k == 0 ? "itramsize" : "xtramsize";
        if (!starts_with(s, n, i0, kw)) continue;
        size_t p = i0 + 9, j = skip_s(s, n, p);
        if (j == p) break;
        size_t e = j;
        while (e < n && is_d(s[e])) ++e;
        int match = 0; size_t d0 = j, d1 = e;
        if (e > j) match = (e + 1 == n && is_s(s[e]));            /* digits then exactly one \s */
        else { match = (j == n && j - p >= 2); d0 = d1 = 0; }     /* no digits: \s+ gives one back */
        if (!match) break;
        if (d1 == d0) { f->ood |= FXO_OOD_PARSE; push_error(f, ERR_SYNTAX, row); return; } /* reference: stoi("") throws */
        int v = parse_int(f, s + d0, d1 - d0);
        if (k == 0) { if (f->itram_size > MAX_IDELAY_SIZE) { push_error(f, ERR_ITRAM, row); return; } f->itram_size = v; } /* :506-521 tests the OLD size */
        else { if (f->xtram_size > MAX_XDELAY_SIZE) { push_error(f, ERR_XTRAM, row); return; } f->xtram_size = v; }          /* :525-540 */
        return;
    }

    /* pattern4 instruction (:380, handled :548-695) */
    for (int k = 0; k < 18; ++k) {
        if (!starts_with(s, n, i0, KW_OPS[k])) continue;
        size_t p = i0 + strlen(KW_OPS[k]), j = skip_s(s, n, p);
        if (j == p) continue;                    /* e.g. "macs" inside "macsn": try the longer keyword */
        size_t t0[4], t1[4]; int nf = 0, ok = 1; size_t q = j;
        while (ok) {
            size_t a = skip_s(s, n, q), b = a;
            if ((f->opts & FXO_OPT_TRAM_DANE) && b < n && s[b] == '&') ++b;       /* opt-in: &name = position register of a delay tap */
            while (b < n && (is_w(s[b]) || s[b] == '.' || s[b] == '-')) ++b;
            if (b == a || nf == 4) { ok = 0; break; }
            t0[nf] = a; t1[nf] = b; ++nf;
            size_t c = skip_s(s, n, b);
            if (c == n) break;
            if (s[c] != ',') { ok = 0; break; }
            q = c + 1;
        }
        if (!ok || nf != 4) break;
        if (nf > 0 && t0[0] != j) break;
        ins_t in; memset(&in, 0, sizeof in);
        in.op = KW_OPS_C[k];
        int idx[4];
        char tapname[256]; const char* tok[4]; size_t tlen[4];
        for (int o = 0; o < 4; ++o) { tok[o] = s + t0[o]; tlen[o] = t1[o] - t0[o]; }
        /* opt-in DANE taps (docs/TRAM Registermapping.pdf p.1): the position of "idelay read, rd, at, 17" lives in a register
         * of its own, "&rd" (created here with the literal as its value), which other instructions may write */
        if ((f->opts & FXO_OPT_TRAM_DANE) && (in.op == OP_IDELAY || in.op == OP_XDELAY) && is_number(tok[3], tlen[3]) && tlen[1] + 2 < sizeof tapname
            && tok[1][0] != '&') {
            tapname[0] = '&'; memcpy(tapname + 1, tok[1], tlen[1]); tapname[tlen[1] + 1] = 0;
            if (find_reg(f, tapname, tlen[1] + 1) < 0) {
                float pos = parse_float(f, tok[3], tlen[3]);                 /* samples; as a DANE address: 0x800 per sample of 2^31 */
                if (f->opts & FXO_OPT_TRAM_ADDR_SHIFT) pos = pos * 9.5367431640625e-07f;
                push_reg(f, RT_STATIC, tapname, tlen[1] + 1, pos, 0);
            }
            tok[3] = tapname; tlen[3] = tlen[1] + 1;
        }
        for (int o = 0; o < 4; ++o) {            /* R (:574), A (:608), X (:637), Y (:666), early return on the first failure */
            idx[o] = map_register(f, tok[o], tlen[o]);
            if (idx[o] == -1) { push_error(f, ERR_UNDECL, row); return; }
            gpr_t* g = &f->regs[idx[o]];
            if (o == 0) {
                if (g->type == RT_INPUT) { push_error(f, ERR_R_INPUT, row); return; }  /* :587-595 */
                if (g->type == RT_OUTPUT) in.has_out = 1;                              /* :596-600 */
            } else {
                if (g->type == RT_INPUT) in.has_in = 1;                                /* :621-625 */
                else if (strcmp(g->name, "noise") == 0) in.has_noise = 1;              /* :626-629 */
            }
        }
        in.r = idx[0]; in.a = idx[1]; in.x = idx[2]; in.y = idx[3];
        push_ins(f, in);                         /* :693 */
        return;
    }

    /* pattern5 metadata \s*(key)\s+\"([^\"]+)\" — full match, nothing after the quote (:383, :699-708) */
    for (int k = 0; k < 6; ++k) {
        if (!starts_with(s, n, i0, KW_META[k])) continue;
        size_t p = i0 + strlen(KW_META[k]), j = skip_s(s, n, p);
        if (j == p || j >= n || s[j] != '"') continue;
        size_t e = j + 1;
        while (e < n && s[e] != '"') ++e;
        if (e == j + 1 || e != n - 1) continue;
        free(f->meta_val[k]); free(f->meta_key[k]);
        f->meta_key[k] = xstrdup_n(KW_META[k], strlen(KW_META[k]));
        f->meta_val[k] = xstrdup_n(s + j + 1, e - j - 1);
        return;
    }

    /* pattern6 ^\s*(end)\s*$ (:386, :712-718): END with all operands 0 */
    if (starts_with(s, n, i0, "end") && skip_s(s, n, i0 + 3) == n) {
        ins_t in; memset(&in, 0, sizeof in); in.op = OP_END; push_ins(f, in); return;
    }
    /* pattern7 (comment) can never match: ';' was cut in the pre-pass.  else: (:731-739) */
    push_error(f, ERR_SYNTAX, row);
}

/* loadFile body after the file has been read: source/FX8010.cpp:790-875 */
static int load_lines(fxo_t* f, const char* text, size_t len) {
    size_t pos = 0; char* last = NULL; int nlines = 0;
    while (pos < len) {                          /* getline loop :790 */
        size_t e = pos;
        while (e < len && text[e] != '\n') ++e;
        size_t n = e - pos;
        const char* semi = (const char*)memchr(text + pos, ';', n);   /* :794-798 */
        if (semi) n = (size_t)(semi - (text + pos));
        char* line = xstrdup_n(text + pos, n);
        for (size_t i = 0; i < n; ++i) if (line[i] >= 'A' && line[i] <= 'Z') line[i] = (char)(line[i] - 'A' + 'a'); /* :805-808 */
        syntax_check(f, line, n);                /* :819-825 (the reference collects first, checks after; same order) */
        f->error_counter++;
        free(last); last = line; ++nlines;
        pos = e + 1;
    }
    if (nlines == 0 || strcmp(last, "end") != 0) push_error(f, ERR_NO_END, f->error_counter); /* :829-838 */
    free(last);
    if (f->nerrs > 1) return 0;                  /* :844-858 */
    f->ready = 1;                                /* :865 */
    return 1;
}

int fxo_load_text(fxo_t* f, const char* text) { return load_lines(f, text, strlen(text)); }

int fxo_load_file(fxo_t* f, const char* path) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return 0;                           /* :868-873 no error entry */
    fseek(fp, 0, SEEK_END); long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    char* buf = (char*)malloc((size_t)sz + 1);
    size_t got = fread(buf, 1, (size_t)sz, fp); fclose(fp);
    int ok = load_lines(f, buf, got);
    free(buf);
    return ok;
}

/* ------------------------------------------------------------ hot path */

/* The arithmetic of the hot path, operand order included.  The reference is an x86-64 SSE2 build (g++ -O2): every fp32 / fp64
 * operation is ONE two-operand SSE instruction `op first, second`, and what that instruction does with NaNs is part of the
 * reference's observable behaviour (tests/golden/nonfinite.json, nan_collisions.json):
 *   - a NaN in `first` is handed on (quieted, sign and payload kept); otherwise a NaN in `second` is;
 *   - an invalid operation (Inf - Inf, 0 * Inf) makes the negative default NaN 0xFFC00000 / 0xFFF8000000000000;
 *   - cvtss2sd / cvtsd2ss quiet a NaN and keep sign and the payload's top bits.
 * Which C operand g++ made `first` is pinned per expression by nan_collisions.json (generated from the compiled reference) and
 * written out at each call below.  Plain C `a + b` leaves that choice to the compiler (gcc -O0, clang and sanitizer builds pick
 * differently), so the checker does not use it where a NaN can arrive. */
static inline uint32_t f32_bits(float v) { uint32_t u; memcpy(&u, &v, 4); return u; }
static inline float bits_f32(uint32_t u) { float v; memcpy(&v, &u, 4); return v; }
static inline uint64_t f64_bits(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }
static inline double bits_f64(uint64_t u) { double v; memcpy(&v, &u, 8); return v; }
static inline int nan32(float v) { return (f32_bits(v) & 0x7fffffffu) > 0x7f800000u; }
static inline int nan64(double v) { return (f64_bits(v) & 0x7fffffffffffffffull) > 0x7ff0000000000000ull; }
static inline float sse_pick32(float first, float second, float plain) {
    if (nan32(first)) return bits_f32(f32_bits(first) | 0x00400000u);
    if (nan32(second)) return bits_f32(f32_bits(second) | 0x00400000u);
    if (nan32(plain)) return bits_f32(0xffc00000u);
    return plain;
}
static inline double sse_pick64(double first, double second, double plain) {
    if (nan64(first)) return bits_f64(f64_bits(first) | 0x0008000000000000ull);
    if (nan64(second)) return bits_f64(f64_bits(second) | 0x0008000000000000ull);
    if (nan64(plain)) return bits_f64(0xfff8000000000000ull);
    return plain;
}
static inline float addss(float first, float second) { return sse_pick32(first, second, first + second); }
static inline float subss(float first, float second) { return sse_pick32(first, second, first - second); }
static inline float mulss(float first, float second) { return sse_pick32(first, second, first * second); }
static inline double addsd(double first, double second) { return sse_pick64(first, second, first + second); }
static inline double subsd(double first, double second) { return sse_pick64(first, second, first - second); }
static inline double mulsd(double first, double second) { return sse_pick64(first, second, first * second); }
static inline double cvtss2sd(float v) {
    if (!nan32(v)) return (double)v;
    const uint32_t u = f32_bits(v);
    return bits_f64(((uint64_t)(u >> 31) << 63) | 0x7ff8000000000000ull | ((uint64_t)(u & 0x003fffffu) << 29));
}
static inline float cvtsd2ss(double v) {
    if (!nan64(v)) return (float)v;
    const uint64_t u = f64_bits(v);
    return bits_f32((uint32_t)(u >> 63) << 31 | 0x7fc00000u | (uint32_t)((u >> 29) & 0x003fffffu));
}

/* x86 cvttss2si / cvttsd2si: out-of-range and NaN give the "integer indefinite"
 * 0x80000000 — what every static_cast<int>(float) in the reference compiles to. */
static inline int32_t cvtt_f32(float v) { if (!(v < 2147483648.0f) || v < -2147483648.0f) return INT32_MIN; return (int32_t)v; }
static inline int32_t cvtt_f64(double v) { if (!(v < 2147483648.0) || v <= -2147483649.0) return INT32_MIN; return (int32_t)v; }

/* setCCR: source/FX8010.cpp:211-232 */
static inline void set_ccr(fxo_t* f, float r) {
    float c;
    if (r == 0) c = 8.0f;
    else if (r < 0 && r > -1.0) c = 6.0f;
    else if (r > 0 && r < 1.0) c = 2.0f;
    else if (r == 1.0) c = 16.0f;
    else if (r == -1.0) c = 20.0f;
    else c = 0.0f;
    f->regs[0].value = c;
}
/* saturate: source/FX8010.cpp:275-279 */
static inline float saturate(float v, float t) { return (v >= t) ? t : ((v <= -t) ? -t : v); }
/* intToFloat / floatToInt: source/FX8010.cpp:1009-1020; (float)INT32_MAX == 2^31 */
static inline float int_to_float(int32_t i) { return (float)i / 2147483648.0f; }
static inline int32_t float_to_int(float v) { return cvtt_f32(v * 2147483648.0f); }

/* wrapAround: source/FX8010.cpp:299-328 (also rewrites CCR, which setCCR then overwrites) */
static inline float wrap_around(fxo_t* f, float a) {
    float result;
    int32_t ccr_ = float_to_int(f->regs[0].value);
    if (a >= 1.0f) { result = a - 2.0f; ccr_ = 1 | ccr_; }
    else if (a < -1.0f) { result = a + 2.0f; ccr_ = 1 | ccr_; }
    else { result = a; ccr_ = ccr_ & ~1; }
    f->regs[0].value = int_to_float(ccr_);
    return result;
}

/* logicOps: source/FX8010.cpp:330-360 */
static inline int32_t logic_ops(float a_, float x_, float y_) {
    const int32_t A = cvtt_f32(a_), X = cvtt_f32(x_), Y = cvtt_f32(y_);
    if (Y == 0) return A & X;
    else if (X == 0xFFFFFF) return A ^ Y;
    else if (X == 0xFFFFFFF && Y == 0xFFFFFF) return ~A;
    else if (Y == ~X) return A | Y;
    else if (Y == 0xFFFFFF) return ~A & X;
    return (A & X) ^ Y;
}

/* linearInterpolate: source/FX8010.cpp:283-296, x_min=-1.0, x_max=1.0, table size 64 */
static inline double linear_interpolate(fxo_t* f, double x, const double* tbl) {
    const double x_min = -1.0, x_max = 1.0;
    double step = (x_max - x_min) / (double)(64 - 1);
    /* In the domain (x in [-1, 1]) the quotient lies in [0, 63] and this is the reference's (int) cast.  Outside it the
       reference indexes its vector out of bounds; here the quotient is clamped to 0 .. 63 as a number - a NaN counts as
       below (the cast would give INT_MIN), and so does nothing else: +Inf or 1e10 clamp to 63, they do not wrap to INT_MIN. */
    const double q = (x - x_min) / step;
    int index = !(q >= 0.0) ? (q == q && q > -1.0 ? 0 : -1) : (q >= 64.0 ? 64 : cvtt_f64(q));
    if (index < 0 || index > 63) { f->ood |= FXO_OOD_LUT_INDEX; index = index < 0 ? 0 : 63; }
    double x1 = x_min + index * step;
    double x2 = x_min + (index + 1) * step;
    double y1 = tbl[index];
    double y2 = tbl[index + 1];                  /* index 63 reads the pad (reference: out of bounds) */
    double y = addsd(mulsd((y2 - y1) / (x2 - x1), subsd(x, x1)), y1);   /* tables and knots are finite: x is the only NaN source */
    return y;
}
static inline const double* lut_row(fxo_t* f, int kind, float xsel) {
    int32_t t = cvtt_f32(xsel);
    if (t < 0 || t > 31) { f->ood |= FXO_OOD_LUT_TABLE; t = t < 0 ? 0 : 31; }
    return kind ? f->lut_exp[t] : f->lut_log[t];
}

/* TRAM engine: source/FX8010.cpp:909-967 */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* opt-in DANE delay-line model (NOT reference behaviour; docs/TRAM Registermapping.pdf, kX/DANE convention): one address
 * counter per TRAM that steps DOWN once per sample period; every tap addresses (counter + position) mod size, so a value
 * written at position pw is read pr - pw samples later at position pr.  Positions are whole samples, or - with
 * FXO_OPT_TRAM_ADDR_SHIFT - DANE addresses of 0x800 per sample. */
static int dane_slot(fxo_t* f, int size, int base, float value) {
    /* a position register holds whole samples, or (ADDR_SHIFT) a DANE address as the fixed-point fraction the FX8010 keeps in
     * its registers: address = value * 2^31 (the reference's floatToInt, FX8010.cpp:1016-1020), 0x800 of them per sample */
    int position = (f->opts & FXO_OPT_TRAM_ADDR_SHIFT) ? (cvtt_f32(value * 2147483648.0f) >> 11) : cvtt_f32(value);
    long idx = ((long)base + position) % size;
    return (int)(idx < 0 ? idx + size : idx);
}
/* opt-in FXO_OPT_TRAM_INTERP (with ADDR_SHIFT; the reference only says what it wants: "To do linear Interpolation, you need to
 * find the fractional part of the read position and interpolate between the two adjacent samples", source/FX8010.cpp:929-932):
 * a READ tap at DANE address a = floatToInt(value) takes x0 from position a >> 11 and x1 from the position after it and
 * returns x0 + f * (x1 - x0), f = (a & 0x7ff) / 2048 - four fp32 operations in this order, nothing fused; f == 0: x0 itself. */
static float dane_read(fxo_t* f, const float* ring, int size, int base, float value) {
    if (!((f->opts & FXO_OPT_TRAM_INTERP) && (f->opts & FXO_OPT_TRAM_ADDR_SHIFT))) return ring[dane_slot(f, size, base, value)];
    const int32_t a = cvtt_f32(value * 2147483648.0f);
    const int32_t p = a >> 11, frac = a & 0x7ff;
    long i0 = ((long)base + p) % size, i1 = ((long)base + p + 1) % size;
    if (i0 < 0) i0 += size;
    if (i1 < 0) i1 += size;
    const float x0 = ring[i0];
    if (frac == 0) return x0;
    const float x1 = ring[i1], fr = (float)frac * 0.00048828125f;
    const float d = addss(x1, mulss(-1.0f, x0));      /* x1 - x0 without negating a NaN (the device cannot subtract one unchanged) */
    return addss(x0, mulss(fr, d));
}
static void tram_write(fxo_t* f, float* buf, int cap, int size, int* wpos, float sample, int position) {
    if (size <= 0) { f->ood |= FXO_OOD_TRAM_SIZE0; return; }
    position = position > size - 1 ? size - 1 : position; position = position < 0 ? 0 : position; /* max(0,min(p,size-1)) */
    long idx = (long)*wpos + position;           /* no modulo (:914) */
    if (idx >= cap) f->ood |= FXO_OOD_TRAM_WRITE_OOB; else buf[idx] = sample;
    *wpos = (*wpos + 1) % size;
}
static float tram_read(fxo_t* f, const float* buf, int cap, int size, int* rpos, int position) {
    if (size <= 0) { f->ood |= FXO_OOD_TRAM_SIZE0; return 0.0f; }
    position = position > size - 1 ? size - 1 : position; position = position < 0 ? 0 : position;
    int idx = (*rpos - position) % size;         /* C remainder (:952): negative → before the array in the reference */
    if (idx < 0) { f->ood |= FXO_OOD_TRAM_READ_NEG; idx += size; }
    float out = idx < cap ? buf[idx] : 0.0f;
    *rpos = (*rpos + 1) % size;
    return out;
}

/* whitenoise: source/FX8010.cpp:993-1000; g_fScale = 2.0f/0xffffffff == 2^-31 (FX8010.h:288) */
static inline float whitenoise(fxo_t* f) {
    const float g_fScale = 2.0f / 4294967296.0f;
    f->g_x1 ^= f->g_x2;
    float noise = (float)f->g_x2 * g_fScale;
    f->g_x2 = (int32_t)((uint32_t)f->g_x2 + (uint32_t)f->g_x1);
    return noise;
}

/* process: source/FX8010.cpp:1023-1249 */
void fxo_process(fxo_t* f, const float* in, float* out) {
    int is_end = 0, num_skip = 0, passes = 0;
    gpr_t* regs = f->regs;
    if (f->nins == 0) { for (int c = 0; c < f->channels; ++c) out[c] = f->outbuf[c]; return; } /* reference: endless loop */
    do {
        for (int pc = 0; pc < f->nins; ++pc) {
            const ins_t* I = &f->ins[pc];
            if (num_skip == 0) {                                             /* :1037 */
                gpr_t* R = &regs[I->r]; gpr_t* A = &regs[I->a]; gpr_t* X = &regs[I->x]; gpr_t* Y = &regs[I->y]; /* :1047-1050 */
                if (I->has_in) {                                             /* :1053-1061: A's IOIndex for all three */
                    if (A->type == RT_INPUT) A->value = in[A->io];
                    if (X->type == RT_INPUT) X->value = in[A->io];
                    if (Y->type == RT_INPUT) Y->value = in[A->io];
                }
                if (I->has_noise) {                                          /* :1063-1071 */
                    if (strcmp(A->name, "noise") == 0) A->value = whitenoise(f);
                    else if (strcmp(X->name, "noise") == 0) X->value = whitenoise(f);
                    else if (strcmp(Y->name, "noise") == 0) Y->value = whitenoise(f);
                }
                switch (I->op) {
                case OP_MACS: case OP_MACINTS: {                             /* :1077-1085, :1095-1103 */
                    float p = mulss(X->value, Y->value); float t = addss(p, A->value);   /* addss product, A: NaN order X, Y, A */
                    f->acc = t; R->value = saturate(t, 1.0f); set_ccr(f, R->value); break; }
                case OP_MACSN: {                                             /* :1086-1094 */
                    float p = mulss(X->value, Y->value); float t = subss(A->value, p);   /* subss A, product: A, X, Y */
                    f->acc = t; R->value = saturate(t, 1.0f); set_ccr(f, R->value); break; }
                case OP_ACC3: {                                              /* :1104-1112 */
                    float t = addss(A->value, X->value); t = addss(t, Y->value);           /* A, X, Y */
                    f->acc = t; R->value = saturate(t, 1.0f); set_ccr(f, R->value); break; }
                case OP_LOG: {                                               /* :1113-1119 */
                    float r = cvtsd2ss(linear_interpolate(f, cvtss2sd(A->value), lut_row(f, 0, X->value)));
                    R->value = r; f->acc = r; set_ccr(f, r); break; }
                case OP_EXP: {                                               /* :1120-1125 */
                    float r = cvtsd2ss(linear_interpolate(f, cvtss2sd(A->value), lut_row(f, 1, X->value)));
                    R->value = r; f->acc = r; set_ccr(f, r); break; }
                case OP_MACW: {                                              /* :1126-1131; g++ -O2 reads A before calling wrapAround */
                    float a = A->value; float p = mulss(X->value, Y->value); float w = wrap_around(f, p);
                    float r = addss(a, w);                                                  /* A, X, Y */ R->value = r; f->acc = r; set_ccr(f, r); break; }
                case OP_MACWN: {                                             /* :1132-1137 */
                    float a = A->value; float p = mulss(X->value, Y->value); float w = wrap_around(f, p);
                    float r = subss(a, w);                                                  /* A, X, Y */ R->value = r; f->acc = r; set_ccr(f, r); break; }
                case OP_MACINTW: {                                           /* :1138-1143 */
                    float p = mulss(X->value, Y->value); float t = addss(p, A->value); float r = wrap_around(f, t); /* X, Y, A */
                    R->value = r; f->acc = r; set_ccr(f, r); break; }
                case OP_MACMV: {                                             /* :1144-1149 */
                    float p = mulss(X->value, Y->value); f->acc = addsd(f->acc, cvtss2sd(p)); /* acc is never observed */
                    R->value = A->value; set_ccr(f, R->value); break; }
                case OP_ANDXOR: {                                            /* :1150-1154 */
                    R->value = (float)logic_ops(A->value, X->value, Y->value); set_ccr(f, R->value); break; }
                case OP_TSTNEG: {                                            /* :1155-1162 */
                    float r = A->value >= Y->value ? X->value : int_to_float(~float_to_int(X->value));
                    R->value = r; f->acc = r; set_ccr(f, r); break; }
                case OP_LIMIT: {                                             /* :1163-1168 */
                    float r = A->value >= Y->value ? X->value : Y->value;
                    R->value = r; f->acc = r; set_ccr(f, r); break; }
                case OP_LIMITN: {                                            /* :1169-1174 */
                    float r = A->value < Y->value ? X->value : Y->value;
                    R->value = r; f->acc = r; set_ccr(f, r); break; }
                case OP_SKIP:                                                /* :1175-1179 */
                    if ((float)cvtt_f32(X->value) == regs[0].value) num_skip = cvtt_f32(Y->value);
                    break;
                case OP_INTERP: {                                            /* :1180-1187 */
                    float p = mulss(X->value, Y->value);                                    /* NaN order X, A, Y */
                    double d = addsd(mulsd(subsd(1.0, cvtss2sd(X->value)), cvtss2sd(A->value)), cvtss2sd(p));
                    float r = cvtsd2ss(d); f->acc = r; R->value = saturate(r, 1.0f); set_ccr(f, R->value); break; }
                case OP_IDELAY:                                              /* :1188-1199 */
                    if ((f->opts & FXO_OPT_TRAM_DANE) && (R->type == RT_READ || R->type == RT_WRITE)) {
                        if (f->itram_size <= 0) { f->ood |= FXO_OOD_TRAM_SIZE0; if (R->type == RT_READ) A->value = 0.0f; break; }
                        if (R->type == RT_READ) A->value = dane_read(f, f->itram, f->itram_size, f->iw, Y->value);
                        else f->itram[dane_slot(f, f->itram_size, f->iw, Y->value)] = A->value;
                        break;
                    }
                    if (R->type == RT_READ) A->value = tram_read(f, f->itram, MAX_IDELAY_SIZE, f->itram_size, &f->ir, cvtt_f32(Y->value));
                    else if (R->type == RT_WRITE) tram_write(f, f->itram, MAX_IDELAY_SIZE, f->itram_size, &f->iw, A->value, cvtt_f32(Y->value));
                    break;
                case OP_XDELAY:                                              /* :1200-1211 */
                    if (R->type == RT_READ || R->type == RT_WRITE) {
                        if (!f->xtram) f->xtram = (float*)calloc(MAX_XDELAY_SIZE, sizeof(float));
                        if (f->opts & FXO_OPT_TRAM_DANE) {
                            if (f->xtram_size <= 0) { f->ood |= FXO_OOD_TRAM_SIZE0; if (R->type == RT_READ) A->value = 0.0f; break; }
                            if (R->type == RT_READ) A->value = dane_read(f, f->xtram, f->xtram_size, f->xw, Y->value);
                            else f->xtram[dane_slot(f, f->xtram_size, f->xw, Y->value)] = A->value;
                            break;
                        }
                        if (R->type == RT_READ) A->value = tram_read(f, f->xtram, MAX_XDELAY_SIZE, f->xtram_size, &f->xr, cvtt_f32(Y->value));
                        else tram_write(f, f->xtram, MAX_XDELAY_SIZE, f->xtram_size, &f->xw, A->value, cvtt_f32(Y->value));
                    }
                    break;
                case OP_END: is_end = 1; break;                              /* :1212-1215 */
                default: break;
                }
                f->icount++;                                                 /* :1222 */
                if (R->type == RT_OUTPUT) f->outbuf[R->io] = R->value;       /* :1229-1233 */
            } else {
                num_skip = (num_skip > 0) ? num_skip - 1 : 0;                /* :1238 */
            }
        }
        if (!is_end && ++passes >= PASS_CAP) { f->ood |= FXO_OOD_PASS_CAP; break; }
    } while (!is_end);                                                       /* :1243 */
    if (f->opts & FXO_OPT_TRAM_DANE) {           /* opt-in: the address counters step once per sample period */
        if (f->itram_size > 0) f->iw = (f->iw + f->itram_size - 1) % f->itram_size;
        if (f->xtram_size > 0) f->xw = (f->xw + f->xtram_size - 1) % f->xtram_size;
    }
    for (int c = 0; c < f->channels; ++c) out[c] = f->outbuf[c];             /* :1248 */
}

void fxo_process_block(fxo_t* f, const float* in, float* out, int S) {
    for (int s = 0; s < S; ++s) fxo_process(f, in + (size_t)s * f->channels, out + (size_t)s * f->channels);
}

/* ------------------------------------------------------ control / introspection */
int fxo_set_register(fxo_t* f, const char* key, float v) {     /* source/FX8010.cpp:236-253 */
    int i = find_reg(f, key, strlen(key));
    if (i < 0) return 1;
    f->regs[i].value = v; return 0;
}
float fxo_get_register(fxo_t* f, const char* key) {            /* :256-266 */
    int i = find_reg(f, key, strlen(key));
    return i < 0 ? 1.0f : f->regs[i].value;
}
int64_t fxo_instruction_counter(fxo_t* f) { return f->icount; } /* :986-989 */
int fxo_ready(fxo_t* f) { return f->ready; }
int fxo_channels(fxo_t* f) { return f->channels; }
unsigned fxo_ood_flags(fxo_t* f) { return f->ood; }
void fxo_set_option(fxo_t* f, unsigned option, int on) { if (on) f->opts |= option; else f->opts &= ~option; }
void fxo_seed_noise(fxo_t* f, int32_t x1, int32_t x2) { f->g_x1 = x1; f->g_x2 = x2; }
/* the members a test cannot reach through outputs alone (FX8010.h:210-217, 290-291): delay memory, cursors, LFSR words */
int fxo_tram(fxo_t* f, int which, float* out, int n) {
    const float* buf = which ? f->xtram : f->itram;
    const int have = which ? MAX_XDELAY_SIZE : MAX_IDELAY_SIZE;
    if (n > have) n = have;
    for (int k = 0; k < n; ++k) out[k] = buf ? buf[k] : 0.0f;   /* (xTRAM is allocated on first use: all zero until then) */
    return n;
}
void fxo_cursors(fxo_t* f, int out4[4]) { out4[0] = f->iw; out4[1] = f->ir; out4[2] = f->xw; out4[3] = f->xr; }
void fxo_lfsr(fxo_t* f, int32_t out2[2]) { out2[0] = f->g_x1; out2[1] = f->g_x2; }
int fxo_error_count(fxo_t* f) { return f->nerrs; }
const char* fxo_error_desc(fxo_t* f, int i) { return (i < 0 || i >= f->nerrs) ? "" : f->errs[i].desc; }
int fxo_error_row(fxo_t* f, int i) { return (i < 0 || i >= f->nerrs) ? -1 : f->errs[i].row; }
int fxo_control_count(fxo_t* f) { return f->nctl; }
const char* fxo_control_at(fxo_t* f, int i) { return (i < 0 || i >= f->nctl) ? "" : f->controls[i]; }
int fxo_meta_get(fxo_t* f, const char* key, char* buf, int buflen) {
    for (int k = 0; k < 6; ++k)
        if (f->meta_key[k] && strcmp(f->meta_key[k], key) == 0) { snprintf(buf, (size_t)buflen, "%s", f->meta_val[k]); return 1; }
    return 0;
}
int fxo_num_registers(fxo_t* f) { return f->nregs; }
const char* fxo_register_name(fxo_t* f, int i) { return f->regs[i].name; }
int fxo_register_type(fxo_t* f, int i) { return f->regs[i].type; }
int fxo_register_ioindex(fxo_t* f, int i) { return f->regs[i].io; }
float fxo_register_value(fxo_t* f, int i) { return f->regs[i].value; }
int fxo_num_instructions(fxo_t* f) { return f->nins; }
void fxo_instruction(fxo_t* f, int i, int o[8]) {
    const ins_t* I = &f->ins[i];
    o[0] = I->op; o[1] = I->r; o[2] = I->a; o[3] = I->x; o[4] = I->y; o[5] = I->has_in; o[6] = I->has_out; o[7] = I->has_noise;
}
int fxo_itram_size(fxo_t* f) { return f->itram_size; }
int fxo_xtram_size(fxo_t* f) { return f->xtram_size; }
const double* fxo_lut(fxo_t* f, int kind, int e) { return kind ? f->lut_exp[e & 31] : f->lut_log[e & 31]; }

/* ------------------------------------------------------ CPU baseline ("port") */
typedef struct { fxo_t* f; long samples; const float* in; int in_len; int t; double sum; volatile int* go; } bench_arg_t;
static void* bench_worker(void* p) {
    bench_arg_t* a = (bench_arg_t*)p;
    while (!*a->go) {}
    float vin[1], vout[8]; double acc = 0.0;
    for (long s = 0; s < a->samples; ++s) { vin[0] = a->in[(s + 17 * a->t) % a->in_len]; fxo_process(a->f, vin, vout); acc += vout[0]; }
    a->sum = acc;
    return NULL;
}
double fxo_bench(const char* path, long samples, int threads, const float* in, int in_len, long long* instr_out, double* checksum) {
    bench_arg_t* args = (bench_arg_t*)calloc((size_t)threads, sizeof(bench_arg_t));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    volatile int go = 0;
    for (int t = 0; t < threads; ++t) {
        args[t].f = fxo_create(1);
        if (!fxo_load_file(args[t].f, path)) { for (int k = 0; k <= t; ++k) fxo_destroy(args[k].f); free(args); free(th); return -1.0; }
        args[t].samples = samples; args[t].in = in; args[t].in_len = in_len; args[t].t = t; args[t].go = &go;
    }
    for (int t = 1; t < threads; ++t) pthread_create(&th[t], NULL, bench_worker, &args[t]);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    go = 1;
    bench_worker(&args[0]);
    for (int t = 1; t < threads; ++t) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    long long instr = 0; double cs = 0.0;
    for (int t = 0; t < threads; ++t) { instr += args[t].f->icount; cs += args[t].sum; fxo_destroy(args[t].f); }
    if (instr_out) *instr_out = instr;
    if (checksum) *checksum = cs;
    free(args); free(th);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
