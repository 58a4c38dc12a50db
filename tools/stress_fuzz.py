"""Fuzz at scale: random in-domain programs (all opcodes, SKIPs, TRAM, noise, LOG/EXP) on many wavefronts per
SIMD, sampled instances checked bit-for-bit against the oracle.  Exercises the interplay of handlers
(index-mode switches, fp64 ops, EXEC predication) under full occupancy, which small parity tests cannot.

    python tools/stress_fuzz.py [programs] [instances]
"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fx8010-emulator-core_amd/python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402

import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402
from pyoracle import Oracle  # noqa: E402

OPS3 = ["macs", "macsn", "macints", "acc3", "macw", "macwn", "macintw", "macmv", "tstneg", "limit", "limitn", "interp", "andxor",
        "macs", "macs", "interp", "interp", "macsn", "acc3"]


def random_program(rng, n_instr, n_regs):
    regs = ["r%d" % i for i in range(n_regs)]
    lits = ["0", "0.5", "-0.25", "1.0", "0.125", "2", "-1", "0.999", "3", "7", "15"]
    L = ["input in 0", "output out 0", "control c = 0.3", "static noise", "itramsize 11 ", "xtramsize 23 "] + ["static %s" % r for r in regs]
    L.append("idelay read, r1, at, 0")
    L.append("xdelay read, r0, at, 0")
    for i in range(n_instr):
        kind = rng.integers(0, 100)
        src = lambda: str(rng.choice(regs + regs + lits + ["in", "c", "out", "ccr", "noise"]))
        dst = str(rng.choice(regs + ["out"]))
        if kind < 72:
            L.append("%s %s, %s, %s, %s" % (rng.choice(OPS3), dst, src(), src(), src()))
        elif kind < 82:
            L.append("%s %s, %s, %d, 0" % (rng.choice(["log", "exp"]), dst, str(rng.choice(["in", "c", "0.5", "-0.25"])), rng.integers(0, 32)))
        elif kind < 92 and i + 4 < n_instr:
            L.append("skip ccr, ccr, %s, %d" % (rng.choice(["0", "2", "6", "8", "16", "20"]), rng.integers(0, 3)))
        else:
            L.append("macs %s, %s, %s, %s" % (dst, src(), src(), src()))
    L += ["idelay write, r0, at, 0", "xdelay write, r1, at, 0", "macs out, out, r2, 0.5", "end"]
    return "\n".join(L)


def product_cache_program(rng, length):
    """Random program over a tiny vocabulary - three state registers, two coefficients, the input - so that the same
    (coefficient x register) product turns up again and again: with the register unchanged in between (the translated tier
    reuses the product), rewritten in between, rewritten inside a SKIP shadow, or first computed inside one."""
    regs = ["a", "b", "c"]
    coef = ["0.3", "0.75", "cut"]
    lines = ["static a = 0.1", "static b = 0.2", "static c", "static t", "input in 0", "output out 0", "control cut = 0.4"]
    body = []
    for _ in range(length):
        kind = rng.integers(0, 10)
        r = regs[rng.integers(0, 3)]
        y = regs[rng.integers(0, 3)] if rng.integers(0, 4) else ("in", "ccr")[rng.integers(0, 2)]  # (the CCR changes on the side)
        k = coef[rng.integers(0, 3)]
        if kind < 4:
            body.append("%s %s, %s, %s, %s" % (("macs", "macsn")[rng.integers(0, 2)], r, regs[rng.integers(0, 3)], y, k))
        elif kind < 8:
            body.append("interp %s, %s, %s, %s" % (r, regs[rng.integers(0, 3)], k, y))
        elif kind == 8:
            body.append("macs t, %s, 0, 0" % r)
            body.append("skip ccr, ccr, %d, %d" % ((6, 2, 8)[rng.integers(0, 3)], rng.integers(1, 3)))
        else:
            body.append("macs %s, %s, %s, %s" % (r, k, y, regs[rng.integers(0, 3)]))   # a product of two rows in between
    # pad so that no SKIP reaches the end, then mix everything into the output
    body += ["macs out, a, b, 0.5", "macs out, out, c, 0.5", "macs out, out, in, 0.1"]
    return "\n".join(lines + body + ["end"])


def random_program2(rng, n_instr, n_regs):
    """a second flavour: TRAM instructions anywhere (also inside SKIP shadows), longer and negative skip counts,
    LOG/EXP of registers, more `ccr` traffic"""
    regs = ["r%d" % i for i in range(n_regs)]
    lits = ["0", "0.5", "-0.25", "1.0", "0.125", "-1", "0.999", "0.0625", "2"]
    L = ["input in 0", "output out 0", "control c = 0.3", "static noise", "static rd", "static xd",
         "itramsize %d " % rng.integers(1, 14), "xtramsize %d " % rng.integers(1, 40)] + ["static %s" % r for r in regs]
    sat_ops = ["macs", "macsn", "acc3", "interp", "macints"]
    for i in range(n_instr):
        kind = rng.integers(0, 100)
        src = lambda: str(rng.choice(regs + regs + regs + lits + ["in", "c", "out", "ccr", "noise", "rd", "xd"]))
        dst = str(rng.choice(regs + ["out"]))
        if kind < 55:
            L.append("%s %s, %s, %s, %s" % (rng.choice(sat_ops), dst, src(), src(), src()))
        elif kind < 63:
            L.append("%s %s, %s, %s, %s" % (rng.choice(["macw", "macwn", "macintw", "macmv", "tstneg", "limit", "limitn", "andxor"]), dst, src(), src(), src()))
        elif kind < 72:
            L.append("%s %s, %s, %d, 0" % (rng.choice(["log", "exp"]), dst, str(rng.choice(regs[:2] + ["in", "c"])), rng.integers(0, 32)))
        elif kind < 84 and i + 6 < n_instr:
            L.append("skip ccr, ccr, %s, %d" % (rng.choice(["0", "2", "6", "8", "16", "20", "3"]), rng.integers(-1, 5)))
        elif kind < 88:
            L.append("idelay read, rd, at, 0")
        elif kind < 92:
            L.append("idelay write, %s, at, %d" % (str(rng.choice(regs + ["in"])), rng.integers(0, 2)))
        elif kind < 96:
            L.append("xdelay read, xd, at, 0")
        else:
            L.append("xdelay write, %s, at, 0" % str(rng.choice(regs + ["in"])))
    L += ["macs %s, %s, %s, %s" % (regs[0], regs[0], regs[-1], "0.5"), "macs out, out, %s, 0.5" % regs[0], "end"]
    return "\n".join(L)


def main():
    n_prog = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
    S = 8
    x = P.stimulus(N, S) * np.float32(float(os.environ.get("FX_FUZZ_SCALE", "1")))   # > 1: values leave their classes, waves change streams
    if os.environ.get("FX_FUZZ_NAN"):
        words = np.array([0x7FC00000, 0xFFC00000, 0x7FC12345, 0x7F800000, 0xFF800000], dtype=np.uint32).view(np.float32)
        r = np.random.default_rng(98)
        hit = r.random(x.shape) < float(os.environ["FX_FUZZ_NAN"])
        x[hit] = words[r.integers(0, words.size, size=int(hit.sum()))]
    compare_ood = bool(os.environ.get("FX_FUZZ_OOD"))
    picks = (0, 65, N // 3, N - 1) + tuple(int(v) for v in np.random.default_rng(5).integers(0, N, size=12))
    kernels = {}
    for seed in range(n_prog):
        rng = np.random.default_rng(7000 + seed)
        text = random_program(rng, int(rng.integers(8, 120)), int(rng.integers(3, 60)))
        b = A.Batch(N, 1, 0)
        assert b.load_text(text), b.errors()
        y = b.process_block(x)
        y = b.process_block(x)
        kernels[b.info("kernel")] = kernels.get(b.info("kernel"), 0) + 1
        for n in picks:
            o = Oracle(1)
            assert o.load_text(text)
            o.process_block(x[:, n].copy())
            ref = o.process_block(x[:, n].copy())
            if o.ood_flags() and not compare_ood:
                continue  # e.g. LOG of an unclamped register left [-1, 1]: outside the parity domain
            same = np.array_equal(ref.view(np.uint32), y[:, n].view(np.uint32))
            assert same, "seed %d instance %d differs (kernel %d)\n%s" % (seed, n, b.info("kernel"), text)
        del b
    print("fuzz at scale ok:", n_prog, "programs x", N, "instances; kernels used", kernels)


if __name__ == "__main__":
    main()
