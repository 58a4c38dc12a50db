"""Live differential test: the C restatement against the compiled reference (oracle/_ref), on
randomly generated in-domain programs.  Runs wherever oracle/_ref/libfxref.so exists."""
import os

import numpy as np
import pytest

# programs per generator run live against the compiled reference; the compiler-variant runs of test_oracle_variants.py (four more
# passes over this file, one of them under AddressSanitizer) take a fifth of them
LIVE_CASES = int(os.environ.get("FXORACLE_LIVE_CASES", "300"))

import fx8010_programs as progs
from pyoracle import Oracle, Reference

pytestmark = pytest.mark.skipif(not Reference.available(), reason="oracle/_ref not built (needs /root/reference)")

OPS3 = ["macs", "macsn", "macints", "acc3", "macw", "macwn", "macintw", "macmv", "tstneg", "limit", "limitn", "interp", "andxor"]


def random_program(rng, n_instr):
    regs = ["r%d" % i for i in range(6)]
    lits = ["0", "0.5", "-0.25", "1.0", "0.125", "2", "-1", "0.999", "3", "7", "15"]
    L = ["input in 0", "output out 0", "control c = 0.3", "static noise", "itramsize 11 ", "xtramsize 23 "] + ["static %s" % r for r in regs]
    L.append("idelay read, r5, at, 0")
    L.append("xdelay read, r4, at, 0")
    body = []
    for i in range(n_instr):
        kind = rng.integers(0, 100)
        src = lambda: str(rng.choice(regs + lits + ["in", "c", "out", "ccr", "noise"]))
        dst = str(rng.choice(regs + ["out"]))
        if kind < 70:
            body.append("%s %s, %s, %s, %s" % (rng.choice(OPS3), dst, src(), src(), src()))
        elif kind < 80:
            body.append("%s %s, %s, %d, 0" % (rng.choice(["log", "exp"]), dst, str(rng.choice(["in", "c", "0.5", "-0.25"])), rng.integers(0, 32)))
        elif kind < 90 and i + 4 < n_instr:
            body.append("skip ccr, ccr, %s, %d" % (rng.choice(["0", "2", "6", "8", "16", "20"]), rng.integers(0, 3)))
        else:
            body.append("macs %s, %s, %s, %s" % (dst, src(), src(), src()))
    L += body
    L.append("idelay write, r0, at, 0")
    L.append("xdelay write, r1, at, 0")
    L.append("macs out, out, r2, 0.5")
    L.append("end")
    return "\n".join(L)


@pytest.mark.parametrize("seed", range(LIVE_CASES))
def test_random_programs(seed):
    rng = np.random.default_rng(1000 + seed)
    text = random_program(rng, int(rng.integers(5, 60)))
    x = progs.stimulus(1, 200, first_instance=seed)[:, 0].copy()
    o, r = Oracle(1), Reference(1)
    assert o.load_text(text) and r.load_text(text), (o.errors(), r.errors())
    yo, yr = o.process_block(x), r.process_block(x)
    # LOG/EXP of an unclamped register can leave [-1,1]: that is outside the parity domain
    if o.ood_flags() == 0:
        assert np.array_equal(yo.view(np.uint32), yr.view(np.uint32)), text       # strict: NaN words bit for bit as well
        assert o.instruction_counter() == r.instruction_counter()
        for reg in ("r0", "r1", "r2", "r3", "r4", "r5", "ccr", "out", "in"):
            assert o.get_register_bits(reg) == r.get_register_bits(reg), (reg, text)


@pytest.mark.parametrize("seed", range(LIVE_CASES))
def test_random_programs_with_nonfinite_input(seed):
    """NaN (either sign, payloads, signalling) and Inf words sprinkled over the input: which payload survives each instruction is
    the x86 operand order the restatement spells out (sse_pick32/64); strict bit compare against the compiled reference."""
    rng = np.random.default_rng(5000 + seed)
    text = random_program(rng, int(rng.integers(5, 60)))
    x = progs.stimulus(1, 200, first_instance=seed)[:, 0].copy()
    words = np.array([0x7FC00000, 0xFFC00000, 0x7F800001, 0xFFA00123, 0x7FFFFFFF, 0x7F800000, 0xFF800000, 0x7FC12345], dtype=np.uint32)
    hit = rng.random(x.shape[0]) < 0.08
    x.view(np.uint32)[hit] = rng.choice(words, size=int(hit.sum()))
    o, r = Oracle(1), Reference(1)
    assert o.load_text(text) and r.load_text(text)
    yo = o.process_block(x)
    # a NaN that reaches LOG / EXP indexes the reference's table out of bounds (it crashes): the restatement runs first and
    # tells whether the case is inside the parity domain
    if o.ood_flags() == 0:
        yr = r.process_block(x)
        assert np.array_equal(yo.view(np.uint32), yr.view(np.uint32)), text
        assert o.instruction_counter() == r.instruction_counter()
        for reg in ("r0", "r1", "r2", "r3", "r4", "r5", "ccr", "out", "in"):
            assert o.get_register_bits(reg) == r.get_register_bits(reg), (reg, text)


def test_stereo_input_quirk_against_reference():
    text = ("input l 0\ninput r 1\noutput ol 0\noutput or 1\nstatic t\n"
            "macs ol, l, r, 0.5\nmacs or, r, l, 0.5\nmacs t, 0, r, 1.0\nmacs or, or, t, 0.25\nend")
    x = np.stack([progs.stimulus(1, 64)[:, 0], progs.stimulus(1, 64, seed=5)[:, 0]], axis=1).copy()
    o, r = Oracle(2), Reference(2)
    assert o.load_text(text) and r.load_text(text)
    assert np.array_equal(o.process_block(x).view(np.uint32), r.process_block(x).view(np.uint32))


@pytest.mark.parametrize("seed", range(max(LIVE_CASES // 5, 1)))
def test_the_fuzzers_programs_against_reference(seed):
    """the program generators of the GPU fuzzers (tools/stress_fuzz.py: all opcodes, delay lines, SKIP shadows, LOG / EXP with
    uniform and per-lane operands, literals as destinations) - what the GPU path is compared with the restatement on - through the
    compiled reference as well, wherever the restatement says the case is inside the parity domain"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_fuzz
    rng = np.random.default_rng(770000 + seed)
    gen = stress_fuzz.random_program2 if seed % 2 else stress_fuzz.random_program
    text = gen(rng, int(rng.integers(6, 70)), int(rng.integers(3, 30)))
    x = progs.stimulus(1, 160, first_instance=seed)[:, 0].copy()
    o = Oracle(1)
    if not o.load_text(text):
        r = Reference(1)
        assert not r.load_text(text) and [list(e) for e in r.errors()] == [list(e) for e in o.errors()]
        return
    yo = o.process_block(x)
    if o.ood_flags() != 0:
        return   # (outside the parity domain the reference reads out of bounds, hangs or crashes)
    r = Reference(1)
    assert r.load_text(text)
    yr = r.process_block(x)
    assert np.array_equal(yo.view(np.uint32), yr.view(np.uint32)), text
    assert o.instruction_counter() == r.instruction_counter()
    assert o.get_register_bits("ccr") == r.get_register_bits("ccr")
