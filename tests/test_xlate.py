"""Translator (fx_xlate.cpp): the machine code it emits is checked, without a GPU, against the assembler.

The translator returns its code together with an assembler listing of the same instructions; llvm-mc
assembles the listing for gfx950 and the bytes must be identical.  That pins every opcode number and field
position of the hand-written encoder.  Structural checks (register bounds, call targets, stream end) guard
what a wrong instruction could do on a device."""
import os
import re
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

import fx8010_amd as A
import fx8010_programs as P

LLVM = "/opt/rocm/lib/llvm/bin"
MC = os.path.join(LLVM, "llvm-mc")
OBJCOPY = os.path.join(LLVM, "llvm-objcopy")
needs_llvm = pytest.mark.skipif(not (os.path.exists(MC) and os.path.exists(OBJCOPY)), reason="llvm-mc not available")


def assemble(listing):
    d = tempfile.mkdtemp()
    try:
        src, obj, raw = os.path.join(d, "x.s"), os.path.join(d, "x.o"), os.path.join(d, "x.bin")
        with open(src, "w") as fh:
            fh.write(".text\n" + listing)
        subprocess.run([MC, "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", obj, src], check=True, capture_output=True)
        subprocess.run([OBJCOPY, "-O", "binary", "--only-section=.text", obj, raw], check=True, capture_output=True)
        with open(raw, "rb") as fh:
            return fh.read()
    finally:
        shutil.rmtree(d, ignore_errors=True)


def fuzz_text(seed):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_fuzz
    rng = np.random.default_rng(seed)
    return stress_fuzz.random_program(rng, int(rng.integers(8, 120)), int(rng.integers(3, 60)))


PROGRAMS = [(name, P.CONFIGS[name]) for name in ("config1_shipped", "config2", "config3", "config4", "config5") if name in P.CONFIGS]


def program_texts():
    out = [(n, f()) for n, f in PROGRAMS]
    out += [("fuzz%d" % s, fuzz_text(7000 + s)) for s in range(12)]
    return out


@needs_llvm
@pytest.mark.parametrize("last", [0, 1, 2, 3, 4], ids=["steady_fast", "steady_exact", "last_fast", "last_exact", "run_once"])
def test_listing_reassembles_to_the_same_bytes(last):
    seen = 0
    for name, text in program_texts():
        fe = A.FrontEnd(1)
        assert fe.load_text(text), name
        code, listing = fe.translate(0, last)
        if last == 4 and not code:
            continue  # no LOG/EXP tables to stage in LDS
        seen += 1
        assert len(code) > 0 and len(code) % 4 == 0
        again = assemble(listing)
        assert again[: len(code)] == code, "%s: encoder and assembler disagree" % name
        assert len(again) == len(code), name
    assert seen > 0


def test_structure_of_translated_code():
    for name, text in program_texts():
        fe = A.FrontEnd(1)
        assert fe.load_text(text), name
        for vgprs, stream in ((0, 0), (0, 1), (256, 0), (0, 3)):
            code, listing = fe.translate(vgprs, stream)
            lines = listing.strip().split("\n")
            assert lines[-1] == "s_setpc_b64 s[34:35]", name  # back to the end-of-sample frame, nothing after it
            # every VGPR named is inside the build's budget (the smallest build is at least 64)
            budget = vgprs if vgprs else 256
            for m in re.finditer(r"\bv(\d+)\b", listing):
                assert int(m.group(1)) < budget
            for m in re.finditer(r"v\[(\d+):(\d+)\]", listing):
                assert int(m.group(2)) < 14  # temporaries only (v[2:5] .. v[12:13])
            # SGPR writes stay inside the record window, the return address, the scratch pair, the TRAM cursor and LUT base blocks
            allowed = set(range(18, 26)) | {62, 63} | set(range(80, 94))  # + s[88:93]: LUT table bases
            for m in re.finditer(r"^(s_[a-z0-9_]+) s(\d+),", listing, re.M):
                if not m.group(1).startswith(("s_cmp", "s_setpc")):
                    assert int(m.group(2)) in allowed, (name, m.group(0))
            # branches inside a stream are the fixed skips of the inline TRAM code
            for m in re.finditer(r"^(s_branch|s_cbranch_scc0) (\d+)$", listing, re.M):
                assert int(m.group(2)) in (1, 2, 6), (name, m.group(0))
            if stream in (1, 3):
                assert "s_cbranch_scc1" not in listing  # only the fast streams leave (for the exact ones)
            # a call's return address is the instruction after its s_setpc_b64
            words = np.frombuffer(code, dtype=np.uint32)
            assert words[-1] == 0xBE801D22  # s_setpc_b64 s[34:35]


def test_translate_reports_ineligible_programs():
    fe = A.FrontEnd(1)
    # SKIP over END: multi-pass program, runs on the HIP C++ kernel instead
    assert fe.load_text("static a = 0.5\nmacs a, a, 0, 0\nskip ccr, ccr, 8, 1\nend")
    with pytest.raises(RuntimeError):
        fe.translate()


def test_vgpr_budget_is_enforced():
    fe = A.FrontEnd(1)
    assert fe.load_text(P.CONFIGS["config5"]())
    with pytest.raises(RuntimeError):
        fe.translate(64)  # config5 needs the 96-VGPR build
    code96, _ = fe.translate(96)
    code0, _ = fe.translate(0)
    assert code96 == code0
