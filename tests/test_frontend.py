"""Host logic of the product library, no GPU: the loader and the lowering (fxp_*), the C ABI's
exported symbols, and the no-device behaviour (the product must fail loudly, never fall back)."""
import json
import os
import re

import numpy as np
import pytest

import fx8010_programs as progs
from pyoracle import Oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(amd):
    header = open(os.path.join(ROOT, "include", "fx8010_amd.h")).read()
    declared = set(re.findall(r"\b(fx[bp]?_[a-z0-9_]+)\s*\(", header))
    lib = amd.load()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(amd.SYMBOLS), declared ^ set(amd.SYMBOLS)


def test_front_end_matches_reference_corpus(amd):
    with open(os.path.join(GOLD, "parser_corpus.json")) as fh:
        corpus = json.load(fh)
    for c in corpus:
        fe = amd.FrontEnd(1)
        ok = fe.load_text(c["program"])
        assert ok == c["load_ok"], repr(c["program"])
        assert [list(e) for e in fe.errors()] == [list(e) for e in c["errors"]], repr(c["program"])
        assert fe.controls() == c["controls"] and fe.meta() == c["meta"] and fe.ready() == c["ready"], repr(c["program"])


def test_front_end_model_matches_oracle(amd):
    with open(os.path.join(GOLD, "parser_corpus.json")) as fh:
        texts = [c["program"] for c in json.load(fh)]
    with open(os.path.join(GOLD, "opcodes.json")) as fh:
        texts += [c["program"] for c in json.load(fh)]
    texts += [fn() for fn in progs.CONFIGS.values()]
    for text in texts:
        fe, o = amd.FrontEnd(1), Oracle(1)
        assert fe.load_text(text) == o.load_text(text)
        assert fe.registers() == o.registers(), repr(text)       # names, types, channel, initial value bits, ORDER
        assert fe.instructions() == o.instructions(), repr(text)  # opcode, R/A/X/Y indices, hasInput/hasOutput/hasNoise
        assert fe.tram_sizes() == o.tram_sizes()


def test_second_load_accumulates_like_the_reference(amd):
    fe, o = amd.FrontEnd(1), Oracle(1)
    for text in ("static a\nmacs a, 0, 0, 0\nend", "static b\nmacs b, a, 1, 1\nend"):
        assert fe.load_text(text) == o.load_text(text)
    assert fe.registers() == o.registers() and fe.instructions() == o.instructions() and fe.errors() == o.errors()
    assert len(fe.instructions()) == 4


def test_lut_tables_bit_identical(amd):
    o = Oracle(1)
    for kind in (0, 1):
        for e in range(32):
            assert np.array_equal(amd.FrontEnd.lut(kind, e).view(np.uint64), o.lut(kind, e).view(np.uint64)), (kind, e)


@pytest.mark.parametrize("name,expect", [
    ("config2", dict(num_instructions=64, tram_ops=0, num_shadowed=0, multipass=0, num_ccr_live=0)),
    ("config3", dict(num_instructions=256, tram_ops=2, itram_slots=1000, num_shadowed=0, num_ccr_live=0)),
    ("config4", dict(num_instructions=512, num_shadowed=126, num_ccr_live=63, multipass=0)),
    ("config5", dict(num_instructions=512, tram_ops=8, xtram_slots=8192, num_shadowed=0)),
])
def test_lowering_of_benchmark_programs(amd, name, expect):
    fe = amd.FrontEnd(1)
    assert fe.load_text(progs.CONFIGS[name]())
    assert fe.lower() == 0, fe.last_error()
    for k, v in expect.items():
        assert fe.lower_info(k) == v, k
    assert fe.lower_info("num_lane_regs") + fe.lower_info("num_uniform_regs") == fe.lower_info("num_registers")


def test_lowering_classifies_registers(amd):
    fe = amd.FrontEnd(1)
    text = "input in 0\noutput out 0\ncontrol vol = 0.5\nstatic a\nstatic unused\nmacs a, in, vol, 0.25\nmacs out, 0, a, 1.0\nend"
    assert fe.load_text(text) and fe.lower() == 0
    # per instance: ccr, in, out, a ; uniform: read, write, at, vol, unused, 0.25, 0, 1.0
    assert fe.lower_info("num_lane_regs") == 4 and fe.lower_info("num_uniform_regs") == 8
    # the only CCR reader is nobody: no instruction needs to materialise it in steady state
    assert fe.lower_info("num_ccr_live") == 0
    fe = amd.FrontEnd(1)
    assert fe.load_text("input in 0\noutput out 0\nstatic a\nmacs a, in, 0, 0\nskip ccr, ccr, 6, 1\nmacs out, 0, in, 1.0\nend") and fe.lower() == 0
    assert fe.lower_info("num_ccr_live") == 1 and fe.lower_info("num_shadowed") == 1 and fe.lower_info("multipass") == 0
    fe = amd.FrontEnd(1)
    assert fe.load_text("input in 0\noutput out 0\nstatic a\nmacs a, in, 0, 0\nskip ccr, ccr, 6, 2\nmacs out, 0, in, 1.0\nend") and fe.lower() == 0
    assert fe.lower_info("multipass") == 1  # the SKIP can jump over END


def test_programs_that_cannot_be_lowered(amd):
    fe = amd.FrontEnd(1)
    assert fe.lower() < 0  # nothing loaded
    fe = amd.FrontEnd(1)
    assert fe.load_text("itramsize 9000 \nstatic a\nidelay write, a, at, 0\nend")  # the reference accepts the first oversize
    assert fe.lower() < 0 and "parity domain" in fe.last_error()


def test_no_device_fails_loudly(amd):
    if amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no usable HIP device"):
        amd.Batch(64, 1, 0)
    with pytest.raises(RuntimeError, match="no usable HIP device"):
        amd.Single(1)


def test_product_sources_do_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "fx8010-emulator-core_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath) == "build":
            continue
        for f in files:
            if f.endswith((".cpp", ".hpp", ".hip", ".S", ".h", ".py")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "fx8010_oracle" not in text and "libfxoracle" not in text and "libfxref" not in text, f
