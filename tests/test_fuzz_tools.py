"""The fuzzers' own generators (tools/): a generator whose programs the front-end refuses tests nothing - round 5's control panel
first declared a control with a negative initial value, every program failed to load and every sequence "passed"."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tools", "oracle", os.path.join("fx8010-emulator-core_amd", "python")):
    sys.path.insert(0, os.path.join(ROOT, p))

import fx8010_amd as A  # noqa: E402
from pyoracle import Oracle  # noqa: E402


def test_the_control_panel_programs_load_and_use_their_controls():
    """tools/fuzz_api.py with_panel: four declared controls spliced into operand positions of random programs - both generators of
    tools/stress_fuzz.py - load in the library's front-end and in the oracle (the reference's declaration syntax:
    /root/reference/source/FX8010.cpp:408-411), and the controls do turn up as operands"""
    import fuzz_api
    import stress_fuzz
    used = 0
    for seed in range(60):
        rng = np.random.default_rng(880000 + seed)
        gen = stress_fuzz.random_program2 if seed % 2 else stress_fuzz.random_program
        text = fuzz_api.with_panel(rng, gen(rng, int(rng.integers(6, 70)), int(rng.integers(3, 30))))
        fe = A.FrontEnd(1)
        assert fe.load_text(text), (seed, fe.errors())
        o = Oracle(1)
        assert o.load_text(text), seed
        for name, v in (("c", 0.3), ("c2", 0.5), ("c3", 0.25), ("c4", 0.125)):
            assert o.get_register_bits(name) == int(np.float32(v).view(np.uint32)), (seed, name)
        body = [l for l in text.split("\n") if l.split() and l.split()[0] not in ("control", "static", "input", "output", "itramsize", "xtramsize")]
        used += sum(any(tok.strip(",") in ("c2", "c3", "c4") for tok in l.split()[1:]) for l in body)
    assert used > 200, used


def test_the_api_fuzzer_reports_programs_that_did_not_load():
    """its summary counts the programs that loaded and the exit code says so when fewer than half did"""
    import fuzz_api
    src = open(os.path.join(ROOT, "tools", "fuzz_api.py")).read()
    assert 'STATS["loaded"] += 1' in src and 'STATS["loaded"] < count // 2' in src
    assert callable(fuzz_api.run) and "edit" in fuzz_api.run.__code__.co_varnames
