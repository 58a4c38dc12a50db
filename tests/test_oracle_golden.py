"""The CPU oracle (oracle/fx8010_oracle.c) against golden vectors generated from the UNMODIFIED
reference (tests/golden/make_golden.py, run in the build container).  Bit-exact everywhere."""
import json
import os

import numpy as np
import pytest

import fx8010_programs as progs
from pyoracle import Oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as fh:
        return json.load(fh)


def f32(hexstr, shape=None):
    a = np.frombuffer(bytes.fromhex(hexstr), dtype=np.uint32).view(np.float32)
    return a.reshape(shape) if shape else a


def run_oracle(case, text=None):
    o = Oracle(case["channels"])
    ok = o.load_text(text if text is not None else case["program"])
    assert ok == case["load_ok"]
    assert [list(e) for e in o.errors()] == [list(e) for e in case["errors"]]
    assert o.controls() == case["controls"] and o.meta() == case["meta"]
    if not ok:
        return o
    x = f32(case["input"], case["shape"])
    sets = {int(k): v for k, v in case.get("sets", {}).items()}
    if "sets_bits" in case:  # values that JSON cannot carry (NaN payloads): IEEE bits
        sets = {int(k): [(reg, float(np.array([b], dtype=np.uint32).view(np.float32)[0])) for reg, b in v] for k, v in case["sets_bits"].items()}
    if sets:
        cuts = sorted(set([0] + list(sets) + [x.shape[0]]))
        outs = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            for reg, val in sets.get(lo, []):
                o.set_register(reg, val)
            outs.append(o.process_block(x[lo:hi]))
        y = np.concatenate(outs, axis=0)
    else:
        y = o.process_block(x)
    want = f32(case["output"], case["shape"])
    bad = np.nonzero(want.view(np.uint32).reshape(-1) != y.view(np.uint32).reshape(-1))[0]
    assert bad.size == 0, "%s: first mismatch at sample %d" % (case["name"], bad[0])
    assert o.instruction_counter() == case["counter"]
    for reg, bits in case["registers"].items():
        assert o.get_register_bits(reg) == bits, (case["name"], reg)
    return o


@pytest.mark.parametrize("fixture", ["opcodes.json", "known_answers.json", "slider.json", "feedback_delay.json", "nonfinite.json", "nan_collisions.json"])
def test_oracle_matches_reference_vectors(fixture):
    for case in load(fixture):
        o = run_oracle(case)
        if case["name"] != "skip_over_end":
            assert o.ood_flags() == 0, case["name"]


def test_known_answers_are_the_surveyed_bits():
    """The probed words of SURVEY.md §8(a) really are in the fixtures (guards the fixture generator)."""
    cases = {c["name"]: c for c in load("known_answers.json")}
    interp = f32(cases["interp_step"]["output"]).view(np.uint32)
    assert [hex(v) for v in interp] == ["0x3dcccccd", "0x3e428f5c", "0x3e8ac083", "0x3eb013a9", "0x3ed1ab4b", "0x3eefe6f7"]
    noise = f32(cases["noise_first6"]["output"]).view(np.uint32)
    assert [hex(v) for v in noise] == ["0xbe70b07b", "0x3f660df3", "0x3f2a45d6", "0x3dc5c057", "0xbee0fb6e", "0x3f08cac0"]
    log3 = f32(cases["log3_ramp32"]["output"]).view(np.uint32)
    assert hex(log3[0]) == "0xbf800000" and hex(log3[8]) == "0xbf4a123c" and hex(log3[16]) == "0x0" and hex(log3[24]) == "0x3f4a123b"
    assert cases["slider_shipped"]["counter"] == 64 if "slider_shipped" in cases else True


def test_slider_counter_counts_end():
    case = load("slider.json")[0]
    assert case["counter"] == 64  # 2 per sample: END is counted


def test_lut_probe_all_exponents():
    probe = load("lut_probe.json")
    x = f32(probe["input"])
    hdr = "input in 0\noutput out 0\n"
    for c in probe["cases"]:
        op, e = c["name"][:3], int(c["name"][3:])
        o = Oracle(1)
        assert o.load_text(hdr + "%s out, in, %d, 0\nend" % (op, e))
        y = o.process_block(x)
        assert np.array_equal(y.view(np.uint32), f32(c["output"]).view(np.uint32)), c["name"]


def test_config_programs_against_reference():
    for case in load("configs.json"):
        case = dict(case)
        text = progs.CONFIGS[case["config"]]()
        x = progs.stimulus(1, case["shape"][0], first_instance=case["instance"])[:, 0]
        assert np.array_equal(x.view(np.uint32), f32(case["input"]).view(np.uint32)), "stimulus generator drifted"
        o = run_oracle(case, text=text)
        assert o.ood_flags() == 0


def test_stage_planner_programs_against_reference():
    """configs_probe.json: the program shapes of tools/stage_policy_probe.sh (parallel chains, delay line + SKIP + LOG / EXP)"""
    for case in load("configs_probe.json"):
        case = dict(case)
        text = progs.PROBE_PROGRAMS[case["config"]]()
        x = progs.stimulus(1, case["shape"][0], first_instance=case["instance"])[:, 0]
        assert np.array_equal(x.view(np.uint32), f32(case["input"]).view(np.uint32)), "stimulus generator drifted"
        assert run_oracle(case, text=text).ood_flags() == 0


def test_delay_lines_past_the_first_read_back_against_reference():
    """configs_long.json: config3 / config5 over 2304 samples - the reference's cursors advance per executed TRAM instruction
    (source/FX8010.cpp:909-967), so config5's 8192-slot line hands back its first written word at sample 2048 (config3 at 1000
    and 2000); the 384-sample cases of configs.json never get there"""
    import hashlib
    for case in load("configs_long.json"):
        case = dict(case)
        text = progs.CONFIGS[case["config"]]()
        x = progs.stimulus(1, case["shape"][0], first_instance=case["instance"])[:, 0].copy()
        assert hashlib.sha256(x.view(np.uint32).tobytes()).hexdigest() == case["input_sha256"], "stimulus generator drifted"
        case["input"] = x.view(np.uint32).tobytes().hex()
        o = run_oracle(case, text=text)
        assert o.ood_flags() == 0
        if case["config"] == "config5":   # the fixture really reads written words back: d0..d3 are no longer the line's zeros
            assert case["registers"]["d0"] & 0x7fffffff and case["registers"]["d3"] & 0x7fffffff
            short = Oracle(1)
            assert short.load_text(text)
            short.process_block(x[:2048].copy())
            assert short.get_register_bits("d0") == 0 and short.get_register_bits("d3") == 0   # ... and one sample earlier they were


def test_loader_corpus():
    for c in load("parser_corpus.json"):
        o = Oracle(1)
        ok = o.load_text(c["program"])
        assert ok == c["load_ok"], repr(c["program"])
        assert [list(e) for e in o.errors()] == [list(e) for e in c["errors"]], repr(c["program"])
        assert o.controls() == c["controls"] and o.meta() == c["meta"] and o.ready() == c["ready"], repr(c["program"])


def test_out_of_domain_flags():
    hdr = "static rd\ninput in 0\noutput out 0\n"
    o = Oracle(1)
    assert o.load_text("itramsize 8 \n" + hdr + "idelay write, in, at, 0\nidelay read, rd, at, 2\nmacs out, 0, rd, 1.0\nend")
    o.process_block(np.zeros(4, np.float32))
    assert o.ood_flags() & 1
    o = Oracle(1)
    assert o.load_text(hdr + "idelay read, rd, at, 0\nend")  # no itramsize: the reference divides by zero
    o.process_block(np.zeros(2, np.float32))
    assert o.ood_flags() & 4
    o = Oracle(1)
    assert o.load_text(hdr + "log out, in, 40, 0\nend")
    o.process_block(np.zeros(2, np.float32))
    assert o.ood_flags() & 8
