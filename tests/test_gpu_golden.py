"""GPU path against the committed golden vectors (tests/golden/*.json: outputs of the UNMODIFIED reference, see
tests/golden/make_golden.py) - directly, without the oracle in between.  The fixture's single instance is placed
at several lanes of a batch (first, last lane of a wavefront, a ragged tail); every copy must reproduce the
reference's output words, final register bits, instruction count and loader diagnostics."""
import json
import os

import numpy as np
import pytest

import fx8010_programs as progs

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as fh:
        return json.load(fh)


def f32(hexstr, shape=None):
    a = np.frombuffer(bytes.fromhex(hexstr), dtype=np.uint32).view(np.float32)
    return a.reshape(shape) if shape else a


@pytest.fixture(params=["default", "unstaged", "asm", "asm_lds", "asm_v256", "hip"], ids=["xlate", "xlate_unstaged", "asm", "asm_lds", "asm_v256", "hip"])
def tier(request, monkeypatch):
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    monkeypatch.delenv("FX_STAGES", raising=False)
    if request.param == "unstaged":   # (the fixtures' batches are small: by default their programs run as pipelines of stages where they can be cut)
        monkeypatch.setenv("FX_STAGES", "1")
    elif request.param != "default":
        monkeypatch.setenv("FX_KERNEL", request.param)
    return request.param


def run_case(gpu, case, text=None, N=67, compare_registers=True, per_instance_sets=False):
    ch = case["channels"]
    b = gpu.Batch(N, ch, 0)
    ok = b.load_text(text if text is not None else case["program"])
    assert ok == case["load_ok"], case["name"]
    assert [list(e) for e in b.errors()] == [list(e) for e in case["errors"]], case["name"]
    assert b.controls() == case["controls"] and b.meta() == case["meta"], case["name"]
    if not ok:
        return
    x1 = f32(case["input"], case["shape"])  # [S] or [S, channels]
    S = x1.shape[0]
    x = np.repeat(x1.reshape(S, ch, 1), N, axis=2).reshape((S, N) if ch == 1 else (S, ch, N)).copy()
    sets = {int(k): v for k, v in case.get("sets", {}).items()}
    if "sets_bits" in case:  # values that JSON cannot carry (NaN payloads): IEEE bits
        sets = {int(k): [(reg, float(np.array([b], dtype=np.uint32).view(np.float32)[0])) for reg, b in v] for k, v in case["sets_bits"].items()}
    cuts = sorted(set([0] + list(sets) + [S]))
    outs = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        for reg, val in sets.get(lo, []):
            if per_instance_sets:  # the same value as one array entry per instance: the register becomes a per-lane row
                b.set_register_array(reg, np.full(N, val, dtype=np.float32))
            else:
                b.set_register(reg, val)
        if hi > lo:
            outs.append(b.process_block(x[lo:hi]))
    y = np.concatenate(outs, axis=0)
    want = f32(case["output"], case["shape"]).view(np.uint32)
    for n in (0, 63, 64, N - 1):
        got = (y[..., n] if ch == 1 else y[:, :, n]).view(np.uint32)
        bad = np.nonzero(want.reshape(-1) != np.ascontiguousarray(got).reshape(-1))[0]
        assert bad.size == 0, "%s instance %d: first mismatch at flat sample %d" % (case["name"], n, bad[0])
        assert b.instruction_counter_i(n) == case["counter"], (case["name"], n)
        for reg, bits in case["registers"].items():
            if compare_registers:
                assert b.get_register_bits_i(reg, n) == bits, (case["name"], reg, n)


@pytest.mark.parametrize("fixture", ["opcodes.json", "known_answers.json", "slider.json", "feedback_delay.json"])
def test_reference_vectors(gpu, tier, fixture):
    for case in load(fixture):
        run_case(gpu, case)  # (skip_over_end is a multi-pass program: the interpreter runs it where an assembly tier is asked, else the HIP C++ kernel)


# Non-finite values, bit for bit (tests/golden/nonfinite.json, nan_collisions.json: what the x86 build of the reference
# does with NaN payloads and signs, Inf, and NaNs made by the arithmetic itself) - on EVERY tier and every build of the
# hand-written interpreter: its VGPR builds reach a register-file row in either source position (VGPR index mode SRC0 / SRC1),
# so that the source order of each add / multiply is the x86 build's operand order (fx_interp_handlers.inc).
ALL_BUILDS = ["default", "hip", "asm", "asm_lds", "asm_v64", "asm_v72", "asm_v80", "asm_v96", "asm_v128", "asm_v168", "asm_v256", "xlate_v256"]


@pytest.mark.parametrize("per_instance", [False, True], ids=["uniform_controls", "per_instance_rows"])
@pytest.mark.parametrize("fixture", ["nonfinite.json", "nan_collisions.json"])
@pytest.mark.parametrize("build", ALL_BUILDS)
def test_non_finite_words_are_the_references(gpu, monkeypatch, build, fixture, per_instance):
    """per_instance: the NaN operands sit in register-file rows (the device arithmetic decides); otherwise they are
    uniform controls (folded on the host, or literals of the generated code)"""
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    if build != "default":
        monkeypatch.setenv("FX_KERNEL", build)
    for case in load(fixture):
        run_case(gpu, case, per_instance_sets=per_instance)


def test_config_programs(gpu, tier):
    for case in load("configs.json"):
        text = progs.CONFIGS[case["config"]]()
        run_case(gpu, dict(case), text=text)


def test_stage_planner_programs(gpu, tier):
    """configs_probe.json: the reference's own words for the program shapes the stage planner is calibrated with - at 67
    instances the default tier runs them as pipelines of stages (13-row packets; a delay line, SKIP shadows and LOG / EXP inside
    stages)"""
    for case in load("configs_probe.json"):
        run_case(gpu, dict(case), text=progs.PROBE_PROGRAMS[case["config"]]())


def test_delay_lines_past_the_first_read_back(gpu, tier):
    """configs_long.json: config3 / config5 over 2304 samples of the REFERENCE's own output - config5's 8192-slot line hands
    back its first written word at sample 2048 (four reads + four writes per sample, cursors per executed TRAM instruction,
    source/FX8010.cpp:909-967, 1188-1211), config3's at 1000 and 2000 - in uneven blocks, so that reads issued a sample ahead
    cross block ends after the wrap as well"""
    import hashlib
    for case in load("configs_long.json"):
        text = progs.CONFIGS[case["config"]]()
        S = case["shape"][0]
        x1 = progs.stimulus(1, S, first_instance=case["instance"])[:, 0].copy()
        assert hashlib.sha256(x1.view(np.uint32).tobytes()).hexdigest() == case["input_sha256"], "stimulus generator drifted"
        N = 67
        x = np.repeat(x1.reshape(S, 1), N, axis=1).copy()
        b = gpu.Batch(N, 1, 0)
        assert b.load_text(text), b.errors()
        cuts = [0, 999, 1000, 1003, 2047, 2049, 2100, S]
        y = np.concatenate([b.process_block(x[lo:hi]) for lo, hi in zip(cuts[:-1], cuts[1:])], axis=0)
        want = f32(case["output"]).view(np.uint32)
        for n in (0, 63, 64, N - 1):
            got = np.ascontiguousarray(y[:, n]).view(np.uint32)
            bad = np.nonzero(want != got)[0]
            assert bad.size == 0, "%s instance %d: first mismatch at sample %d" % (case["name"], n, bad[0])
            assert b.instruction_counter_i(n) == case["counter"], (case["name"], n)
            for reg, bits in case["registers"].items():
                assert b.get_register_bits_i(reg, n) == bits, (case["name"], reg, n)
        assert b.ood_flags() == 0


def test_lut_probe_all_exponents(gpu, tier):
    probe = load("lut_probe.json")
    x1 = f32(probe["input"])
    S, N = x1.shape[0], 65
    x = np.repeat(x1.reshape(S, 1), N, axis=1).copy()
    hdr = "input in 0\noutput out 0\n"
    for c in probe["cases"]:
        op, e = c["name"][:3], int(c["name"][3:])
        b = gpu.Batch(N, 1, 0)
        assert b.load_text(hdr + "%s out, in, %d, 0\nend" % (op, e))
        y = b.process_block(x)
        want = f32(c["output"]).view(np.uint32)
        for n in (0, 64):
            assert np.array_equal(np.ascontiguousarray(y[:, n]).view(np.uint32), want), (c["name"], n)
