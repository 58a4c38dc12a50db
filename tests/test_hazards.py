"""Hazard lint (tools/gfx950_lint.py) over everything this repository puts on a gfx950: the hand-written interpreter (all
builds, both flavours) and the code the translator generates for the benchmark programs and a corpus of random programs.
llvm-mc checks encodings (tests/test_xlate.py); nothing in the assembler inserts the wait states the hardware does not
interlock, or knows the interpreter's VGPR-index-mode convention - this does.  CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gfx950_lint as L  # noqa: E402

import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402

BUILD = os.path.join(ROOT, "fx8010-emulator-core_amd", "csrc", "build")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(L.LLVM, "llvm-objdump")), reason="llvm tools not available")

# scalar registers the interpreter's handlers write from the vector ALU (compares, carries): what a generated stream must not
# read from the vector ALU right after a handler returns
HANDLER_VALU_SGPRS = {"vcc", "s62", "s63", "s64", "s65", "s66", "s67"}


def lint_text(asm, **kw):
    return L.lint_hazards(L.disassemble_listing(asm), **kw)


# ------------------------------------------------------------------------------------------------ the lint itself
def test_lint_sees_each_rule():
    assert lint_text("v_cmp_gt_f32_e32 vcc, 0, v2\nv_cndmask_b32_e32 v3, v4, v5, vcc\ns_endpgm")                       # R1, 0 wait states
    assert lint_text("v_cmp_gt_f32_e32 vcc, 0, v2\ns_nop 0\nv_cndmask_b32_e32 v3, v4, v5, vcc\ns_endpgm")               # R1, 1
    assert not lint_text("v_cmp_gt_f32_e32 vcc, 0, v2\ns_nop 1\nv_cndmask_b32_e32 v3, v4, v5, vcc\ns_endpgm")
    assert not lint_text("v_cmp_gt_f32_e32 vcc, 0, v2\nv_mov_b32_e32 v9, v8\nv_mov_b32_e32 v7, v8\nv_cndmask_b32_e32 v3, v4, v5, vcc\ns_endpgm")
    assert lint_text("v_cmp_gt_f32_e64 s[62:63], 0, v2\ns_nop 0\nv_cndmask_b32_e64 v3, v4, v5, s[62:63]\ns_endpgm")
    assert lint_text("v_add_co_u32_e32 v3, vcc, v3, v15\nv_addc_co_u32_e32 v4, vcc, 0, v4, vcc\ns_endpgm")              # carry chain
    assert lint_text("v_readfirstlane_b32 s20, v3\ns_nop 0\nv_add_f32_e64 v2, s20, v2\ns_endpgm")
    # a scalar instruction in between that overwrites the register hands ITS value on: no hazard
    assert not lint_text("v_cmp_gt_f32_e64 s[62:63], 0, v2\ns_and_b64 s[62:63], s[62:63], vcc\nv_cndmask_b32_e64 v3, v4, v5, s[62:63]\ns_endpgm",
                         assume_entry_defs=())
    # R2: a vector-written SGPR as the base of a vector memory instruction, 5 wait states
    assert lint_text("v_readfirstlane_b32 s20, v3\nv_readfirstlane_b32 s21, v4\ns_nop 3\nglobal_load_dword v2, v7, s[20:21]\ns_endpgm")
    assert not lint_text("v_readfirstlane_b32 s20, v3\nv_readfirstlane_b32 s21, v4\ns_nop 4\nglobal_load_dword v2, v7, s[20:21]\ns_endpgm")
    # R3 / R6 / R7: lane accesses
    assert lint_text("v_readfirstlane_b32 s20, v3\ns_nop 2\nv_readlane_b32 s21, v4, s20\ns_endpgm")
    assert lint_text("v_cmpx_gt_f32_e32 0, v2\ns_nop 2\nv_readfirstlane_b32 s21, v4\ns_endpgm")
    assert lint_text("v_mov_b32_e32 v4, v1\nv_readfirstlane_b32 s21, v4\ns_endpgm")
    # R5: M0 written by the scalar ALU, used by an LDS add-tid store
    assert lint_text("s_mov_b32 m0, s4\nds_write_addtid_b32 v2\ns_endpgm")
    assert not lint_text("s_mov_b32 m0, s4\ns_nop 0\nds_write_addtid_b32 v2\ns_endpgm")
    # R8: wide store data
    assert lint_text("v_mov_b32_e32 v4, v1\ns_nop 0\nglobal_store_dwordx4 v9, v[2:5], s[10:11]\ns_endpgm")
    assert not lint_text("v_mov_b32_e32 v4, v1\ns_nop 1\nglobal_store_dwordx4 v9, v[2:5], s[10:11]\ns_endpgm")
    # R10: transcendental result
    assert lint_text("v_exp_f32_e32 v4, v1\nv_add_f32_e32 v5, v4, v4\ns_endpgm")
    # across a branch: the hazard sits on the taken path only
    taken = "v_cmp_gt_f32_e32 vcc, 0, v2\ns_cbranch_scc1 2\ns_nop 1\ns_endpgm\nv_cndmask_b32_e32 v3, v4, v5, vcc\ns_endpgm"
    assert lint_text(taken)
    # the instruction after a call returns from a handler that may just have written VCC
    ret = "s_setpc_b64 s[62:63]\nv_cndmask_b32_e32 v3, v4, v5, vcc\ns_endpgm"
    assert lint_text(ret, assume_entry_defs=HANDLER_VALU_SGPRS)
    assert not lint_text("s_setpc_b64 s[62:63]\ns_nop 0\nv_cndmask_b32_e32 v3, v4, v5, vcc\ns_endpgm", assume_entry_defs=HANDLER_VALU_SGPRS)


def test_index_mode_lint_sees_the_convention():
    def problems(asm, state=L.UNKNOWN):
        ins = L.disassemble_listing(asm)
        return L.lint_index_mode(ins, entries={0: state})
    # a handler's first VALU instruction before it has set or cleared the mode (the bug class: v(3 + M0) is written)
    assert problems("v_mul_f32_e64 v3, s19, -1.0\ns_set_gpr_idx_on s18, gpr_idx(SRC0)\nv_add_f32_e32 v2, v32, v3\ns_setpc_b64 s[24:25]")
    assert not problems("s_set_gpr_idx_on s18, gpr_idx(SRC0)\nv_mul_f32_e64 v3, s19, -1.0\nv_add_f32_e32 v2, v32, v3\ns_setpc_b64 s[24:25]")
    # a plain register in the relative position
    assert problems("s_set_gpr_idx_on s18, gpr_idx(SRC0)\nv_add_f32_e32 v2, v3, v32\ns_setpc_b64 s[24:25]")
    assert not problems("s_set_gpr_idx_on s18, gpr_idx(SRC1)\nv_add_f32_e32 v2, v3, v32\ns_setpc_b64 s[24:25]")
    assert problems("s_set_gpr_idx_on s18, gpr_idx(SRC1)\nv_med3_f32 v5, -1.0, v2, 1.0\ns_setpc_b64 s[24:25]")
    assert not problems("s_set_gpr_idx_on s18, gpr_idx(SRC1)\nv_med3_f32 v5, v2, -1.0, 1.0\ns_setpc_b64 s[24:25]")
    # DST mode writes the register file only
    assert problems("s_set_gpr_idx_on s21, gpr_idx(DST)\nv_mov_b32_e32 v2, v5\ns_setpc_b64 s[24:25]")
    assert not problems("s_set_gpr_idx_on s21, gpr_idx(DST)\nv_cndmask_b32_e32 v32, v5, v2, vcc\ns_setpc_b64 s[24:25]")
    # two paths that meet with different modes
    assert problems("s_cbranch_scc1 1\ns_set_gpr_idx_on s18, gpr_idx(SRC0)\nv_mov_b32_e32 v2, v5\ns_setpc_b64 s[24:25]", state=L.OFF)


# ------------------------------------------------------------------------------------------------ the hand-written kernels
OBJECTS = ["fx_interp_lds"] + ["fx_%s_v%d" % (f, n) for f in ("interp", "xlate") for n in (64, 72, 80, 96, 128, 168, 256)]


@pytest.mark.parametrize("name", OBJECTS)
def test_handwritten_kernels(name):
    path = os.path.join(BUILD, name + ".o")
    assert os.path.exists(path), "build the library first (make -C fx8010-emulator-core_amd/csrc)"
    ins = L.disassemble_object(path)
    assert len(ins) > 1500
    findings = L.lint_hazards(ins)
    assert not findings, findings[:10]
    if name != "fx_interp_lds":
        problems = L.lint_index_mode(ins, kernel_labels=(name, name + "_probe"))
        assert not problems, problems[:10]


# ------------------------------------------------------------------------------------------------ generated code
def lint_program(text, vgprs=0, options=0, tracked=()):
    fe = A.FrontEnd(1)
    if options:
        fe.set_option(options)
    if not fe.load_text(text):
        return None
    for key in tracked:
        assert fe.track_register(key) == 0, key
    try:
        listing, size = L.image_listing(fe, vgprs)
    except RuntimeError:
        return None   # not eligible for the translated tier (multi-pass programs ...)
    ins = L.disassemble_listing(listing)
    assert ins[-1].addr + ins[-1].size == size, "the listing does not re-assemble to the encoder's layout (branch targets would be off)"
    findings = L.lint_hazards(ins, assume_entry_defs=HANDLER_VALU_SGPRS)
    problems = L.lint_index_mode(ins, entries=L.stream_entries(ins))
    return len(ins), findings, problems


@pytest.mark.parametrize("name", ["config1_shipped", "config1_logtube", "config2", "config3", "config4", "config5", "tram_bound"])
def test_generated_code_of_the_benchmark_programs(name):
    n, findings, problems = lint_program(P.CONFIGS[name]())
    assert n > 50
    assert not findings, findings[:10]
    assert not problems, problems[:10]


def test_generated_code_of_the_dane_model():
    n, findings, problems = lint_program(P.CONFIGS["config5_dane"](), options=A.OPT_TRAM_DANE)
    assert n > 1000 and not findings and not problems, (findings[:5], problems[:5])
    # taps gathered per lane (a modulated position) and interpolated reads between two slots
    from test_dane_tram import CHORUS, FRACTIONAL_TAPS
    for text in (CHORUS, FRACTIONAL_TAPS):
        n, findings, problems = lint_program(text, options=A.OPT_TRAM_DANE | A.OPT_TRAM_ADDR_SHIFT | A.OPT_TRAM_INTERP)
        assert n > 300 and not findings and not problems, (findings[:5], problems[:5])


def test_generated_code_with_control_tracks():
    """registers with schedules (fxb_set_register_track): the head's one compare, the event walk behind the loop - scalar loads,
    a per-slot compare chain, one value for all or one per instance, the taint check of what arrived"""
    cases = [("config5", ["damp", "decay", "diff", "lp1", "y3", "w2", "y17"]), ("config3", ["cutoff", "fb", "s3"]), ("config4", ["cutoff"]),
             ("config2", ["cutoff", "s0", "s30", "t"])]
    for name, tracked in cases:
        n, findings, problems = lint_program(P.CONFIGS[name](), tracked=tracked)
        assert n > 100 and not findings and not problems, (name, findings[:5], problems[:5])
    import stress_fuzz
    checked = 0
    for seed in range(0, 90, 3):
        rng = np.random.default_rng(8100000 + seed)
        text = stress_fuzz.random_program(rng, int(rng.integers(4, 60)), int(rng.integers(3, 30)))
        regs = [l.split()[1] for l in text.split("\n") if l.startswith(("static r", "control c"))]
        tracked = [str(r) for r in rng.choice(regs, size=min(len(regs), int(rng.integers(1, 6))), replace=False)]
        res = lint_program(text, tracked=tracked)
        if res is None:
            continue
        checked += 1
        assert not res[1] and not res[2], (seed, res[1][:5], res[2][:5])
    assert checked > 15


def test_generated_code_of_random_programs():
    """the random-program corpus of tests/test_xlate.py::test_random_programs_all_translate (both generators of the fuzzers)"""
    import stress_fuzz
    checked = 0
    for seed in range(0, 400, 3):
        rng = np.random.default_rng(500000 + 3000000 + seed)
        gen = stress_fuzz.random_program2 if seed % 2 else stress_fuzz.random_program
        text = gen(rng, int(rng.integers(4, 100)), int(rng.integers(2, 50)))
        res = lint_program(text)
        if res is None:
            continue
        checked += 1
        assert not res[1], (seed, res[1][:5])
        assert not res[2], (seed, res[2][:5])
    assert checked > 100
