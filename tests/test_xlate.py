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


@needs_llvm
def test_listing_with_control_tracks_reassembles():
    for name, tracked in (("config5", ["damp", "decay", "diff", "lp1", "y3", "w2", "y17"]), ("config3", ["cutoff", "fb"])):
        fe = A.FrontEnd(1)
        assert fe.load_text(P.CONFIGS[name]())
        plain = fe.translate(0, 0)[0]
        for key in tracked:
            assert fe.track_register(key) == 0
        assert fe.track_register("nosuch") == 1
        for stream in range(4):
            code, listing = fe.translate(0, stream)
            assert assemble(listing) == code, (name, stream)
            # ONE compare of the sample counter with the next event in the loop, whatever the number of schedules
            assert listing.count("s_cmp_eq_u32 s3, s28") == 2, (name, stream)   # (the head's, and the event walk's own)
        assert fe.translate(0, 0)[0] != plain
    fe = A.FrontEnd(1)
    assert fe.load_text(P.CONFIGS["config5"]())
    names = ["damp", "decay", "diff"] + ["lp%d" % i for i in range(4)] + ["y%d" % i for i in range(9)]
    for key in names:
        assert fe.track_register(key) == 0
    with pytest.raises(RuntimeError):
        fe.track_register("y20")   # the seventeenth


def test_hardware_instruction_counters_are_of_this_code():
    """profiles/*_pmc_valu.json (rocprofv3, SQ_INSTS_VALU_* per wavefront and sample of the benchmark run) against the
    translator's own listing of config5's steady fast stream: the committed counters describe the code that is generated today"""
    import collections
    import glob
    import json
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "*config5_pmc_valu.json")))
    if not files:
        pytest.skip("no committed VALU counter pass")
    hw = json.load(open(files[-1]))["per_wave_sample"]
    fe = A.FrontEnd(1)
    assert fe.load_text(P.CONFIGS["config5"]())
    listing = fe.translate(128, 0)[1]
    n = collections.Counter(l.split()[0] for l in listing.split("\n") if l)
    mine = {"SQ_INSTS_VALU_MUL_F32": n["v_mul_f32_e32"], "SQ_INSTS_VALU_ADD_F32": n["v_add_f32_e32"] + n["v_sub_f32_e32"],
            "SQ_INSTS_VALU_FMA_F32": n["v_fma_f32"], "SQ_INSTS_VALU_FMA_F64": n["v_fma_f64"],
            "SQ_INSTS_VALU_CVT": n["v_cvt_f64_f32_e32"] + n["v_cvt_f32_f64_e32"]}
    for key, count in mine.items():
        assert abs(hw[key] - count) < 0.5, (key, hw[key], count)


def test_code_fingerprint_tells_code_objects_apart():
    """fxp_code_hash (= FXB_INFO_XLATE_CODE_HASH of a batch in the same situation): equal for equal inputs, different for another
    program, another VGPR build, another stage count, non-temporal delay-line accesses"""
    def h(name, *a, **k):
        fe = A.FrontEnd(1)
        assert fe.load_text((P.CONFIGS.get(name) or P.PROBE_PROGRAMS[name])())
        return fe.code_hash(*a, **k)
    base = h("config5", 128)
    assert base == h("config5", 128) and 0 < base < 2 ** 63
    others = {h("config5", 168), h("config5", 128, 1, True), h("config4", 128), h("config2", 128), h("config2", 128, 8), h("config2", 128, 4)}
    assert base not in others and len(others) == 6
    with pytest.raises(RuntimeError):
        h("config5", 64)   # 59 rows do not fit the 64-register build


def test_committed_counter_passes_name_the_code_generated_today():
    """every profiles/*_pmc_valu.json that carries the fingerprint of the code object it was collected on (bench.code_hash, since
    round 4) - the unstaged streams of config3 / 4 / 5 and config2's eight stages - against the fingerprint of what the
    translator generates now for that configuration (fxp_code_hash, no device): bench.py prints
    simd_issue_busy_from_counters from a committed pass only when the two agree, this test says when a pass has gone stale"""
    import glob
    import json
    newest = {}
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "*_pmc_valu.json"))):
        d = json.load(open(f))
        b = d.get("bench", {})
        if b.get("code_hash") and b.get("config") in P.CONFIGS:
            newest[b["config"]] = (f, d)
    for config, (f, d) in newest.items():
        b = d["bench"]
        m = re.match(r"fx_xlate_v(\d+)", d["workload"]["kernel"])
        assert m, (f, d["workload"]["kernel"])
        fe = A.FrontEnd(1)
        option = getattr(P, "CONFIG_OPTIONS", {}).get(config, 0)
        if option:
            fe.set_option(option)
        assert fe.load_text(P.CONFIGS[config]())
        slots = {"config5": 8192, "config3": 1000, "tram_bound": 8192, "config5_dane": 8192}.get(config, 0)
        streaming = slots * ((b["instances"] + 63) // 64) * 256 > (512 << 20)   # (fx_batch.cpp: delay lines beyond the caches)
        waves = (b["instances"] + 63) // 64
        resident = {64: 8, 72: 7, 80: 6, 96: 5, 128: 4, 168: 3, 256: 2}[int(m.group(1))]
        slices = (b.get("stages") or 1) == 1 and waves >= 2048 and resident <= 4   # (fx_batch.cpp: two or more wavefronts per SIMD on a build of at most four slots)
        now = "%016x" % fe.code_hash(int(m.group(1)), b.get("stages") or 1, streaming, slices)
        assert now == b["code_hash"], "%s was collected on other code than is generated today (%s vs %s): run tools/profile_configs.sh again" % (os.path.basename(f), b["code_hash"], now)


def branch_targets(listing):
    """(line index, target line index) of every SOPP branch of a listing (targets resolved through instruction sizes)"""
    lines = listing.strip().split("\n")
    code = assemble(listing)
    # byte offset of every line: assemble prefixes to find instruction sizes cheaply via a single pass of llvm-mc output is
    # overkill - every instruction here is 4 or 8 bytes, and a literal or a VOP3/DS/global encoding makes it 8
    sizes = []
    for l in lines:
        if l.startswith(";"):   # (a comment of the listing - the exact streams' sync points - is no code)
            sizes.append(0)
            continue
        op = l.split()[0]
        eight = (op.endswith("_e64") or op in ("v_med3_f32", "v_med3_i32", "v_fma_f32", "v_fma_f64", "v_add_f64", "v_mul_f64") or op.startswith(("global_", "ds_", "s_memrealtime"))
                 or re.search(r"0x[0-9a-f]+", l) is not None)
        sizes.append(8 if eight else 4)
    assert sum(sizes) == len(code), "instruction size model"
    offs = np.concatenate([[0], np.cumsum(sizes)])
    out = []
    for k, l in enumerate(lines):
        m = re.match(r"^(s_branch|s_cbranch_\w+) (\d+)$", l)
        if not m:
            continue
        simm = int(m.group(2))
        simm = simm - 65536 if simm >= 32768 else simm
        target = int(offs[k + 1]) + 4 * simm
        out.append((k, target, int(offs[-1])))
    return lines, offs, out


@needs_llvm
def test_structure_of_translated_code():
    """A stream is a whole sample loop: head, program, PCM out, loop branch (steady) or the jump to the epilogue (last),
    then the cold-entry stub that branches back to the head."""
    for name, text in program_texts():
        fe = A.FrontEnd(1)
        assert fe.load_text(text), name
        for vgprs, stream in ((0, 0), (0, 1), (256, 0), (0, 2), (0, 3)):
            code, listing = fe.translate(vgprs, stream)
            if not code:
                continue  # no fast stream (non-finite uniform)
            lines, offs, branches = branch_targets(listing)
            last = stream >= 2
            assert lines[-1].startswith("s_branch "), name  # the cold stub ends with the branch to the head
            assert branches[-1][1] == 0, (name, "cold stub must enter at the head")
            # the jump to the template's epilogue appears exactly once, in last-sample streams only
            assert listing.count("s_setpc_b64 s[34:35]") == (1 if last else 0), name
            # every VGPR named is inside the build's budget (the smallest build is at least 64)
            budget = vgprs if vgprs else 256
            for m in re.finditer(r"\bv(\d+)\b", listing):
                assert int(m.group(1)) < budget
            fe.lower()
            first_spare = 32 + fe.lower_info("num_rows")
            for m in re.finditer(r"v\[(\d+):(\d+)\]", listing):
                lo, hi = int(m.group(1)), int(m.group(2))
                # temporaries (v[2:5] .. v[12:13]), or an even-aligned pair of the product cache above the register file
                assert hi < 14 or (lo % 2 == 0 and lo >= first_spare and hi < budget and stream in (0, 2)), (name, m.group(0))
            # SGPR writes: sample index, PCM pointers, the record window, return address, scratch, TRAM cursors / LUT bases,
            # the two flags of the early TRAM reads
            allowed = {3, 12, 13, 14, 15} | set(range(18, 26)) | set(range(62, 68)) | set(range(80, 96))
            for m in re.finditer(r"^(s_[a-z0-9_]+) s\[?(\d+)", listing, re.M):
                if not m.group(1).startswith(("s_cmp", "s_setpc", "s_waitcnt", "s_cbranch", "s_branch", "s_nop", "s_set_gpr")):
                    assert int(m.group(2)) in allowed | {78}, (name, m.group(0))
            # branches stay inside the stream, except the fast streams' escapes (forward, into an exact stream) and a
            # steady stream's hand-over to its last-sample stream
            outside = [(k, t) for k, t, end in branches if not (0 <= t <= end)]
            for k, t in outside:
                assert lines[k].startswith("s_cbranch_scc1") or (not last and lines[k].startswith("s_branch")), (name, lines[k])
            if stream in (1, 3):
                assert all(lines[k].startswith("s_branch") for k, t in outside), name  # exact streams never leave for another flavour
            if not last:
                back = [k for k, t, end in branches if t == 0 and lines[k].startswith("s_cbranch_scc1")]
                assert len(back) == 1, (name, "one loop branch to the head")


@needs_llvm
def test_dane_model_listing_reassembles_and_has_no_cursor_advance():
    """opt-in DANE delay-line model: taps at (counter + position) mod size generated inline, counters stepped at the end of the
    sample, next sample's taps loaded early with the translate-time collision rule"""
    fe = A.FrontEnd(1)
    fe.set_option(A.OPT_TRAM_DANE)
    assert fe.load_text(P.CONFIGS["config5_dane"]()), fe.errors()
    for stream in (0, 1, 2, 3, 4):
        code, listing = fe.translate(0, stream)
        assert assemble(listing) == code
    code, listing = fe.translate(0, 0)
    assert listing.count("global_load_dword") >= 8 + 1   # 4 taps in place (first sample) + 4 early + the PCM input
    assert "s_cmp_eq_u32 s95, 0" in listing                # the early reads are compiled in
    assert listing.count("s_sub_i32 s82, s82, 1") == 1     # one step of the xTRAM counter per sample, nothing per instruction
    fe2 = A.FrontEnd(1)
    assert not fe2.load_text(P.CONFIGS["config5_dane"]().replace("at, 1187", "at, &d0"))  # without the option: reference syntax only
    # a tap whose position is a per-instance value in whole samples: the taps are the interpreter's handlers (their DANE path),
    # called as subroutines, and the counters step per lane
    chorus = ("itramsize 2880 \ninput in 0\noutput out 0\nstatic rd1\nstatic lfo = 0.25\ncontrol depth = 0.4\nidelay write, in, at, 0\n"
              "idelay read, rd1, at, 1439\nmacs lfo, lfo, 0.01, 0.3\nmacs &rd1, 0.5, lfo, depth\nmacs out, 0, rd1, 0.5\nend")
    fe3 = A.FrontEnd(1)
    fe3.set_option(A.OPT_TRAM_DANE)
    assert fe3.load_text(chorus), fe3.errors()
    for stream in (0, 1, 2, 3):
        code, listing = fe3.translate(0, stream)
        assert assemble(listing) == code
        assert listing.count("v_cmp_gt_i32_e32 vcc, 1, v16") == 1 and "v_cmp_gt_i32_e32 vcc, 1, v18" not in listing   # (no xTRAM in this program)
        assert "s_sub_i32 s80, s80, 1" not in listing


@needs_llvm
def test_priority_turns_listing_reassembles(monkeypatch):
    """what a batch of two or more wavefronts per SIMD generates (fx_batch.cpp; on a build of at most four wave slots): every
    fourth sample the wavefront reads the 100 MHz clock and takes the priority ((clock >> s8) + wave-buffer slot) & 3; s8 comes from
    the run-once code (log2 of the block length + log2 of the modelled sample period / 24, at least 16).  The new encodings
    (s_memrealtime, s_getreg_b32, s_setprio, s_flbit_i32_b32, s_max_i32, s_lshr_b32) against llvm-mc; off by default here."""
    fe = A.FrontEnd(1)
    assert fe.load_text(P.CONFIGS["config5"]())
    code0, listing0 = fe.translate(128, 0)
    assert "s_setprio" not in listing0 and "s_memrealtime" not in listing0
    monkeypatch.setenv("FX_XLATE_PRIO", "1")
    for stream in (0, 1, 2, 3, 4):
        code, listing = fe.translate(128, stream)
        assert assemble(listing) == code, stream
        if stream < 4:
            assert listing.count("s_memrealtime s[62:63]") == 1 and [listing.count("s_setprio %d" % k) for k in range(4)] == [1, 1, 1, 1]
            assert "s_getreg_b32 s64, hwreg(HW_REG_HW_ID, 0, 4)" in listing and "s_lshr_b32 s62, s62, s8" in listing
        else:
            assert listing.startswith("s_flbit_i32_b32 s8, s9\ns_sub_i32 s8, 36, s8\ns_max_i32 s8, s8, 16\ns_min_i32 s8, s8, 20\n"), listing[:120]
    assert len(fe.translate(128, 0)[0]) > len(code0)
    assert fe.code_hash(128, 1, False, True) != fe.code_hash(128, 1, False, False)


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


def test_quick_lut_guess_is_within_one_segment():
    """The translated LOG/EXP code guesses the segment with three double-rate instructions (fx_xlate.cpp Translator::lut):
    q = RN(RN(x * 252 + 251.5) + 1.5 * 2^23), byte offset = bits(q) & 0x1f8.  Its miss path can move the index by one
    only, so the guess must never be further than that from the reference's (int)((x + 1.0) / step), and always in the table."""
    rng = np.random.default_rng(5)
    knots = (-1.0 + np.arange(64, dtype=np.float64) * (2.0 / 63.0)).astype(np.float32)
    near = [knots]
    for d in range(1, 5):
        up, dn = knots.copy(), knots.copy()
        for _ in range(d):
            up = np.nextafter(up, np.float32(2.0)); dn = np.nextafter(dn, np.float32(-2.0))
        near += [up, dn]
    x = np.concatenate(near + [rng.uniform(-1.0, 1.0, size=2_000_000).astype(np.float32),
                               np.array([1.0, -1.0, 0.0, -0.0, 1e-30, -1e-30, 1e-45, 0.99999994, -0.99999994], dtype=np.float32)])
    x = np.clip(x, -1.0, 1.0).astype(np.float32)
    # fp32 fma and add, each one rounding: the exact values fit a double
    t = (x.astype(np.float64) * 252.0 + 251.5).astype(np.float32)
    q = (t.astype(np.float64) + 12582912.0).astype(np.float32)
    offset = q.view(np.uint32) & np.uint32(0x1F8)
    guess = (offset >> 3).astype(np.int64)
    step = (1.0 - -1.0) / 63.0
    truth = ((x.astype(np.float64) - -1.0) / step).astype(np.int64)
    assert guess.min() >= 0 and guess.max() <= 63
    assert np.abs(guess - truth).max() <= 1
    assert (guess != truth).mean() < 1e-4          # the second LDS round trip stays rare
    assert guess[x == np.float32(1.0)].min() == 63 and guess[x == np.float32(-1.0)].max() == 0   # saturated operands hit


@needs_llvm
def test_quick_lut_guess_is_generated_for_bounded_operands():
    fe = A.FrontEnd(1)
    assert fe.load_text(P.CONFIGS["config4"]())
    code, listing = fe.translate(0, 0)
    # the miss paths live behind the loop: every LOG/EXP falls through an s_cbranch_vccnz whose target lies behind the loop
    # branch, and each miss path returns to the instruction after its branch
    lines, offs, branches = branch_targets(listing)
    loop = [k for k, t, end in branches if t == 0 and lines[k].startswith("s_cbranch_scc1")][0]
    misses = [(k, t) for k, t, end in branches if lines[k].startswith("s_cbranch_vccnz")]
    assert len(misses) == listing.count("v_cmp_le_u32_e32 vcc") and all(t > offs[loop] for k, t in misses)
    returns = sorted(t for k, t, end in branches if lines[k].startswith("s_branch") and k > loop and 0 < t < offs[loop])
    assert returns == sorted(int(offs[k + 1]) for k, t in misses)
    assert listing.count("v_and_b32_e32 v7") == listing.count("v_cmp_le_u32_e32 vcc") > 0
    assert "v_cvt_i32_f32" not in listing           # config4's LOG/EXP operands are all results of saturating instructions
    assert assemble(listing) == code


def test_random_programs_all_translate():
    """The batch falls back to the interpreter tier silently when a translation fails - e.g. when the fast and the exact stream
    of a program disagree about where they wait (their sync points).  Every random program of the fuzzers' generators must
    translate, in all four streams."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_fuzz
    failures = []
    for seed in range(400):
        rng = np.random.default_rng(500000 + 3000000 + seed)
        gen = stress_fuzz.random_program2 if seed % 2 else stress_fuzz.random_program
        text = gen(rng, int(rng.integers(4, 100)), int(rng.integers(2, 50)))
        fe = A.FrontEnd(1)
        if not fe.load_text(text):
            continue
        for stream in range(4):
            try:
                fe.translate(0, stream)
            except RuntimeError as e:
                failures.append((seed, stream, str(e)[:120]))
                break
    assert not failures, failures[:5]


def _record_word_faults(listing):
    """walk an exact stream's listing: a vector instruction that reads one of the record words s16..s23 needs that word written
    since the last sync point (`; sync point`: where a wavefront of the fast stream may arrive - with whatever THAT stream left
    in the record words)"""
    written, faults = set(), []
    for k, l in enumerate(listing.split("\n")):
        if l.startswith("; sync point"):
            written = set()
            continue
        m = re.match(r"^s_mov_b32 s(\d+),", l)
        if m and 16 <= int(m.group(1)) <= 23:
            written.add(int(m.group(1)))
            continue
        m = re.match(r"^s_mov_b64 s\[(\d+):(\d+)\],", l)
        if m and 16 <= int(m.group(1)) <= 23:
            written.update(range(int(m.group(1)), int(m.group(2)) + 1))
            continue
        if not l.startswith("v_"):
            continue
        operands = l.split(None, 1)[1] if " " in l else ""
        for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", operands):
            for r in range(int(a), int(b) + 1):
                if 16 <= r <= 23 and r not in written:
                    faults.append((k, l, r))
        for r in re.findall(r"\bs(\d+)\b", operands):
            if 16 <= int(r) <= 23 and int(r) not in written:
                faults.append((k, l, int(r)))
    return faults


def test_the_exact_stream_takes_no_record_word_for_granted_across_a_sync_point():
    """A wavefront leaves the fast stream for the exact one at a sync point (a lane met a value outside the bounded class) and
    continues there mid-sample: whatever the exact stream loaded into s16..s23 EARLIER in the sample - the fp64 (1 - X) of an
    INTERP with a constant X in s[22:23] - that wavefront has not executed, and the fast stream does not load the same words (it
    drops dead instructions).  The exact stream therefore forgets its record words at every sync point (fx_xlate.cpp syncPoint;
    the API fuzzer's control panel found an INTERP running with (1 - X) = 0, seed 2605911; on the device:
    tests/test_gpu_parity.py::test_a_wavefront_that_changes_streams_finds_the_record_words_set).  Here, without a GPU: every
    listing of an exact stream - the benchmark programs, fuzzed programs, the fuzzer's control-panel programs, a program made for
    it - is walked; no vector instruction may read a record word that has not been written since the last sync point."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import fuzz_api
    made = ("itramsize 3 \ninput in 0\noutput out 0\ncontrol k = 0.125\nstatic a\nstatic b\nstatic c\nstatic t\nstatic rd\n"
            "interp a, in, k, a\nmacs a, in, 0, 0\nidelay read, rd, at, 0\ninterp b, rd, k, b\nandxor t, 2, 2, 0\nidelay write, t, at, 0\n"
            "interp c, a, 0.5, c\nmacs out, b, c, 0.25\nend")
    texts = [("made for it", made)] + program_texts()
    for s in range(40):
        rng = np.random.default_rng(990000 + s)
        texts.append(("panel%d" % s, fuzz_api.with_panel(rng, fuzz_text(7100 + s))))
    walked = marks = reads = 0
    for name, text in texts:
        fe = A.FrontEnd(1)
        assert fe.load_text(text), name
        for stream in (1, 3):
            code, listing = fe.translate(0, stream)
            assert code, name
            walked += 1
            marks += listing.count("; sync point")
            reads += len(re.findall(r"^v_\w+ .*\bs\[22:23\]", listing, flags=re.M))
            faults = _record_word_faults(listing)
            assert not faults, (name, stream, faults[:3])
        assert "; sync point" not in fe.translate(0, 0)[1]          # (the fast streams leave, they are not arrived at)
    assert walked >= 100 and marks > 500 and reads > 50, (walked, marks, reads)
