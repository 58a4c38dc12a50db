"""GPU parity: the HIP interpreter (through the C ABI) against the CPU oracle, bit for bit.

Every test drives libfx8010_amd.so exactly as a user would (load .da text, process blocks,
read registers / counters) and compares with oracle/ (the C restatement of the reference,
itself pinned by tests/golden).  Bar: bit-exact outputs, registers and instruction counts.
"""
import numpy as np
import pytest

import fx8010_programs as progs
from pyoracle import Oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["default", "unstaged", "xlate_v256", "asm", "asm_v256", "asm_lds", 1, 2, 4],
                ids=["xlate", "xlate_unstaged", "xlate_v256", "asm", "asm_v256", "asm_lds", "k1", "k2", "k4"])
def k(request, monkeypatch):
    """kernel variant: the default choice (program translated to gfx950 code; the small batches of these tests run it as a
    pipeline of stages wherever the program can be cut), the same with FX_STAGES=1 (one wavefront runs the whole program: what
    large batches get), the translation forced into the 256-VGPR build, the hand-written interpreter (smallest VGPR build,
    forced 256-VGPR build, forced LDS build), or the HIP C++ kernel with 1/2/4 instances per lane
    (FX_INST_PER_LANE pins it)"""
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    monkeypatch.delenv("FX_STAGES", raising=False)
    if request.param == "default":
        pass
    elif request.param == "unstaged":
        monkeypatch.setenv("FX_STAGES", "1")
    elif isinstance(request.param, str):
        monkeypatch.setenv("FX_KERNEL", request.param)
    else:
        monkeypatch.setenv("FX_INST_PER_LANE", str(request.param))
    return request.param

HDR = "static a\nstatic b\ninput in 0\noutput out 0\nstatic noise\nstatic rd\ncontrol vol = 0.5\n"


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def oracle_run(text, x, channels=1, pre=None, regs=()):
    """x: [S] or [S, channels] for ONE instance -> (out, counter, {reg: bits}, ood)"""
    o = Oracle(channels)
    assert o.load_text(text), o.errors()
    if pre:
        pre(o)
    y = o.process_block(x)
    return y, o.instruction_counter(), {r: o.get_register_bits(r) for r in regs}, o.ood_flags()


def check_batch(gpu, text, x, regs=("ccr",), channels=1, instances=None, blocks=1, expect_ood=0):
    """x: [S, N] (mono) or [S, channels, N]."""
    S = x.shape[0]
    N = x.shape[-1]
    b = gpu.Batch(N, channels, 0)
    assert b.load_text(text), b.errors()
    if blocks == 1:
        y = b.process_block(x)
    else:
        cuts = np.linspace(0, S, blocks + 1).astype(int)
        y = np.concatenate([b.process_block(x[cuts[i]:cuts[i + 1]]) for i in range(blocks) if cuts[i + 1] > cuts[i]], axis=0)
    total = 0
    for n in (range(N) if instances is None else instances):
        xin = x[..., n] if channels == 1 else np.ascontiguousarray(x[:, :, n])
        ref, cnt, rv, ood = oracle_run(text, xin, channels, regs=regs)
        got = y[..., n] if channels == 1 else y[:, :, n]
        bad = np.nonzero(bits(ref).reshape(-1) != bits(got).reshape(-1))[0]
        assert bad.size == 0, "instance %d: first mismatch at flat sample %d: ref %08x got %08x" % (
            n, bad[0], bits(ref).reshape(-1)[bad[0]], bits(got).reshape(-1)[bad[0]])
        assert b.instruction_counter_i(n) == cnt, "instance %d instruction counter" % n
        for r in regs:
            assert b.get_register_bits_i(r, n) == rv[r], "instance %d register %s" % (n, r)
        assert ood == expect_ood
        total += cnt
    if instances is None:
        assert b.instruction_counter() == total
    assert b.ood_flags() == expect_ood
    return b, y


@pytest.mark.parametrize("name", ["config1_shipped", "config1_logtube", "config2", "config3", "config4", "config5", "tram_bound"])
def test_config_programs_bit_exact(gpu, name, k):
    text = progs.CONFIGS[name]()
    N, S = 130, 257  # ragged last wavefront, odd block length
    x = progs.stimulus(N, S)
    regs = {"config2": ("t", "s30", "in", "out", "ccr"), "config3": ("rd", "a", "t", "ccr"), "config4": ("x", "a", "b", "o", "ccr"),
            "config5": ("m", "u", "v", "w3", "ccr")}.get(name, ("ccr",))
    b, _ = check_batch(gpu, text, x, regs=regs)
    if isinstance(k, str):
        want = {"default": tuple(range(9, 16)), "unstaged": tuple(range(9, 16)), "xlate_v256": (15,), "asm": (2, 3, 4, 5, 6, 7, 8), "asm_v256": (8,), "asm_lds": (1,)}[k]
        assert b.info("kernel") in want and b.info("inst_per_lane") == 1
    else:
        assert b.info("kernel") == 0 and b.info("inst_per_lane") == k


def test_block_boundaries_do_not_matter(gpu, k):
    text = progs.config3()
    x = progs.stimulus(70, 1200)  # > 1000-sample delay: the feedback path wraps
    _, y1 = check_batch(gpu, text, x, instances=[0, 69], regs=("rd", "ccr"))
    _, y7 = check_batch(gpu, text, x, instances=[0, 69], regs=("rd", "ccr"), blocks=7)
    assert np.array_equal(bits(y1), bits(y7))


OPCODE_PROGRAMS = {
    "macs": "macs out, in, vol, 0.75",
    "macsn": "macsn out, in, vol, 0.75",
    "macints": "macints out, in, in, 2",
    "acc3": "acc3 out, in, vol, 0.25",
    "macw": "macw out, in, 1.5, 1.0",
    "macwn": "macwn out, in, 1.5, in",
    "macintw": "macintw out, in, 1.5, 1.0",
    "macw_ccr_as_a": "macw out, ccr, 1.5, in",
    "macmv": "macmv out, in, 0.5, 0.5",
    "andxor_and": "macs a, 0, in, 100\nandxor out, a, 15, 0",
    "andxor_xor": "macw a, 0, in, 100\nandxor out, a, 16777215, 5",
    "andxor_generic": "macw a, 0, in, 100\nandxor out, a, 12, 3",
    "andxor_literals": "andxor out, 7, 5, 2",
    "tstneg": "tstneg out, in, 0.25, 0",
    "tstneg_overflow": "tstneg out, in, 1.0, 0",
    "limit": "limit out, in, 0.5, 0.25",
    "limitn": "limitn out, in, 0.5, 0.25",
    "log3": "log a, in, 3, 0\nmacs out, 0, a, 1.0",
    "log0": "log a, in, 0, 0\nmacs out, 0, a, 1.0",
    "log31": "log a, in, 31, 1\nmacs out, 0, a, 1.0",
    "exp7": "exp a, in, 7, 0\nmacs out, 0, a, 1.0",
    "exp0": "exp a, in, 0, 0\nmacs out, 0, a, 1.0",
    "exp_unclamped_r": "exp out, in, 2, 0",
    "interp": "interp out, out, 0.1, in",
    "interp_literals": "interp out, -0.25, vol, 0.25",
    "highpass": "interp a, a, 0.1, in\nmacsn out, in, a, 1",
    "skip_neg": "macs a, in, 0, 0\nskip ccr, ccr, 6, 1\nmacs out, 0, in, 1.0",
    "skip_zero": "macs a, in, 0, 0\nskip ccr, ccr, 8, 2\nmacs out, 0, in, 1.0\nmacs out, out, 0.5, 0.5",
    "skip_sat": "macs a, in, in, 1.0\nskip ccr, ccr, 16, 1\nmacs out, 0, a, 0.5",
    "skip_negative_count": "macs a, in, 0, 0\nskip ccr, ccr, 6, -3\nmacs out, 0, in, 1.0\nmacs b, out, 0.5, 0.5",
    "skip_never": "macs a, in, 0, 0\nskip ccr, ccr, 384, 1\nmacs out, 0, in, 1.0",
    "ccr_as_operand": "macs a, in, 0, 0\nmacs out, 0, ccr, 0.03125",
    "noise": "macs out, 0, noise, 1.0",
    "noise_twice": "macs a, 0, noise, 0.5\nmacs out, a, noise, 0.5",
    "literal_as_r": "macs 0.5, in, 0.5, 0.5\nmacs out, 0, 0.5, 1.0",
    "write_ccr_directly": "macs ccr, in, 0, 0\nmacs out, 0, ccr, 0.03125",
    "out_read_back": "macs out, out, in, 0.1",
    "idelay_nop_r": "idelay a, in, at, 0\nmacs out, 0, in, 1.0",
    "latch_on_skip": "macs out, 0, in, 1.0\nmacs out, out, 0.5, 0.5\nskip out, ccr, 2, 0",
}


@pytest.mark.parametrize("name", sorted(OPCODE_PROGRAMS))
def test_opcode_programs(gpu, name, k):
    text = HDR + OPCODE_PROGRAMS[name] + "\nend"
    S, N = 48, 70
    ramp = np.array([i / 16.0 for i in range(-16, 16)] + [1.0, -1.0, 0.0, -0.0, 1e-39, -1e-39, 0.999999, -0.999999] * 2, dtype=np.float32)
    x = np.stack([np.roll(ramp, n) for n in range(N)], axis=1)  # [S, N], each instance a rotation
    check_batch(gpu, text, x, regs=("a", "b", "out", "ccr", "in"))


@pytest.mark.parametrize("op,table", [("log", 1), ("log", 3), ("log", 16), ("log", 31), ("exp", 0), ("exp", 2), ("exp", 7), ("exp", 31)])
@pytest.mark.parametrize("kern", ["xlate", "asm", "asm_lds", "hip"])
@pytest.mark.parametrize("operand", ["input", "saturated"])
def test_log_exp_dense_sweep(gpu, op, table, kern, operand, monkeypatch):
    """LOG/EXP on the device use precomputed thresholds/slopes instead of the reference's two fp64
    divisions: sweep random x, every table knot and its float neighbours, the domain edges.
    The translated tier checks its segment guess in two ways: against the thresholds for an operand that can be
    anything (the input), by the window test for one known to lie in [-1, 1] (the result of a saturating instruction)."""
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    monkeypatch.setenv("FX_KERNEL", kern)
    if operand == "input":
        text = HDR + "%s out, in, %d, 0\nend" % (op, table)
    else:
        text = HDR + "macs a, in, 0, 0\n%s out, a, %d, 0\nend" % (op, table)
    rng = np.random.default_rng(table * 7 + len(op))
    knots = (-1.0 + np.arange(64, dtype=np.float64) * (2.0 / 63.0)).astype(np.float32)
    near = [knots]
    for d in (1, 2, 3):
        up, dn = knots.copy(), knots.copy()
        for _ in range(d):
            up = np.nextafter(up, np.float32(2.0)); dn = np.nextafter(dn, np.float32(-2.0))
        near += [up, dn]
    special = np.clip(np.concatenate(near + [np.array([1.0, -1.0, 0.0, -0.0, 1e-30, -1e-30, 1e-42, 0.99999994, -0.99999994], dtype=np.float32)]), -1.0, 1.0)
    S, N = 48, 128
    x = rng.uniform(-1.0, 1.0, size=S * N).astype(np.float32)
    x[:special.size] = special
    x = x.reshape(S, N)
    b, _ = check_batch(gpu, text, x, regs=("out", "ccr"), instances=range(0, N, 1))
    assert (b.info("kernel") > 0) == (kern != "hip")


def test_interp_uniform_x_sweep(gpu, k):
    """INTERP (FX8010.cpp:1180-1187) rounds (1-X)*A and the sum separately in fp64.  The translator uses one fma
    where (1-X)*A is exact for every float A (X with few significant bits after 1-X) and mul + add elsewhere:
    both sides of that decision, over operands from denormal to saturated."""
    consts = ["0.3", "0.5", "0.0001", "0.999", "1.0", "0.0", "0.75", "0.015625", "0.01", "0.0000001", "0.6180339", "0.99999994"]
    regs = ["r%d" % i for i in range(len(consts))]
    text = HDR + "".join("static %s\n" % r for r in regs)
    for r, c in zip(regs, consts):
        text += "interp %s, %s, %s, in\n" % (r, r, c)       # one-pole smoothing with state
        text += "interp a, in, %s, %s\n" % (c, r)
        text += "macs out, out, a, 0.03125\n"
    text += "end"
    rng = np.random.default_rng(99)
    S, N = 96, 128
    x = rng.uniform(-1.0, 1.0, size=(S, N)).astype(np.float32)
    x[:, 1] *= np.float32(1e-38)      # denormal products
    x[:, 2] *= np.float32(1e-20)
    x[:, 3] = np.where(rng.uniform(size=S) < 0.5, 1.0, -1.0).astype(np.float32)
    x[:, 4] = np.float32(2.0) * x[:, 4]  # beyond the saturation range on the input side
    x[5::7, 5] = 0.0
    check_batch(gpu, text, x, regs=tuple(regs) + ("a", "out", "ccr"))


def same_with_nan(ref, got, k="default"):
    """Bit-equal INCLUDING NaN words (sign, payload, quiet bit), on every tier: generated code, the hand-written interpreter
    and the HIP C++ kernel order their sources like the x86 build of the reference and never negate a NaN
    (tests/golden/nan_collisions.json, tools/micro/nanrules.hip)."""
    return np.array_equal(bits(np.asarray(ref, dtype=np.float32).reshape(-1)), bits(np.asarray(got, dtype=np.float32).reshape(-1)))


NONFINITE_PROGRAM = HDR + """static big = 100000000000000000000000000000000000000.0
static t
static u
static w
itramsize 6 
idelay read, rd, at, 0
macs a, a, vol, in
macw t, big, in, big
macs u, t, t, -1
acc3 w, u, a, 0.25
interp b, w, 0.25, a
idelay write, in, at, 0
macs out, rd, b, 0.5
end"""


NONFINITE_PLAIN_PROGRAM = HDR + """static t
static u
static w
itramsize 6 
idelay read, rd, at, 0
macs a, a, vol, in
macsn t, a, in, 0.5
macs u, t, rd, -1
acc3 w, u, a, 0.25
interp b, w, 0.25, a
idelay write, in, at, 0
macs out, rd, b, 0.5
end"""


@pytest.mark.parametrize("case", ["nan_input", "inf_input", "nan_state", "inf_uniform", "plain"])
def test_non_finite_values_without_wrap_or_skip(gpu, k, case):
    """as test_non_finite_values_follow_the_reference, with a program made of saturating instructions and TRAM only
    (no wrap-around instruction, no SKIP, no handler call): prologue / input / TRAM taint checks"""
    N, S = 200, 40  # four wavefronts, the last one ragged (8 instances)
    x = progs.stimulus(N, S).copy()
    if case == "nan_input":
        x[7, 3] = np.nan
        x[20, 70] = -np.nan
        x[25, 199] = np.nan      # last instance
    elif case == "inf_input":
        x[5, 64] = np.inf
        x[9, 1] = -np.inf
        x[9, 130] = np.inf
    elif case == "inf_uniform":
        x[4, 10] = 0.0
        x[4, 75] = 0.0
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(NONFINITE_PLAIN_PROGRAM), b.errors()
    if case == "nan_state":
        b.set_register_i("a", 33, float("nan"))
        b.set_register_i("a", 65, float("inf"))
        b.set_register_i("a", 197, float("nan"))
    elif case == "inf_uniform":
        b.set_register("vol", float("inf"))
    y = b.process_block(x)
    saw_nan = False
    for n in range(N):
        o = Oracle(1)
        assert o.load_text(NONFINITE_PLAIN_PROGRAM)
        if case == "nan_state" and n in (33, 197):
            o.set_register("a", float("nan"))
        if case == "nan_state" and n == 65:
            o.set_register("a", float("inf"))
        if case == "inf_uniform":
            o.set_register("vol", float("inf"))
        ref = o.process_block(x[:, n].copy())
        assert same_with_nan(ref, y[:, n], k), "instance %d" % n
        saw_nan = saw_nan or bool(np.isnan(ref).any())
        for r in ("a", "b", "t", "u", "w", "out"):
            rb, gb = o.get_register_bits(r), b.get_register_bits_i(r, n)
            assert rb == gb, "instance %d register %s: ref %08x got %08x" % (n, r, rb, gb)
        assert b.instruction_counter_i(n) == o.instruction_counter()
    if case in ("nan_input", "nan_state", "inf_uniform"):
        assert saw_nan


@pytest.mark.parametrize("case", ["nan_input", "inf_input", "overflow_in_macw", "nan_state", "inf_uniform", "plain"])
def test_non_finite_values_follow_the_reference(gpu, k, case):
    """NaN passes the reference's saturate() (FX8010.cpp:275-279) and Inf saturates; the translated program's
    fast stream assumes finite registers and must hand over to the exact stream wherever a non-finite value can
    enter: PCM input, state rows, TRAM reads (the delayed input), a non-saturating result (MACW overflow),
    a non-finite uniform."""
    N, S = 96, 40
    x = progs.stimulus(N, S).copy()
    pre = None
    if case == "nan_input":
        x[7, 3] = np.nan
        x[20, 70] = -np.nan
    elif case == "inf_input":
        x[5, 64] = np.inf
        x[9, 1] = -np.inf
    elif case == "inf_uniform":
        x[4, 10] = 0.0  # 0 * Inf
    elif case == "overflow_in_macw":
        x[11, 5] = 3.0  # big + wrap(3 * big) = +Inf in `t`, then Inf - Inf
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(NONFINITE_PROGRAM), b.errors()
    if case == "nan_state":
        b.set_register_i("a", 33, float("nan"))
        b.set_register_i("a", 65, float("inf"))
    elif case == "inf_uniform":
        b.set_register("vol", float("inf"))
        b.set_register("big", float("inf"))
    y = b.process_block(x)
    saw_nan = False
    for n in range(N):
        o = Oracle(1)
        assert o.load_text(NONFINITE_PROGRAM)
        if case == "nan_state" and n == 33:
            o.set_register("a", float("nan"))
        if case == "nan_state" and n == 65:
            o.set_register("a", float("inf"))
        if case == "inf_uniform":
            o.set_register("vol", float("inf"))
            o.set_register("big", float("inf"))
        ref = o.process_block(x[:, n].copy())
        assert same_with_nan(ref, y[:, n], k), "instance %d" % n
        saw_nan = saw_nan or bool(np.isnan(ref).any())
        for r in ("a", "b", "t", "u", "w", "out"):
            rb, gb = o.get_register_bits(r), b.get_register_bits_i(r, n)
            assert rb == gb, "instance %d register %s: ref %08x got %08x" % (n, r, rb, gb)
    if case in ("nan_input", "overflow_in_macw", "nan_state", "inf_uniform"):
        assert saw_nan  # the case does exercise NaN through saturating instructions

def test_a_wavefront_that_changes_streams_finds_the_record_words_set(gpu, monkeypatch):
    """The translated program has a fast stream and an exact one; a wavefront leaves the fast stream at the next sync point once a
    lane has met a value outside the bounded class (here: 2.0 read back from the delay line) and continues in the exact stream at
    the same point.  The two streams do not issue the same instructions - the fast one drops a dead INTERP - so nothing the exact
    stream loaded into the record words (s16..s23: the fp64 (1 - X) of an INTERP with a constant X) before a sync point may be
    taken for granted behind it.  Found by the API fuzzer's control panel (seed 2605911: an INTERP computed with (1 - X) = 0 in
    the first sample of a launch).  Blocks of one sample (every launch starts with whatever the prologue left in s23), of two and
    of five, a second constant X in the same program; against the oracle, /root/reference/source/FX8010.cpp:1180-1187."""
    for k in ("FX_KERNEL", "FX_INST_PER_LANE", "FX_STAGES"):
        monkeypatch.delenv(k, raising=False)
    text = ("itramsize 3 \ninput in 0\noutput out 0\ncontrol k = 0.125\nstatic a\nstatic b\nstatic c\nstatic t\nstatic rd\n"
            "interp a, in, k, a\nmacs a, in, 0, 0\nidelay read, rd, at, 0\ninterp b, rd, k, b\nandxor t, 2, 2, 0\nidelay write, t, at, 0\n"
            "interp c, a, 0.5, c\nmacs out, b, c, 0.25\nend")
    for N in (64, 300, 20000):
        for cuts in ([1] * 8, [2, 2, 2, 2], [5, 5], [1, 2, 5, 1, 3]):
            b = gpu.Batch(N, 1, 0)
            assert b.load_text(text), b.errors()
            x = (progs.stimulus(N, sum(cuts)) * 0.5).astype(np.float32)
            ys, at = [], 0
            for n in cuts:
                ys.append(b.process_block(x[at:at + n]))
                at += n
            y = np.concatenate(ys, axis=0)
            assert b.tier_note().startswith("translated to gfx950 code"), b.tier_note()
            for inst in (0, 63, N - 1):
                o = Oracle(1)
                assert o.load_text(text)
                ref = o.process_block(x[:, inst].copy())
                assert o.ood_flags() == 0
                assert np.array_equal(bits(ref), bits(y[:, inst])), (N, cuts, inst)
                assert b.get_register_bits_i("b", inst) == o.get_register_bits("b") and b.instruction_counter_i(inst) == o.instruction_counter()



def test_saturation_elision_is_sound(gpu, k):
    """The translator drops the saturation of an instruction whose result provably lies in [-1, 1] given that its
    row operands do (rows written only by saturating instructions, checked against 1.0 where they enter: state,
    inline TRAM reads).  Values beyond 1 arriving through every such entrance must put the saturation back."""
    text = ("itramsize 9 \n" + HDR + "static t\nstatic u\nstatic w\n"
            "idelay read, rd, at, 0\n"
            "macs t, 0, rd, 0.5\n"          # rd: bounded class through the checked inline read
            "macs u, 0, a, vol\n"            # a: bounded class (only saturating writers), state may be set beyond 1
            "interp w, u, 0.25, t\n"
            "interp b, in, 0.25, w\n"        # in: wild, keeps its saturation
            "macs a, a, 0.5, in\n"
            "idelay write, in, at, 0\n"
            "macs out, 0, w, 1.0\n"
            "acc3 t, t, u, b\n"
            "end")
    N, S = 130, 64
    x = progs.stimulus(N, S).copy()
    x[3, 7] = 3.0          # through the delay line into rd nine samples later
    x[20, 64] = -2.5
    x[30:34, 100] = 1.5
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    b.set_register_i("a", 12, 5.0)     # a bounded-class row starting outside [-1, 1]
    b.set_register_i("w", 70, -3.0)
    y = b.process_block(x)
    for n in range(N):
        o = Oracle(1)
        assert o.load_text(text)
        if n == 12:
            o.set_register("a", 5.0)
        if n == 70:
            o.set_register("w", -3.0)
        ref = o.process_block(x[:, n].copy())
        assert np.array_equal(bits(ref), bits(y[:, n])), "instance %d" % n
        for r in ("a", "b", "t", "u", "w", "rd", "out"):
            assert b.get_register_bits_i(r, n) == o.get_register_bits(r), "instance %d register %s" % (n, r)
    if k in ("default", "unstaged"):
        assert b.info("xlate_unsaturated") >= 3


def test_log_of_a_bounded_row_that_was_set_beyond_one(gpu, k):
    """LOG/EXP of a register the program only ever saturates needs no clamps - unless the host put 5.0 there: the
    taint check at block start moves the wave to the exact stream, whose LOG clamps the index and raises the
    out-of-domain flag exactly as the oracle defines it (flag 16 on that instance only)."""
    text = HDR + "static t\nlog b, a, 3, 0\nexp t, a, 7, 0\nmacs a, a, 0.25, in\nmacs out, b, t, 0.5\nend"
    N, S = 130, 9
    x = progs.stimulus(N, S)
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    b.set_register_i("a", 70, 5.0)
    b.set_register_i("a", 3, -1.5)
    y = b.process_block(x)
    for n in range(N):
        o = Oracle(1)
        assert o.load_text(text)
        if n == 70:
            o.set_register("a", 5.0)
        if n == 3:
            o.set_register("a", -1.5)
        ref = o.process_block(x[:, n].copy())
        assert np.array_equal(bits(ref), bits(y[:, n])), "instance %d" % n
        assert (o.ood_flags() == 16) == (n in (3, 70))
        for r in ("a", "b", "t"):
            assert b.get_register_bits_i(r, n) == o.get_register_bits(r)
    assert b.ood_flags() == 16  # the OR over all instances
    clean = gpu.Batch(N, 1, 0)
    assert clean.load_text(text)
    clean.process_block(x)
    assert clean.ood_flags() == 0


@pytest.mark.parametrize("op,table", [("log", 3), ("exp", 7)])
def test_log_of_the_input_inside_and_beyond_the_table(gpu, op, table, k):
    """LOG/EXP of a wild operand (the PCM input): the translated tier runs the bounded form while every lane of a wave
    holds |x| <= 1 and the guarded form - index clamp, out-of-domain flag 16, taint check of the result - as soon as one
    lane does not.  Waves that are entirely inside, waves with one lane at 1.01 / -1.5 / +-Inf / NaN, per instance
    against the oracle (whose restatement defines the clamped read beyond the table)."""
    text = HDR + "%s b, in, %d, 0\nmacs out, b, in, 0.25\nend" % (op, table)
    N, S = 200, 12
    rng = np.random.default_rng(table)
    x = rng.uniform(-1.0, 1.0, size=(S, N)).astype(np.float32)
    x[2, 5] = 1.01            # beyond 1.0 but inside the grid's last segment: no flag, clamped index
    x[3, 70] = -1.5           # out of the domain
    x[4, 71] = np.float32(np.inf)
    x[5, 72] = np.float32(np.nan)
    x[6, 130] = -1.0
    x[7, 131] = 1.0
    x[8, 9] = 1e10            # the quotient no longer fits an int32: clamped as a number (63), not wrapped to INT_MIN
    x[9, 10] = -1e10
    x[1, 11] = np.float32(-np.inf)
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    y = b.process_block(x)
    flagged = set()
    for n in range(N):
        o = Oracle(1)
        assert o.load_text(text)
        ref = o.process_block(x[:, n].copy())
        assert np.array_equal(bits(ref), bits(y[:, n])), "instance %d" % n
        assert b.get_register_bits_i("b", n) == o.get_register_bits("b")
        assert b.instruction_counter_i(n) == o.instruction_counter()
        if o.ood_flags():
            flagged.add(n)
            assert o.ood_flags() == 16
    assert flagged >= {70, 71, 72, 9, 10, 11} and not (flagged & {5, 130, 131, 0, 199})
    assert b.ood_flags() == 16


def test_table_number_per_instance(gpu, k):
    """LOG / EXP whose table number - the X operand, `(int32)X` (FX8010.cpp:1114,1121) - is a control with a different value on every
    instance (N reference objects with N settings of a "drive" slider): every assembly tier takes the table from the lane's own
    row (the translated program calls the interpreter's handler for these records, the rest of it stays generated code), and a
    number outside 0..31 - negative, 32, NaN, beyond int32 - is flagged (8) on that instance only and reads the nearest table,
    as the oracle defines it.  Also inside a SKIP shadow, and moving from block to block."""
    text = (HDR + "control e = 3\ncontrol f = 7\nstatic t\nstatic u\nlog t, in, e, 0\nexp u, t, f, 0\nmacs a, in, 0, 0\nskip ccr, ccr, 6, 1\n"
            "log u, a, f, 0\nmacs out, t, u, 0.5\nend")
    N, S = 200, 24
    x = progs.stimulus(N, S * 3)
    rng = np.random.default_rng(31)
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    oracles = []
    for n in range(N):
        o = Oracle(1)
        assert o.load_text(text)
        oracles.append(o)
    wild = {3: -1.0, 64: 32.0, 65: 31.999, 70: float(np.float32(np.nan)), 71: 4.0e9, 72: -0.5, 130: 1.0e-30, 131: float(np.float32(np.inf)), 199: -3.0e9}
    for blk in range(3):
        e = rng.integers(0, 32, size=N).astype(np.float32) + rng.uniform(0.0, 0.99, size=N).astype(np.float32)
        f = rng.integers(0, 32, size=N).astype(np.float32)
        if blk == 1:
            for n, v in wild.items():
                (e if n % 2 else f)[n] = v
        if blk < 2:      # (the third block: the values of the second stay)
            assert b.set_register_array("e", e) == 0 and b.set_register_array("f", f) == 0
            for n in range(N):
                oracles[n].set_register("e", float(e[n]))
                oracles[n].set_register("f", float(f[n]))
        y = b.process_block(x[blk * S:(blk + 1) * S])
        if blk == 0:
            kern = b.info("kernel")
            assert {"default": kern >= 9, "unstaged": kern >= 9, "xlate_v256": kern == 15, "asm": 2 <= kern <= 8, "asm_v256": kern == 8, "asm_lds": kern == 1}.get(k, kern == 0), (k, kern)
        flagged = set()
        for n in range(N):
            ref = oracles[n].process_block(x[blk * S:(blk + 1) * S, n].copy())
            assert np.array_equal(bits(ref), bits(y[:, n])), "block %d instance %d" % (blk, n)
            assert b.instruction_counter_i(n) == oracles[n].instruction_counter()
            for r in ("t", "u", "ccr"):
                assert b.get_register_bits_i(r, n) == oracles[n].get_register_bits(r), (blk, n, r)
            if oracles[n].ood_flags():
                flagged.add(n)
        if blk == 0:
            assert not flagged and b.ood_flags() == 0
        else:
            assert flagged >= {3, 64, 70, 71, 131, 199} and not (flagged & {65, 130, 0, 1}), flagged
            assert b.ood_flags() == 8


@pytest.mark.parametrize("trigger", ["delay line hands back |x| > 1", "NaN on the input", "Inf on the input"])
def test_leaving_the_fast_stream_at_the_head_of_a_sample(gpu, trigger, k):
    """A wave leaves the fast stream for the exact one at the *head* of a sample when the PCM input is non-finite or a
    delay-line read issued a sample ahead returns a value outside its row's class.  The program ends in a delay-line read
    that nothing consumes (a wait of its own behind the last instruction): the head's sync point and that wait's used to
    share one slot, and the wave landed behind the program - the whole sample skipped (fuzz seed 502415 with inputs x 3)."""
    text = ("input in 0\noutput out 0\ncontrol c = 0.3\nstatic rd\nstatic xd\nitramsize 7 \nxtramsize 30 \nstatic r4\nstatic r5\nstatic r7\n"
            "static r8\nstatic r9\nstatic r10\nstatic r13\n"
            "xdelay read, xd, at, 0\nxdelay read, xd, at, 0\nxdelay write, in, at, 0\nskip ccr, ccr, 3, 4\nlimit r9, xd, 0, r8\n"
            "macs r5, r13, r7, c\nmacints r9, ccr, 0.125, r5\nandxor r10, in, r4, c\nmacs out, xd, r9, 0.5\nidelay read, rd, at, 0\nend")
    N, S = 70, 40
    x = progs.stimulus(N, S)
    if trigger.startswith("delay"):
        x = x * np.float32(3.0)                 # the inputs go through the delay line into the row `xd`
    elif trigger.startswith("NaN"):
        x[17, ::3] = np.float32(np.nan)
    else:
        x[17, ::3] = np.float32(np.inf)
    check_batch(gpu, text, x, regs=("xd", "rd", "r9", "r10", "out", "ccr"))


def test_delay_line_exact(gpu, k):
    text = "itramsize 5 \n" + HDR + "idelay read, rd, at, 0\nidelay write, in, at, 0\nmacs out, 0, rd, 1.0\nend"
    x = progs.stimulus(66, 64)
    check_batch(gpu, text, x, regs=("rd", "ccr"))
    text = "xtramsize 37 \n" + HDR + "xdelay read, rd, at, 0\nmacs a, in, rd, 0.5\nxdelay write, a, at, 0\nxdelay read, b, at, 0\nxdelay write, in, at, 0\nmacs out, b, rd, 0.5\nend"
    check_batch(gpu, text, x, regs=("rd", "b", "a"))


def test_delay_inside_skip_shadow(gpu, k):
    """A TRAM instruction a lane skips does not advance that lane's cursor (FX8010.cpp:1037: skipped instructions
    are not executed), so cursors differ from lane to lane: the translator must not keep them in SGPRs here."""
    text = ("itramsize 7 \nxtramsize 11 \n" + HDR + "static t\n"
            "idelay read, rd, at, 0\nxdelay read, b, at, 0\n"
            "macs t, in, 0, 0\nskip ccr, ccr, 6, 2\n"
            "idelay write, in, at, 0\nxdelay write, rd, at, 0\n"
            "macs out, rd, b, 0.5\nend")
    x = progs.stimulus(130, 200)
    b, _ = check_batch(gpu, text, x, regs=("rd", "b", "t", "ccr"))
    if k in ("default", "unstaged"):
        assert b.info("xlate_called") >= 4  # the four TRAM instructions run in the interpreter's per-lane handlers


def test_delay_write_offset_and_ood(gpu, k):
    # write offset 3 stays inside the reference's array (wpos+3 < 8192) and lands beyond `size`
    text = "itramsize 8 \n" + HDR + "idelay read, rd, at, 0\nidelay write, in, at, 3\nmacs out, 0, rd, 1.0\nend"
    x = progs.stimulus(64, 40)
    check_batch(gpu, text, x, regs=("rd",))
    # read offset 2 goes negative when the cursor is < 2: outside the parity domain, flagged
    text = "itramsize 8 \n" + HDR + "idelay write, in, at, 0\nidelay read, rd, at, 2\nmacs out, 0, rd, 1.0\nend"
    check_batch(gpu, text, x, regs=("rd",), expect_ood=1)


def test_skip_over_end_multipass(gpu, k):
    # the SKIP can jump over END: the reference re-runs the program with the leftover count
    text = HDR + "macs a, in, 0, 0\nmacs out, out, 0.125, 0.5\nskip ccr, ccr, 6, 2\nmacs b, in, 0.5, 0.5\nend"
    x = progs.stimulus(70, 50)
    b, _ = check_batch(gpu, text, x, regs=("a", "b", "out", "ccr"))
    assert b.info("multipass") == 1
    # the interpreter runs the passes itself (its end-of-sample handler starts the next one for the lanes that skipped END);
    # generated code is one pass, so the default choice is the interpreter too
    kern = b.info("kernel")
    assert {"asm_lds": kern == 1, "asm_v256": kern == 8}.get(k, (kern == 0) if isinstance(k, int) else 2 <= kern <= 8), (k, kern)


def test_skip_counts_per_instance_and_passes_to_the_cap(gpu, k):
    """a SKIP whose count is a per-instance value can skip anything, END included: a multi-pass program for the decoder.  Counts
    0 .. 9 (whole passes are skipped with the count that is left), negative (exactly one instruction), NaN - and a program that
    never reaches END on some instances: 64 passes per sample period, flag 32 on those instances only, like the oracle."""
    text = HDR + "control n = 1\nstatic t\nmacs a, in, 0, 0\nskip ccr, ccr, 6, n\nmacs t, t, 0.125, 0.5\nmacs b, in, 0.5, 0.5\nmacs out, t, b, 0.5\nend"
    N, S = 140, 20
    x = progs.stimulus(N, S)
    counts = np.resize(np.array([0, 1, 2, 3, 4, 5, 6, 7, 9, -1, -7, 2.9, np.nan, 13, 29], np.float32), N)
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    assert b.set_register_array("n", counts) == 0
    y = np.concatenate([b.process_block(x[:7]), b.process_block(x[7:])], axis=0)
    assert b.info("multipass") == 1
    flags = 0
    for n in range(N):
        o = Oracle(1)
        assert o.load_text(text)
        o.set_register("n", float(counts[n]))
        ref = o.process_block(x[:, n].copy())
        assert np.array_equal(bits(ref), bits(y[:, n])), "instance %d (count %r)" % (n, counts[n])
        assert b.instruction_counter_i(n) == o.instruction_counter(), n
        for r in ("t", "b", "a", "ccr"):
            assert b.get_register_bits_i(r, n) == o.get_register_bits(r), (n, r)
        flags |= o.ood_flags()
    assert flags == 32 and b.ood_flags() == 32   # (a count of 4 covers exactly the rest of the program, END included, in every pass)
    # never END: every pass executes the SKIP again, and a count of 2 covers the instruction behind it and END
    forever = HDR + "control n = 2\nmacs a, -0.5, 0, 0\nskip ccr, ccr, 6, n\nmacs out, in, 0, 0\nend"
    b2 = gpu.Batch(N, 1, 0)
    assert b2.load_text(forever), b2.errors()
    lanes = np.where(np.arange(N) % 3 == 0, 2.0, 1.0).astype(np.float32)     # 2: jumps over END in every pass; 1: skips one instruction
    assert b2.set_register_array("n", lanes) == 0
    y2 = b2.process_block(x[:4])
    flagged = 0
    for n in range(N):
        o = Oracle(1)
        assert o.load_text(forever)
        o.set_register("n", float(lanes[n]))
        ref = o.process_block(x[:4, n].copy())
        assert np.array_equal(bits(ref), bits(y2[:, n])), n
        assert b2.instruction_counter_i(n) == o.instruction_counter(), n
        assert (o.ood_flags() == 32) == (n % 3 == 0), (n, o.ood_flags())
        flagged += o.ood_flags() == 32
    assert flagged and b2.ood_flags() == 32


def test_stereo_and_input_channel_quirk(gpu, k):
    # X and Y inputs are read through A's channel (source/FX8010.cpp:1058,1060)
    text = ("input l 0\ninput r 1\noutput ol 0\noutput or 1\nstatic t\n"
            "macs ol, l, r, 0.5\nmacs or, r, l, 0.5\nmacs t, 0, r, 1.0\nmacs or, or, t, 0.25\nend")
    N, S = 70, 33
    x = np.stack([progs.stimulus(N, S), progs.stimulus(N, S, seed=99)], axis=1)  # [S, 2, N]
    check_batch(gpu, text, x, regs=("l", "r", "t", "ccr"), channels=2)


def test_set_register_broadcast_and_per_instance(gpu, k):
    text = progs.config1_shipped()
    N, S = 70, 32
    ramp = np.array([i / 16.0 for i in range(-16, 16)], dtype=np.float32)
    x = np.repeat(ramp[:, None], N, axis=1)
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text)
    o = [Oracle(1) for _ in range(N)]
    for q in o:
        assert q.load_text(text)
    ys, refs = [], [[] for _ in range(N)]
    for blk, v in enumerate((0.1, 0.25, 0.5, 1.0)):  # the reference harness's slider schedule (main.cpp:80,107-114)
        assert b.set_register("volume", v) == 0
        if blk == 2:
            assert b.set_register_i("volume", 5, 0.75) == 0
            o[5].set_register("volume", 0.75)
        for n, q in enumerate(o):
            if not (blk == 2 and n == 5):
                q.set_register("volume", v)
        ys.append(b.process_block(x[blk * 8:(blk + 1) * 8]))
        for n, q in enumerate(o):
            refs[n].append(q.process_block(x[blk * 8:(blk + 1) * 8, n].copy()))
    y = np.concatenate(ys, axis=0)
    for n in range(N):
        assert np.array_equal(bits(np.concatenate(refs[n])), bits(y[:, n])), n
        assert b.get_register_bits_i("volume", n) == o[n].get_register_bits("volume")
    assert b.set_register("nonexistent", 1.0) == 1
    assert b.get_register_i("nonexistent", 0) == 1.0


@pytest.mark.parametrize("builder", [True, False], ids=["builder_thread", "callers_thread_only"])
def test_moving_controls_become_rows_once(gpu, monkeypatch, builder):
    """The reference's setRegisterValue is a store (source/FX8010.cpp:236-253), called every 8 samples by its harness
    (source/main.cpp:107-114).  Here a control starts out compiled into the generated code; the first change after the program
    has run gives the declared controls - all of them: a host that moves one moves others - rows of the register file, and
    every later change of any of them is a fill of its row: the translated tier runs every block of a slider sweep.  The
    variant with the rows is generated ahead of time on the handle's builder thread, so the first touch is a pointer swap;
    without that thread (FX_BUILDER=0) it is ONE re-translation on the caller's.  Results are the reference's throughout."""
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    monkeypatch.delenv("FX_BUILDER", raising=False)
    if not builder:
        monkeypatch.setenv("FX_BUILDER", "0")
    text = HDR + "control mix = 0.25\ncontrol unused = 0.5\nmacs a, a, vol, in\ninterp b, b, vol, a\nmacs out, b, a, mix\nend"
    N = 70
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    o = Oracle(1)
    assert o.load_text(text)
    x = progs.stimulus(N, 8 * 40)
    tiers, builds, rows = [], [], []
    for blk in range(40):
        if 3 <= blk < 30:
            v = float(np.float32(0.1 + 0.03 * blk))
            b.set_register("vol", v)
            o.set_register("vol", v)
        if blk in (10, 11, 25):
            b.set_register("mix", 0.5 - 0.01 * blk)
            o.set_register("mix", 0.5 - 0.01 * blk)
            b.set_register("unused", 0.01 * blk)   # a control no instruction reads: its value lives in the state row only
        if blk == 20:   # non-finite values through a moving control: the row is checked like any state row
            b.set_register("vol", float("inf"))
            o.set_register("vol", float("inf"))
        xs = x[8 * blk:8 * blk + 8]
        y = b.process_block(xs)
        ref = o.process_block(xs[:, 5].copy())
        assert np.array_equal(bits(ref), bits(y[:, 5])), "block %d" % blk
        tiers.append(b.info("kernel"))
        builds.append(b.info("xlate_builds"))
        rows.append(b.info("num_rows"))
    assert all(t >= 9 for t in tiers), tiers                 # the translated tier on every block
    if builder:
        # (the variant with the panel in rows, and - while only `vol` has moved - the lean one with `mix` folded back in:
        # both from the builder thread, each adopted by a pointer swap; tests/test_gpu_boundary.py has the lean variant's own test)
        assert builds[-1] == 1 and b.info("xlate_background_builds") in (1, 2) and 1 <= b.info("code_cache_hits") <= 3, (builds, b.info("code_cache_hits"))
    else:
        assert builds[2] == 1 and builds[3] == 2 and builds[-1] == 2, builds   # the first change of vol: one more translation (mix joins it), then none
    assert rows[3] == rows[2] + 2 and rows[-1] == rows[3], rows   # vol and mix; `unused` never gets a row
    assert b.get_register_i("unused", 7) == np.float32(0.25)
    assert b.instruction_counter_i(5) == o.instruction_counter()
    for n in (0, 63, 69):
        assert b.get_register_bits_i("vol", n) == o.get_register_bits("vol") and b.get_register_bits_i("b", n) is not None


def test_rows_do_not_pile_up(gpu, monkeypatch):
    """a per-instance write gives a register a row; a broadcast write after it gives the row back (every instance holds the same
    value again) - coming back to code that was generated before is a swap; registers no instruction reads take per-instance
    values without ever getting a row"""
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    text = HDR + "static g = 0.5\nstatic memo = 0.125\nmacs a, a, g, in\nmacs out, a, vol, 0.5\nend"
    N = 70
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    oracles = [Oracle(1) for _ in range(N)]
    for o in oracles:
        assert o.load_text(text)
    x = progs.stimulus(N, 8 * 12)
    rows = []
    for blk in range(12):
        if blk == 2:
            vals = np.linspace(0.1, 0.9, N).astype(np.float32)
            b.set_register_array("g", vals)
            b.set_register_array("memo", vals)
            for n, o in enumerate(oracles):
                o.set_register("g", float(vals[n]))
                o.set_register("memo", float(vals[n]))
        if blk == 5:
            b.set_register("g", 0.5)
            for o in oracles:
                o.set_register("g", 0.5)
        if blk == 8:
            b.set_register_i("g", 3, 0.75)
            oracles[3].set_register("g", 0.75)
        xs = x[8 * blk:8 * blk + 8]
        y = b.process_block(xs)
        for n in (0, 3, 64, 69):
            assert np.array_equal(bits(oracles[n].process_block(xs[:, n].copy())), bits(y[:, n])), (blk, n)
        rows.append(b.info("num_rows"))
    assert rows[2] == rows[1] + 1 and rows[5] == rows[1] and rows[8] == rows[2], rows
    assert b.info("code_cache_hits") >= 2   # back to the code of blocks 0-1 at block 5, to that of blocks 2-4 at block 8
    assert b.get_register_bits_i("memo", 69) == oracles[69].get_register_bits("memo") and b.get_register_bits_i("memo", 0) == oracles[0].get_register_bits("memo")


def test_controls_that_shape_the_code_are_compiled_in(gpu, monkeypatch):
    """a SKIP's count, a LOG / EXP table number, a delay-line offset given as a control: changing them re-lowers (the
    interpreter tier runs the blocks while such a control keeps moving, the translation comes back when it rests)"""
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    text = HDR + "control n = 1\ncontrol e = 3\nmacs a, in, 0, 0\nskip ccr, ccr, 6, n\nmacs out, 0, in, 1.0\nmacs b, out, 0.5, 0.5\nlog out, b, e, 0\nend"
    N = 70
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    o = Oracle(1)
    assert o.load_text(text)
    x = progs.stimulus(N, 8 * 24)
    tiers = []
    for blk in range(24):
        if blk in (3, 4, 5):
            b.set_register("n", float(blk % 3))
            o.set_register("n", float(blk % 3))
        if blk == 6:
            b.set_register("e", 7.0)
            o.set_register("e", 7.0)
        xs = x[8 * blk:8 * blk + 8]
        y = b.process_block(xs)
        ref = o.process_block(xs[:, 5].copy())
        assert np.array_equal(bits(ref), bits(y[:, 5])), "block %d" % blk
        tiers.append(b.info("kernel"))
    assert tiers[0] >= 9 and tiers[-1] >= 9 and all(t >= 2 for t in tiers)
    assert b.instruction_counter_i(5) == o.instruction_counter()


def test_register_arrays_per_instance_automation(gpu, k):
    """fxb_set_register_array: one control value per instance before each block - N reference objects whose
    caller moves a different slider on each (main.cpp:107-114 per object)."""
    text = HDR + "macs a, a, vol, in\ninterp b, b, vol, a\nmacs out, b, a, 0.5\nend"
    N, S, blocks = 96, 16, 4
    x = progs.stimulus(N, S * blocks)
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    oracles = []
    for n in range(N):
        o = Oracle(1)
        assert o.load_text(text)
        oracles.append(o)
    rng = np.random.default_rng(5)
    for blk in range(blocks):
        vols = rng.uniform(0.0, 1.0, size=N).astype(np.float32)
        assert b.set_register_array("vol", vols) == 0
        y = b.process_block(x[blk * S:(blk + 1) * S])
        for n in range(N):
            oracles[n].set_register("vol", float(vols[n]))
            ref = oracles[n].process_block(x[blk * S:(blk + 1) * S, n].copy())
            assert np.array_equal(bits(ref), bits(y[:, n])), "block %d instance %d" % (blk, n)
        assert np.array_equal(bits(b.get_register_array("vol")), bits(vols))
    got = b.get_register_array("a")
    for n in range(N):
        assert int(bits(got[n:n + 1])[0]) == oracles[n].get_register_bits("a")
    assert b.set_register_array("nosuch", np.zeros(N, dtype=np.float32)) == 1


def test_programs_beyond_the_translator_fall_back(gpu, monkeypatch):
    """A program whose translation does not fit the template's code hole runs on the interpreter, one whose
    register file exceeds 224 rows on the interpreter's LDS build - same results."""
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    rng = np.random.default_rng(11)
    regs = ["r%d" % i for i in range(40)]
    body = []
    for i in range(6000):
        a, b_, c = rng.choice(regs, 3)
        body.append(("macs %s, %s, %s, 0.37" if i % 3 else "interp %s, %s, 0.3, %s") % (a, b_, c))
    big = HDR + "".join("static %s\n" % r for r in regs) + "macs r0, in, 0.5, 0.5\n" + "\n".join(body) + "\nmacs out, r1, r2, 0.5\nend"
    x = progs.stimulus(66, 6)
    b, _ = check_batch(gpu, big, x, regs=("r3", "r17", "out"), instances=[0, 65])
    assert 2 <= b.info("kernel") <= 8  # interpreter, VGPR register file
    wide_regs = ["w%d" % i for i in range(240)]
    wide = HDR + "".join("static %s\n" % r for r in wide_regs) + "macs w0, in, 0.5, 0.5\n"
    wide += "".join("macs w%d, w%d, w%d, 0.9\n" % (i, i - 1, (i * 7) % 240) for i in range(1, 240)) + "macs out, w239, w100, 0.5\nend"
    b, _ = check_batch(gpu, wide, x, regs=("w239", "out"), instances=[0, 65])
    assert b.info("kernel") == 1  # interpreter, LDS register file


@pytest.mark.parametrize("chunk", range(6))
def test_random_programs(gpu, chunk, monkeypatch):
    """differential fuzz on the default tier: random in-domain programs over all opcodes, SKIPs of every CCR value
    and count, `ccr` / `noise` / literals as operands, delay lines - every instance against the oracle"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_fuzz
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    N, S = 70, 12
    x = progs.stimulus(N, S)
    for seed in range(chunk * 16, chunk * 16 + 16):
        rng = np.random.default_rng(91000 + seed)
        gen = stress_fuzz.random_program2 if seed % 2 else stress_fuzz.random_program
        text = gen(rng, int(rng.integers(6, 90)), int(rng.integers(3, 40)))
        b = gpu.Batch(N, 1, 0)
        assert b.load_text(text), b.errors()
        y = b.process_block(x)
        y2 = b.process_block(x)
        for n in range(0, N, 3):
            o = Oracle(1)
            assert o.load_text(text)
            r1 = o.process_block(x[:, n].copy())
            r2 = o.process_block(x[:, n].copy())
            if o.ood_flags():
                continue
            assert same_with_nan(r1, y[:, n]) and same_with_nan(r2, y2[:, n]), "seed %d instance %d (kernel %d)\n%s" % (seed, n, b.info("kernel"), text)
            assert b.instruction_counter_i(n) == o.instruction_counter(), "seed %d instance %d counter" % (seed, n)
            rb, gb = o.get_register_bits("ccr"), b.get_register_bits_i("ccr", n)
            assert rb == gb, "seed %d instance %d ccr %08x %08x" % (seed, n, rb, gb)


def test_noise_seed_per_instance(gpu, k):
    text = HDR + "macs out, 0, noise, 1.0\nend"
    N, S = 66, 40
    x = np.zeros((S, N), dtype=np.float32)
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text)
    b.seed_noise_i(3, 12345, -99999)
    y = b.process_block(x)
    for n in (0, 3, 65):
        o = Oracle(1)
        assert o.load_text(text)
        if n == 3:
            o.seed_noise(12345, -99999)
        assert np.array_equal(bits(o.process_block(x[:, n].copy())), bits(y[:, n]))


def test_single_instance_api_matches_reference_harness(gpu):
    """fx_* mirror: the reference's own main.cpp loop (slider every 8 samples, bipolar ramp)."""
    import os
    import tempfile

    text = progs.config1_shipped()
    fd, path = tempfile.mkstemp(suffix=".da")
    with os.fdopen(fd, "wb") as fh:
        fh.write(text.encode())
    try:
        fx = gpu.Single(1)
        assert fx.load_file(path)
        o = Oracle(1)
        assert o.load_file(path)
        ramp = np.array([i / 16.0 for i in range(-16, 16)], dtype=np.float32)
        sliders = [0.1, 0.25, 0.5, 1.0]
        for i in range(32):
            if i % 8 == 0:
                assert fx.set_register("volume", sliders[i // 8]) == 0
                o.set_register("volume", sliders[i // 8])
            got = fx.process(ramp[i:i + 1])
            ref = o.process_block(ramp[i:i + 1])
            assert bits(got)[0] == bits(ref)[0], i
        assert fx.instruction_counter() == o.instruction_counter() == 64
        assert fx.get_register("filter_cutoff") == o.get_register("filter_cutoff")
        assert fx.controls() == o.controls() and fx.meta() == o.meta() and fx.errors() == o.errors()
    finally:
        os.unlink(path)


def test_load_failure_reports_errors(gpu):
    b = gpu.Batch(64, 1, 0)
    assert not b.load_text("static a\nmacs a, b, 0, 0\nend")
    assert b.errors()[1] == ("Variable nicht deklariert", 2)
    with pytest.raises(RuntimeError):
        b.process_block(np.zeros((4, 64), dtype=np.float32))


def test_cpp_drop_in_class_harness(gpu):
    """host/FX8010.h (the reference's class surface over the C ABI) through its console harness."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "fx8010-emulator-core_amd")
    subprocess.check_call(["make", "-s", "-C", os.path.join(pkg, "csrc"), "demo"])
    prog = os.path.join(pkg, "programs", "config1_shipped.da")
    out = subprocess.run([os.path.join(pkg, "host", "fx8010_demo"), prog, "4096", "300"], stdout=subprocess.PIPE, text=True, check=True).stdout
    lines = out.strip().split("\n")
    pairs = [tuple(float(v) for v in l.split(",")) for l in lines[:32]]
    o = Oracle(1)
    assert o.load_file(prog)
    sliders = [0.1, 0.25, 0.5, 1.0]
    for i, (xin, yout) in enumerate(pairs):
        if i % 8 == 0:
            o.set_register("volume", sliders[i // 8])
        ref = o.process_block(np.array([xin], dtype=np.float32))[0]
        assert abs(ref - yout) <= 1e-6 * max(1.0, abs(ref)), i  # printed with 6 significant digits
    assert "64 instructions" in out and "emulated MIPS" in out and "control: volume" in out
    # the third part: the reference's real-time question (source/main.cpp:155) asked of the batch, PCM in pinned buffers (in place)
    rt = [l for l in lines if l.startswith("real time:")]
    assert len(rt) == 1 and "4096 instances, 300 blocks of 32 samples" in rt[0] and " 0 late" in rt[0], out[-600:]


def test_wav_front_end(gpu, tmp_path):
    """host/fx8010_wav.cpp: a 16-bit stereo WAV through 5 instances with a swept control; the picked instance's float
    WAV equals the oracle run on s/32768 with that instance's control value."""
    import os
    import struct
    import subprocess
    import wave

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "fx8010-emulator-core_amd")
    subprocess.check_call(["make", "-s", "-C", os.path.join(pkg, "csrc"), "wav"])
    text = ("input l 0\ninput r 1\noutput ol 0\noutput or 1\ncontrol gain = 0.5\nstatic s\nitramsize 7 \nstatic rd\n"
            "idelay read, rd, at, 0\nmacs s, l, r, gain\ninterp ol, ol, 0.25, s\nmacs or, rd, r, gain\nidelay write, s, at, 0\nend")
    prog = tmp_path / "p.da"
    prog.write_text(text)
    rng = np.random.default_rng(3)
    frames = 5000
    pcm = rng.integers(-32768, 32767, size=(frames, 2), dtype=np.int16)
    with wave.open(str(tmp_path / "in.wav"), "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(48000)
        w.writeframes(pcm.tobytes())
    out = subprocess.run([os.path.join(pkg, "host", "fx8010_wav"), str(prog), str(tmp_path / "in.wav"), str(tmp_path / "out.wav"),
                          "--instances", "5", "--pick", "3", "--block", "1024", "--sweep", "gain=0.1:0.9"],
                         stdout=subprocess.PIPE, text=True, check=True).stdout
    assert "5000 frames x 2 channel(s) x 5 instance(s)" in out
    raw = (tmp_path / "out.wav").read_bytes()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"WAVE" and struct.unpack("<H", raw[20:22])[0] == 3
    y = np.frombuffer(raw[44:], dtype=np.float32).reshape(frames, 2)
    o = Oracle(2)
    assert o.load_text(text), o.errors()
    o.set_register("gain", float(np.float32(0.1) + (np.float32(0.9) - np.float32(0.1)) * np.float32(3) / np.float32(4)))
    x = pcm.astype(np.float32) / np.float32(32768.0)
    ref = o.process_block(x)
    assert np.array_equal(bits(ref), bits(y))


@pytest.mark.parametrize("name", ["config3", "config4", "config5"])
def test_full_size_shard(gpu, name, monkeypatch):
    """BASELINE's per-GPU instance counts (config3: 65 536, config4 / config5: 262 144), less a few so that the last
    wavefront is ragged: sampled instances across the whole shard against the oracle over three blocks, and a
    size-independent property - instances fed the same PCM produce the same words wherever they sit in the batch
    (first wavefront, middle, ragged end of the last one).  config3's blocks are long enough for its 1000-sample delay
    line to wrap (reads issued one sample ahead across block ends)."""
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    N, S, blocks = (65536 - 37, 400, 3) if name == "config3" else (262144 - 37, 24, 3)
    text = progs.CONFIGS[name]()
    x = progs.stimulus(N, S * blocks).copy()
    twins = [5, 64 * 500 + 63, N // 2 + 1, N - 1]
    for t in twins[1:]:
        x[:, t] = x[:, twins[0]]
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    ys = [b.process_block(x[i * S:(i + 1) * S]) for i in range(blocks)]
    y = np.concatenate(ys, axis=0)
    for t in twins[1:]:
        assert np.array_equal(bits(y[:, t]), bits(y[:, twins[0]])), "instance %d differs from its twin" % t
    total = 0
    for n in (0, 63, 64, 4097, min(99999, N - 200), N // 2, N - 65, N - 2):
        o = Oracle(1)
        assert o.load_text(text)
        ref = np.concatenate([o.process_block(x[i * S:(i + 1) * S, n].copy()) for i in range(blocks)])
        assert np.array_equal(bits(ref), bits(y[:, n])), "instance %d" % n
        assert b.instruction_counter_i(n) == o.instruction_counter()
        total += 1
    assert b.ood_flags() == 0
    if name in ("config3", "config5"):  # no SKIP: every instance executes every instruction
        assert b.instruction_counter() == N * progs.count_instructions(text) * S * blocks


def test_config5_feedback_in_uneven_blocks(gpu, monkeypatch):
    """the headline program past its delay line's first read-back: 4 reads + 4 writes per sample walk the 8192-slot line in
    2048 samples (cursors advance per executed TRAM instruction, reference source/FX8010.cpp:909-967, 1188-1211), so only runs
    longer than that see written data again.  4 100 instances fed from the host over 2 250 samples in uneven blocks - ends at,
    one before and one behind the wrap, blocks of one and two samples - against the oracle; tests/golden/configs_long.json
    pins the same stretch with the reference's own words (test_gpu_golden.py)"""
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    text = progs.config5()
    N, cuts = 4100, [0, 700, 1399, 2047, 2048, 2049, 2051, 2188, 2250]
    x = progs.stimulus(N, cuts[-1])
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    y = np.concatenate([b.process_block(x[lo:hi]) for lo, hi in zip(cuts[:-1], cuts[1:])], axis=0)
    assert b.info("kernel") >= 9
    for n in (0, 1, 63, 64, 2049, 4031, 4032, 4095, 4096, N - 1):
        o = Oracle(1)
        assert o.load_text(text)
        ref = o.process_block(x[:, n].copy())
        bad = np.nonzero(bits(ref) != bits(y[:, n]))[0]
        assert bad.size == 0, "instance %d: first mismatch at sample %d" % (n, bad[0])
        assert b.instruction_counter_i(n) == o.instruction_counter()
        for r in ("d0", "d1", "d2", "d3", "m", "w0", "w3", "lp2", "ccr"):
            assert b.get_register_bits_i(r, n) == o.get_register_bits(r), (r, n)
        assert o.get_register_bits("d0") & 0x7fffffff, "the line never handed a written word back"
    assert b.ood_flags() == 0
    assert b.instruction_counter() == N * progs.count_instructions(text) * cuts[-1]


def test_config5_full_shard_beyond_the_delay_line(gpu, monkeypatch):
    """bench.py's workload as a test: the per-GPU shard of BASELINE configs[4] (262 144 instances, less 37: a ragged last
    wavefront), PCM resident in HBM and generated there (bench.py device_stimulus), 2 304 samples in three launches - the
    delay lines (8 GiB) wrap at sample 2048.  Instances spread over the shard against the oracle, bit for bit, and the
    size-independent property: instances fed the same PCM produce the same words wherever they sit"""
    import torch

    import bench
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    text = progs.config5()
    N, cuts = 262144 - 37, [0, 1500, 2050, 2304]
    S = cuts[-1]
    dev = torch.device("cuda", 0)
    x = bench.device_stimulus(torch, N, S, 0, dev)
    twins = [5, 64 * 500 + 63, N // 2 + 1, N - 1]
    for t in twins[1:]:
        x[:, t] = x[:, twins[0]]
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        b.process_block_dev(x[lo:hi].data_ptr(), y[lo:hi].data_ptr(), hi - lo)
    b.sync()
    assert b.info("kernel") >= 9 and b.info("waves_per_wg") == 1
    picks = sorted(set([0, 63, 64, 4097, N - 65, N - 2] + [int(v) for v in np.linspace(0, N - 1, 40)]) - set(twins[1:]))
    cols = torch.tensor(picks + twins, device=dev)
    xs = x[:, cols].cpu().numpy()
    ys = y[:, cols].cpu().numpy()
    for j, t in enumerate(twins[1:], start=len(picks) + 1):
        assert np.array_equal(bits(ys[:, j]), bits(ys[:, len(picks)])), "instance %d differs from its twin" % t
    for j, n in enumerate(picks):
        assert np.array_equal(bits(xs[:, j]), bits(progs.stimulus(1, S, first_instance=n)[:, 0])), "device stimulus differs from the host's"
        o = Oracle(1)
        assert o.load_text(text)
        ref = o.process_block(xs[:, j].copy())
        bad = np.nonzero(bits(ref) != bits(ys[:, j]))[0]
        assert bad.size == 0, "instance %d: first mismatch at sample %d" % (n, bad[0])
        assert b.instruction_counter_i(n) == o.instruction_counter()
        assert b.get_register_bits_i("d3", n) == o.get_register_bits("d3") and o.get_register_bits("d3") & 0x7fffffff
    assert b.ood_flags() == 0
    assert b.instruction_counter() == N * progs.count_instructions(text) * S
    del b, x, y
    torch.cuda.empty_cache()


def test_config5_all_instances_on_one_gpu(gpu, monkeypatch):
    """BASELINE configs[4] itself on ONE MI355X - what bench.py times at --gpus 1 (strong scaling: the whole job on the one GPU):
    2 097 152 instances (less 37: a ragged last wavefront) of the 512-instruction reverb, 64 GiB of xTRAM, 32 rounds of
    wavefronts per SIMD in one launch of the same code the 1/8 shard runs.  PCM generated in HBM (bench.py device_stimulus),
    2 304 samples in three launches (the delay lines wrap at sample 2048).  48 instances spread over the job - first / last
    lanes of a wavefront, the ragged tail, both sides of every 262 144-instance shard boundary of the 8-GPU split - against the
    oracle bit for bit with their instruction counters, twins fed the same PCM at far-apart places, and the exact
    instruction total of the job (no SKIP in this program: every instance executes every instruction)"""
    import torch

    import bench
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    text = progs.config5()
    N, cuts = progs.CONFIG_TOTAL_INSTANCES["config5"] - 37, [0, 1500, 2050, 2304]
    S = cuts[-1]
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info(dev)
    need = 2 * N * S * 4 + N * 8192 * 4 + (8 << 30)
    assert free > need, "this test needs %.0f GiB of HBM, %.0f are free" % (need / 2 ** 30, free / 2 ** 30)
    x = bench.device_stimulus(torch, N, S, 0, dev)
    twins = [7, 64 * 4097 + 63, 3 * 262144 + 1, N - 1]
    for t in twins[1:]:
        x[:, t] = x[:, twins[0]]
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        b.process_block_dev(x[lo:hi].data_ptr(), y[lo:hi].data_ptr(), hi - lo)
    b.sync()
    assert b.info("kernel") >= 9 and b.info("waves_per_wg") == 1
    assert b.info("xtram_slots") == 8192 and b.info("grid") == (N + 63) // 64
    edges = [k * 262144 + d for k in range(1, 8) for d in (-1, 0)]
    picks = sorted(set([0, 63, 64, 4097, N - 65, N - 2] + edges + [int(v) for v in np.linspace(0, N - 1, 30)]) - set(twins[1:]))
    assert len(picks) >= 48
    cols = torch.tensor(picks + twins, device=dev)
    xs = x[:, cols].cpu().numpy()
    ys = y[:, cols].cpu().numpy()
    for j, t in enumerate(twins[1:], start=len(picks) + 1):
        assert np.array_equal(bits(ys[:, j]), bits(ys[:, len(picks)])), "instance %d differs from its twin" % t
    for j, n in enumerate(picks):
        assert np.array_equal(bits(xs[:, j]), bits(progs.stimulus(1, S, first_instance=n)[:, 0])), "device stimulus differs from the host's"
        o = Oracle(1)
        assert o.load_text(text)
        ref = o.process_block(xs[:, j].copy())
        bad = np.nonzero(bits(ref) != bits(ys[:, j]))[0]
        assert bad.size == 0, "instance %d: first mismatch at sample %d" % (n, bad[0])
        assert b.instruction_counter_i(n) == o.instruction_counter()
        assert b.get_register_bits_i("d3", n) == o.get_register_bits("d3") and o.get_register_bits("d3") & 0x7fffffff
    assert b.ood_flags() == 0
    assert b.instruction_counter() == N * progs.count_instructions(text) * S
    del b, x, y
    torch.cuda.empty_cache()


@pytest.mark.parametrize("seed", range(24))
def test_random_api_sequences(gpu, seed, monkeypatch):
    """Random sequences of the calls a host makes between blocks - block lengths from 1 sample up, broadcast control
    changes (they re-lower, defer the translation, bring it back), per-instance and whole-array register writes,
    register reads - mirrored on one oracle per checked instance.  Exercises every tier transition with live state."""
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    rng = np.random.default_rng(4200 + seed)
    text = ("itramsize 9 \nxtramsize 21 \n" + HDR + "control mix = 0.25\nstatic t\nstatic u\n"
            "idelay read, rd, at, 0\nxdelay read, t, at, 0\n"
            "macs a, a, vol, in\ninterp b, b, mix, a\nmacsn u, t, rd, mix\n"
            + ("log u, u, 3, 0\n" if seed % 2 else "macs u, u, b, 0.5\n")
            + ("macs t, u, 0, 0\nskip ccr, ccr, 6, 1\nmacs u, u, noise, 0.125\n" if seed % 3 == 0 else "")
            + "idelay write, b, at, 0\nxdelay write, u, at, 0\nmacs out, u, b, vol\nend")
    N = 150
    check = [0, 63, 64, 149]
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    oracles = {}
    for n in check:
        o = Oracle(1)
        assert o.load_text(text)
        oracles[n] = o
    pos = 0
    x = progs.stimulus(N, 2000)
    tiers = set()
    for step in range(40):
        op = rng.integers(0, 10)
        if op < 5:
            S = int(rng.choice([1, 2, 7, 8, 33, 64]))
            xs = x[pos:pos + S]
            pos += S
            y = b.process_block(xs)
            tiers.add(b.info("kernel"))
            for n in check:
                ref = oracles[n].process_block(xs[:, n].copy())
                assert np.array_equal(bits(ref), bits(y[:, n])), "seed %d step %d instance %d (kernel %d)" % (seed, step, n, b.info("kernel"))
        elif op < 7:
            name, v = str(rng.choice(["vol", "mix"])), float(np.float32(rng.uniform(0.0, 1.0)))
            assert b.set_register(name, v) == 0
            for o in oracles.values():
                o.set_register(name, v)
        elif op < 8:
            name, n, v = str(rng.choice(["a", "vol", "u"])), int(rng.choice(check)), float(np.float32(rng.uniform(-1.0, 1.0)))
            assert b.set_register_i(name, n, v) == 0
            oracles[n].set_register(name, v)
        elif op < 9:
            vals = rng.uniform(0.0, 1.0, size=N).astype(np.float32)
            assert b.set_register_array("mix", vals) == 0
            for n in check:
                oracles[n].set_register("mix", float(vals[n]))
        else:
            for n in check:
                for r in ("a", "b", "u", "t", "rd", "vol", "mix", "ccr"):
                    assert b.get_register_bits_i(r, n) == oracles[n].get_register_bits(r), "seed %d step %d %s[%d]" % (seed, step, r, n)
    for n in check:
        assert b.instruction_counter_i(n) == oracles[n].instruction_counter()
    assert b.ood_flags() == 0
    assert len(tiers) >= 1


@pytest.mark.parametrize("seed", range(40))
def test_product_cache_patterns(gpu, seed, monkeypatch):
    import os
    import sys
    monkeypatch.delenv("FX_KERNEL", raising=False)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_fuzz
    rng = np.random.default_rng(4200 + seed)
    text = stress_fuzz.product_cache_program(rng, int(rng.integers(6, 60)))
    N, S = 70, 40
    x = (rng.uniform(-1.0, 1.0, size=(S, N)) * rng.choice([1.0, 0.5, 1e-3], size=(1, N))).astype(np.float32)
    b, _ = check_batch(gpu, text, x, regs=("a", "b", "c", "t", "out", "ccr"))
    assert b.info("kernel") >= 9


@pytest.mark.parametrize("kernel", ["default", "asm", "hip"])
def test_maximum_delay_lines_wrap(gpu, kernel, monkeypatch):
    """the largest lines the reference has room for - smallDelayBuffer[8192], largeDelayBuffer[1048576] (FX8010.h:210-211) - each
    with a feedback loop through it, run past the first wrap of the large one (sample 1 048 576): 256 MiB of delay memory per
    wavefront, three instances against the oracle, positions and the slots around the wrap compared directly"""
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    if kernel != "default":
        monkeypatch.setenv("FX_KERNEL", kernel)
    text = ("itramsize 8192 \nxtramsize 1048576 \ninput in 0\noutput out 0\nstatic rd\nstatic xd\nstatic a\nstatic b\n"
            "idelay read, rd, at, 0\nxdelay read, xd, at, 0\nmacs a, in, rd, 0.5\nmacs b, in, xd, 0.25\nidelay write, a, at, 0\nxdelay write, b, at, 0\n"
            "macs out, a, b, 0.5\nend")
    N, S, piece = 70, 1048576 + 2048, 131072
    rng = np.random.default_rng(4)
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    watch = (0, 37, 69)
    oracles = {n: Oracle(1) for n in watch}
    for o in oracles.values():
        assert o.load_text(text)
    for at in range(0, S, piece):
        ns = min(piece, S - at)
        x = rng.uniform(-0.9, 0.9, size=(ns, N)).astype(np.float32)
        y = b.process_block(x)
        for n, o in oracles.items():
            ref = o.process_block(x[:, n].copy())
            assert np.array_equal(bits(ref), bits(y[:, n])), (kernel, at, n)
    assert b.ood_flags() == 0
    for n, o in oracles.items():
        assert b.get_cursors_i(n) == o.cursors() == [2048 % 8192, 2048 % 8192, 2048, 2048], (n, b.get_cursors_i(n), o.cursors())
        assert b.instruction_counter_i(n) == o.instruction_counter()
        assert np.array_equal(bits(b.get_tram_i(1, n, 1048576)), bits(o.tram(1, 1048576))), n
        assert np.array_equal(bits(b.get_tram_i(0, n, 8192)), bits(o.tram(0, 8192))), n


def test_empty_blocks_and_single_instances(gpu, k):
    """a block of no samples changes nothing (the reference's caller simply does not call process()); a batch of ONE instance is
    one reference object; 63 / 64 / 65 instances: a wavefront one lane short, full, and one lane into the next"""
    text = HDR + "itramsize 5 \nidelay read, rd, at, 0\nmacs a, in, rd, 0.5\nidelay write, a, at, 0\nmacs out, a, noise, 0.25\nend"
    for N in (1, 63, 64, 65):
        x = progs.stimulus(N, 24)
        b = gpu.Batch(N, 1, 0)
        assert b.load_text(text), b.errors()
        y0 = b.process_block(x[:0])
        assert y0.shape == (0, N) and b.instruction_counter() == 0
        y = np.concatenate([b.process_block(x[:9]), b.process_block(x[9:9]), b.process_block(x[9:])], axis=0)
        for n in sorted({0, N // 2, N - 1}):
            o = Oracle(1)
            assert o.load_text(text)
            ref = o.process_block(x[:, n].copy())
            assert np.array_equal(bits(ref), bits(y[:, n])), (N, n)
            assert b.instruction_counter_i(n) == o.instruction_counter() and b.get_cursors_i(n) == o.cursors()
        assert b.instruction_counter() == N * 24 * 5
