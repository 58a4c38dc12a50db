"""Opt-in DANE delay-line model (FX_OPT_TRAM_DANE / FX_OPT_TRAM_ADDR_SHIFT, include/fx8010_amd.h): NOT reference behaviour
(the reference's own design note docs/TRAM Registermapping.pdf describes it; its code defines offset 0 only), so there
is no golden vector from the reference - the checker is oracle/ with the same switch, pinned here by the delay-line
properties themselves (CPU tests), and the GPU path is compared with it bit for bit (gpu tests).  With the options off
nothing changes: every other test in this directory runs with them off."""
import numpy as np
import pytest

import fx8010_programs as progs
from pyoracle import Oracle

OPT_DANE, OPT_SHIFT, OPT_INTERP = 1, 2, 4

TWO_TAPS = """itramsize 64 
xtramsize 500 
input in 0
output out 0
static w
static r1
static r2
static xr
idelay write, w, at, 0
idelay read, r1, at, 7
idelay read, r2, at, 19
xdelay write, in, at, 3
xdelay read, xr, at, 403
macs w, in, 0, 0
macs out, r1, r2, 0.5
end"""

CHORUS = """itramsize 2880 
input in 0
output out 0
static wrt
static rd1
static rd2
static lfo = 0.25
static cosv = 0.9
static t
control speed = 0.3
control depth = 0.0008
idelay write, in, at, 0
idelay read, rd1, at, 1439
idelay read, rd2, at, 700
macs lfo, lfo, speed, cosv
macsn cosv, cosv, speed, lfo
macs &rd1, 0.0013, lfo, depth
macs t, rd1, rd2, 0.5
macs out, 0, t, 0.5
end"""


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_taps_are_delays_of_their_position_difference():
    """a value written at position pw comes back pr - pw samples later at position pr; any number of taps per line"""
    o = Oracle(1)
    o.set_option(OPT_DANE)
    assert o.load_text(TWO_TAPS), o.errors()
    x = np.zeros(600, np.float32)
    x[0], x[5] = 1.0, -0.5
    y = o.process_block(x)
    # w = in one instruction after the write tap: the impulse enters the line at sample 1; taps at 7 and 19
    want = np.zeros(600, np.float32)
    for t0, v in ((0, 1.0), (5, -0.5)):
        want[t0 + 1 + 7] += v
        want[t0 + 1 + 19] += 0.5 * v
    assert np.array_equal(bits(y), bits(want))
    assert o.get_register_bits("&r1") == bits(np.float32(7.0))[()] and o.get_register_bits("&r2") == bits(np.float32(19.0))[()]
    assert o.ood_flags() == 0
    # the xTRAM line: written at 3, read at 403: 400 samples
    o2 = Oracle(1)
    o2.set_option(OPT_DANE)
    assert o2.load_text(TWO_TAPS.replace("macs out, r1, r2, 0.5", "macs out, 0, xr, 1.0"))
    y2 = o2.process_block(x)
    assert y2[400] == 1.0 and y2[405] == -0.5 and np.count_nonzero(y2) == 2


def test_reference_mode_rejects_the_tap_register_syntax_and_keeps_its_cursor_model():
    o = Oracle(1)
    assert not o.load_text(CHORUS)  # '&' is not an operand character of the reference dialect
    o = Oracle(1)
    assert o.load_text(TWO_TAPS)    # the same text IS a reference program: cursor per executed instruction, offsets flagged
    o.process_block(np.ones(30, np.float32))
    assert o.ood_flags() != 0       # read offsets > cursor: outside the parity domain there


def test_address_shift_positions_are_fixed_point_fractions():
    """with FX_OPT_TRAM_ADDR_SHIFT a tap 'at 1439' starts as 1439 * 2^-20 and addresses floor(value * 2^20) samples"""
    o = Oracle(1)
    o.set_option(OPT_DANE)
    o.set_option(OPT_SHIFT)
    assert o.load_text(CHORUS.replace("macs &rd1, 0.0013, lfo, depth\n", "")), o.errors()
    assert o.get_register_bits("&rd1") == bits(np.float32(1439.0 * 2.0 ** -20))[()]
    x = np.zeros(1600, np.float32)
    x[0] = 1.0
    y = o.process_block(x)
    assert y[700] == 0.25 and y[1439] == 0.5 and np.count_nonzero(y) == 2  # out = 0.5 * (rd1 + 0.5 * rd2)
    # a position written by the program: 100.7 samples -> tap at 100
    o.set_register("&rd2", float(np.float32(100.7 * 2.0 ** -20)))
    y = o.process_block(x)
    assert y[100] == 0.25


def test_interpolated_reads_weigh_the_two_neighbours_with_the_address_fraction():
    """FX_OPT_TRAM_INTERP: a READ tap at DANE address a returns x0 + f * (x1 - x0), x0 at position a >> 11, x1 one position
    further (one sample older), f = (a & 0x7ff) / 2048; f == 0 is x0 itself; writes ignore the fraction"""
    text = ("itramsize 64 \ninput in 0\noutput out 0\nstatic rd\nidelay write, in, at, 0\nidelay read, rd, at, 10\nmacs out, 0, rd, 1.0\nend")
    x = (np.arange(200, dtype=np.float32) * np.float32(2.0 ** -10)).astype(np.float32)   # a ramp: interpolation is exact on it
    for frac in (0, 1024, 512, 1, 2047):
        o = Oracle(1)
        for opt in (OPT_DANE, OPT_SHIFT, OPT_INTERP):
            o.set_option(opt)
        assert o.load_text(text), o.errors()
        o.set_register("&rd", float(np.float32((10 * 2048 + frac) * 2.0 ** -31)))
        y = o.process_block(x)
        n = np.arange(20, 200)
        want = (x[n - 10].astype(np.float64) + (frac / 2048.0) * (x[n - 11].astype(np.float64) - x[n - 10].astype(np.float64))).astype(np.float32)
        assert np.array_equal(bits(y[20:]), bits(want)), frac
    # without the option the fraction is dropped (FX_OPT_TRAM_ADDR_SHIFT alone)
    o = Oracle(1)
    o.set_option(OPT_DANE)
    o.set_option(OPT_SHIFT)
    assert o.load_text(text)
    o.set_register("&rd", float(np.float32((10 * 2048 + 1024) * 2.0 ** -31)))
    y = o.process_block(x)
    assert np.array_equal(bits(y[20:]), bits(x[10:190]))


FRACTIONAL_TAPS = """itramsize 97 
xtramsize 300 
input in 0
output out 0
static w
static r1
static r2
static xr
idelay write, w, at, 0
idelay read, r1, at, 7
idelay read, r2, at, 19
xdelay write, in, at, 3
xdelay read, xr, at, 203
macs w, in, 0, 0
macs r1, r1, r2, 0.5
macs out, r1, xr, 0.25
end"""


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["default", "hip", "asm", "asm_lds", "asm_v256"])
def test_gpu_interpolated_and_modulated_taps(gpu, kernel, monkeypatch):
    """FX_OPT_TRAM_INTERP on the device: taps with a fixed fractional position (two scalar slots per tap) and the chorus, whose
    tap position is computed per instance and per sample - generated code gathers it per lane (fp32 modulo, exact below 2^23),
    the interpreter's TRAM handlers take the per-lane path for every tap (fp64 modulo) - against the oracle, on the translated
    tier, on the interpreter (register file in VGPRs and in LDS) and on the HIP C++ kernel"""
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    monkeypatch.delenv("FX_KERNEL", raising=False)
    if kernel != "default":
        monkeypatch.setenv("FX_KERNEL", kernel)
    n, s = 150, 3100
    x = progs.stimulus(n, s)
    for text, regs in ((FRACTIONAL_TAPS, ("&r1", "&r2", "&xr")), (CHORUS, ())):
        b = gpu.Batch(n, 1, 0)
        for opt in (gpu.OPT_TRAM_DANE, gpu.OPT_TRAM_ADDR_SHIFT, gpu.OPT_TRAM_INTERP):
            b.set_option(opt)
        assert b.load_text(text), b.errors()
        pos = {"&r1": (7 * 2048 + 1024), "&r2": (19 * 2048 + 1), "&xr": (203 * 2048 + 2047)}
        for reg in regs:
            b.set_register(reg, float(np.float32(pos[reg] * 2.0 ** -31)))
        depth = np.linspace(0.0, 0.0012, n).astype(np.float32)
        speed = np.linspace(0.05, 0.6, n).astype(np.float32)
        if text is CHORUS:
            b.set_register_array("depth", depth)
            b.set_register_array("speed", speed)
        y = np.concatenate([b.process_block(x[:1000]), b.process_block(x[1000:1001]), b.process_block(x[1001:])], axis=0)
        k = b.info("kernel")
        assert {"default": k >= 9, "hip": k == 0, "asm": 2 <= k <= 8, "asm_lds": k == 1, "asm_v256": k == 8}[kernel], (kernel, k)
        assert b.ood_flags() == 0
        for inst in (0, 63, 64, 99, n - 1):
            o = Oracle(1)
            for opt in (OPT_DANE, OPT_SHIFT, OPT_INTERP):
                o.set_option(opt)
            assert o.load_text(text)
            for reg in regs:
                o.set_register(reg, float(np.float32(pos[reg] * 2.0 ** -31)))
            if text is CHORUS:
                o.set_register("depth", float(depth[inst]))
                o.set_register("speed", float(speed[inst]))
            ref = o.process_block(x[:, inst].copy())
            assert np.array_equal(bits(ref), bits(y[:, inst])), (kernel, inst)
            assert b.instruction_counter_i(inst) == o.instruction_counter()


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [False, True], ids=["sample_positions", "dane_addresses"])
@pytest.mark.parametrize("text", [TWO_TAPS, CHORUS], ids=["two_taps", "modulated_chorus"])
def test_gpu_matches_the_oracle_in_the_dane_model(gpu, text, shift, monkeypatch):
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    if shift and text is TWO_TAPS:
        pytest.skip("literal positions only: the shifted form is covered by the chorus")
    n, s = 150, 3100
    x = progs.stimulus(n, s)
    b = gpu.Batch(n, 1, 0)
    b.set_option(gpu.OPT_TRAM_DANE)
    if shift:
        b.set_option(gpu.OPT_TRAM_ADDR_SHIFT)
    assert b.load_text(text), b.errors()
    if text is CHORUS:  # per-instance modulation depth and speed: every lane addresses its own slots
        b.set_register_array("depth", np.linspace(0.0, 0.0012 if shift else 0.9, n).astype(np.float32))
        b.set_register_array("speed", np.linspace(0.05, 0.6, n).astype(np.float32))
    ys = [b.process_block(x[:1000]), b.process_block(x[1000:])]
    y = np.concatenate(ys, axis=0)
    # static taps are generated inline by the translator, and so are per-instance positions given as DANE addresses (the modulated
    # chorus with the address shift: gathered per lane); with per-instance positions in whole samples the translated program calls
    # the interpreter's tap handlers (fp64 modulo per lane) and steps the lanes' own counters
    assert b.info("kernel") >= 9
    assert b.ood_flags() == 0
    for inst in (0, 63, 64, 99, n - 1):
        o = Oracle(1)
        o.set_option(OPT_DANE)
        if shift:
            o.set_option(OPT_SHIFT)
        assert o.load_text(text)
        if text is CHORUS:
            o.set_register("depth", float(np.linspace(0.0, 0.0012 if shift else 0.9, n).astype(np.float32)[inst]))
            o.set_register("speed", float(np.linspace(0.05, 0.6, n).astype(np.float32)[inst]))
        ref = o.process_block(x[:, inst].copy())
        assert np.array_equal(bits(ref), bits(y[:, inst])), inst
        assert b.instruction_counter_i(inst) == o.instruction_counter()
        for r in ("out", "ccr") + (("&rd1", "lfo") if text is CHORUS else ("&r1", "w")):
            assert b.get_register_bits_i(r, inst) == o.get_register_bits(r), (inst, r)


EXTREME_POSITIONS = np.array([0.0, 1.0, -1.0, 63.0, 64.0, 65.0, -64.0, -65.0, 8388609.0, -8388609.0, 2147483520.0, -2147483648.0, 3.0e9,
                              -3.0e9, 1.0e30, np.inf, -np.inf, np.nan, -5.5, 5.999, 0.99, -0.99, 1.0e-40, 499.0, 500.0, 12345678.0], dtype=np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [False, True], ids=["sample_positions", "dane_addresses"])
@pytest.mark.parametrize("kernel", ["default", "hip", "asm", "asm_lds"])
def test_gpu_tap_positions_at_the_edges_of_the_modulo(gpu, kernel, shift, monkeypatch):
    """per-instance tap positions, read and write, on both lines: multiples of the line length, negative, beyond 2^23 (where an
    fp32 modulo would no longer be exact), at the ends of int32 and beyond (cvttss2si gives 0x80000000), non-finite - the slot is
    the C `(long)(counter + position) % size` of the oracle everywhere (interpreter: fp64 quotient by a refined reciprocal)"""
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    monkeypatch.delenv("FX_KERNEL", raising=False)
    if kernel != "default":
        monkeypatch.setenv("FX_KERNEL", kernel)
    n, s = 130, 700
    x = progs.stimulus(n, s)
    pos = np.resize(EXTREME_POSITIONS, n).astype(np.float32)
    if shift:   # DANE addresses: value * 2^31 >> 11; keep the interesting ones interesting
        pos = np.where(np.isfinite(pos) & (np.abs(pos) < 3.0e6), pos * np.float32(2.0 ** -20), pos).astype(np.float32)
    b = gpu.Batch(n, 1, 0)
    b.set_option(gpu.OPT_TRAM_DANE)
    if shift:
        b.set_option(gpu.OPT_TRAM_ADDR_SHIFT)
        b.set_option(gpu.OPT_TRAM_INTERP)
    assert b.load_text(TWO_TAPS), b.errors()
    for reg, roll in (("&r1", 0), ("&r2", 3), ("&xr", 7), ("&w", 11)):
        b.set_register_array(reg, np.roll(pos, roll))
    y = np.concatenate([b.process_block(x[:300]), b.process_block(x[300:])], axis=0)
    k = b.info("kernel")
    assert {"default": k >= 9, "hip": k == 0, "asm": 2 <= k <= 8, "asm_lds": k == 1}[kernel], (kernel, k)
    for inst in range(0, n, 1 if kernel in ("asm", "asm_lds") else 3):
        o = Oracle(1)
        o.set_option(OPT_DANE)
        if shift:
            o.set_option(OPT_SHIFT)
            o.set_option(OPT_INTERP)
        assert o.load_text(TWO_TAPS)
        for reg, roll in (("&r1", 0), ("&r2", 3), ("&xr", 7), ("&w", 11)):
            o.set_register(reg, float(np.roll(pos, roll)[inst]))
        ref = o.process_block(x[:, inst].copy())
        assert np.array_equal(bits(ref), bits(y[:, inst])), (kernel, inst)
        assert b.instruction_counter_i(inst) == o.instruction_counter()
        assert b.get_cursors_i(inst) == o.cursors(), inst


@pytest.mark.gpu
def test_options_off_is_the_reference(gpu):
    """the same two-tap text without the option: reference cursor model, domain flags - and identical to the oracle without it"""
    n, s = 70, 40
    x = progs.stimulus(n, s)
    b = gpu.Batch(n, 1, 0)
    assert b.load_text(TWO_TAPS)
    y = b.process_block(x)
    o = Oracle(1)
    assert o.load_text(TWO_TAPS)
    ref = o.process_block(x[:, 9].copy())
    assert np.array_equal(bits(ref), bits(y[:, 9])) and b.ood_flags() == o.ood_flags() != 0
    b2 = gpu.Batch(n, 1, 0)
    assert not b2.load_text(CHORUS)


REVERB_DANE = None


def dane_reverb():
    """config5's reverb shape in the DANE convention: ONE write tap per line, reads at different positions of the same line"""
    L = ["xtramsize 8192 ", "input in 0", "output out 0", "control damp = 0.3", "control decay = 0.45", "static m", "static w"]
    taps = (1187, 2909, 4523, 7919)
    L += ["static d%d" % j for j in range(4)] + ["static lp%d" % j for j in range(4)]
    L += ["xdelay read, d%d, at, %d" % (j, taps[j]) for j in range(4)]
    L += ["interp lp%d, lp%d, damp, d%d" % (j, j, j) for j in range(4)]
    L += ["acc3 m, lp0, lp1, lp2", "macs m, m, lp3, 1.0", "macs m, 0, m, 0.25", "macs w, in, m, decay", "xdelay write, w, at, 0", "macs out, 0, m, 0.5", "end"]
    return "\n".join(L)


@pytest.mark.gpu
def test_static_multi_tap_line_translated_with_reads_a_sample_ahead(gpu, monkeypatch):
    """four taps on one line + one write: generated inline, the taps' loads issued one sample ahead (the collision rule is a
    translate-time constant in this model), across block boundaries; vs the oracle"""
    monkeypatch.delenv("FX_KERNEL", raising=False)
    n, s = 200, 9000
    text = dane_reverb()
    x = progs.stimulus(n, s)
    b = gpu.Batch(n, 1, 0)
    b.set_option(gpu.OPT_TRAM_DANE)
    assert b.load_text(text), b.errors()
    y = np.concatenate([b.process_block(x[:1]), b.process_block(x[1:4000]), b.process_block(x[4000:])], axis=0)
    assert b.info("kernel") >= 9 and b.ood_flags() == 0
    for inst in (0, 64, n - 1):
        o = Oracle(1)
        o.set_option(OPT_DANE)
        assert o.load_text(text)
        ref = o.process_block(x[:, inst].copy())
        assert np.array_equal(bits(ref), bits(y[:, inst])), inst
    # a write one position below a tap is the slot that tap reads NEXT sample: the early read must stay in place
    clash = text.replace("xdelay write, w, at, 0", "xdelay write, w, at, 1186")
    b2 = gpu.Batch(n, 1, 0)
    b2.set_option(gpu.OPT_TRAM_DANE)
    assert b2.load_text(clash)
    y2 = b2.process_block(x[:3000])
    o = Oracle(1)
    o.set_option(OPT_DANE)
    assert o.load_text(clash)
    assert np.array_equal(bits(o.process_block(x[:3000, 5].copy())), bits(y2[:, 5]))


@pytest.mark.gpu
@pytest.mark.parametrize("mode, kernel", [("", None), ("dane", None), ("dane", "asm"), ("dane", "asm_lds"), ("dane_lanes", None), ("dane_lanes", "asm"),
                                          ("dane_lanes", "asm_lds"), ("dane_lanes", "hip"), ("dane_shift", None), ("dane_shift", "asm"), ("dane_shift", "hip")])
def test_delay_line_fuzz(gpu, mode, kernel, monkeypatch):
    """tools/fuzz_tram.py, 150 programs per model and tier: programs that start with a group of TRAM reads (issued a sample ahead by
    the translated tier) on tiny lines - an early read meeting a later write of the same slot is the norm there -, balanced and
    unbalanced read/write counts, offsets, several short blocks; in the DANE model also with per-instance tap positions (whole
    samples: negative, multiples of the line, beyond int32, non-finite; DANE addresses with interpolated reads)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_tram
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    if kernel:
        monkeypatch.setenv("FX_KERNEL", kernel)
    monkeypatch.setattr(sys, "argv", ["fuzz_tram.py", "5000", "150"] + ([mode] if mode else []))
    assert fuzz_tram.main() == 0


@pytest.mark.gpu
def test_gpu_state_image_under_the_dane_model(gpu, monkeypatch):
    """fxb_save_state / fxb_load_state with the opt-in delay-line model in force: the per-sample address counters, the tap registers
    (&name) and the delay memory travel with the image - the chorus (per-instance, modulated taps) is stopped in the middle of a
    run and carries on in a new handle and in a two-shard handle exactly like the batch that was never stopped"""
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    monkeypatch.delenv("FX_KERNEL", raising=False)
    n, s = 150, 1500
    x = progs.stimulus(n, s + 300)
    depth = np.linspace(0.0, 0.0012, n).astype(np.float32)
    speed = np.linspace(0.05, 0.6, n).astype(np.float32)

    def make(**kw):
        b = gpu.Batch(n, 1, **kw)
        for opt in (gpu.OPT_TRAM_DANE, gpu.OPT_TRAM_ADDR_SHIFT, gpu.OPT_TRAM_INTERP):
            b.set_option(opt)
        assert b.load_text(CHORUS), b.errors()
        return b

    a = make(device=0)
    a.set_register_array("depth", depth)
    a.set_register_array("speed", speed)
    a.process_block(x[:s])
    img = a.save_state()
    ya = a.process_block(x[s:])
    for kw in ({"device": 0}, {"devices": [0, 0]}):
        b = make(**kw)
        b.load_state(img)
        assert np.array_equal(bits(b.process_block(x[s:])), bits(ya)), kw
        assert b.instruction_counter() == a.instruction_counter() and b.ood_flags() == 0
    o = Oracle(1)
    for opt in (OPT_DANE, OPT_SHIFT, OPT_INTERP):
        o.set_option(opt)
    assert o.load_text(CHORUS)
    o.set_register("depth", float(depth[99]))
    o.set_register("speed", float(speed[99]))
    ref = o.process_block(x[:, 99].copy())
    assert np.array_equal(bits(ref[s:]), bits(ya[:, 99]))
