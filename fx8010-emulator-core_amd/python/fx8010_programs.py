"""Benchmark / parity programs for the five BASELINE.json configurations, in the reference's
DANE-like ``.da`` dialect, plus the synthetic PCM generator of
SURVEY.md §8(d).

The program shapes follow SURVEY.md §8(d) items 1-5.  Dialect quirks that matter here:
``itramsize N `` needs exactly one trailing blank (reference regex, source/FX8010.cpp:377) and
the file must end with ``end`` (no blank line after it, source/FX8010.cpp:829-838).

``python fx8010_programs.py <dir>`` writes config2.da … config5.da and the two config-1
variants into <dir> (the committed copies live in ../programs/).
"""
import os
import sys

import numpy as np

SEED = 0xF8010


def stimulus(n_instances, n_samples, first_instance=0, first_sample=0, seed=SEED):
    """Counter-based PCM, SURVEY.md §8(d): u = hash32(seed, n, s); in = int32(u) * 2^-31 * 0.9f.

    Returns float32 [n_samples, n_instances] (sample-major, instance fastest — the layout the
    batch API takes).  Values lie in (-0.9, 0.9): LOG/EXP stay inside their table domain.
    """
    n = (np.arange(n_instances, dtype=np.uint64) + np.uint64(first_instance))[None, :]
    s = (np.arange(n_samples, dtype=np.uint64) + np.uint64(first_sample))[:, None]
    x = (n * np.uint64(0x9E3779B1) + s * np.uint64(0x85EBCA77) + np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    # murmur3 fmix32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    i32 = x.astype(np.uint32).view(np.int32)
    f = i32.astype(np.float32) * np.float32(2.0 ** -31)
    return (f * np.float32(0.9)).astype(np.float32)


_HEADER = 'name "{name}"\nengine "fx8010_emulator_v0"\ncomment "{comment}"\n'


def config1_shipped():
    """Active part of the reference's source/testcode.da:9-24,59 (slider test: one MACS + END)."""
    return (
        _HEADER.format(name="cfg1_slider", comment="config 1: shipped testcode shape")
        + "static a\nitramsize 1000 \nxtramsize 48000 \ninput in_l 0\ncontrol volume = 1.0\ncontrol pan = 0.5\n"
        "control filter_cutoff = 0.1\noutput out_l 0\nstatic rd\nstatic wr\nstatic noise\n"
        "macs out_l, 0, in_l, volume\nend"
    )


def config1_logtube():
    """README.md:33-35 LOG vacuum-tube variant (3 instructions incl. END)."""
    return (
        _HEADER.format(name="cfg1_logtube", comment="config 1: readme log tube")
        + "static a\ninput in_l 0\ncontrol volume = 0.5\noutput out_l 0\n"
        "log a, in_l, 3, 0\nmacs out_l, 0, a, 1.0\nend"
    )


def config2():
    """64 instructions: 31 x (INTERP one-pole low-pass ; MACS mix) + output MACS + END.  No TRAM."""
    L = [_HEADER.format(name="cfg2_lowpass", comment="config 2: 64-instr macs/interp one-pole chain")]
    L += ["input in 0", "output out 0", "control cutoff = 0.1", "static t"]
    L += ["static s%d" % i for i in range(31)]
    prev = "in"
    for i in range(31):
        L.append("interp s%d, s%d, cutoff, %s" % (i, i, prev))
        L.append("macs t, s%d, in, 0.05" % i)
        prev = "t"
    L.append("macs out, 0, t, 1.0")
    L.append("end")
    return "\n".join(L)


def config3():
    """256 instructions: 1000-sample iTRAM feedback delay (read@0 then write@0) + 251 filler."""
    L = [_HEADER.format(name="cfg3_delay", comment="config 3: 256-instr, 1000-sample itram feedback")]
    L += ["itramsize 1000 ", "input in 0", "output out 0", "control cutoff = 0.1", "control fb = 0.5",
          "static rd", "static a", "static t"]
    ns = 32
    L += ["static s%d" % i for i in range(ns)]
    L.append("idelay read, rd, at, 0")
    L.append("macs a, in, rd, fb")
    L.append("idelay write, a, at, 0")
    prev = "a"
    n = 0
    k = 0
    while n < 251:
        L.append("interp s%d, s%d, cutoff, %s" % (k % ns, k % ns, prev))
        n += 1
        if n < 251:
            L.append("macs t, s%d, a, 0.03" % (k % ns))
            n += 1
        prev = "t"
        k += 1
    L.append("macs out, 0, t, 1.0")
    L.append("end")
    return "\n".join(L)


def config4():
    """512 instructions: LOG/EXP tube + INTERP cells with an input-sign-dependent SKIP per cell."""
    L = [_HEADER.format(name="cfg4_tube", comment="config 4: 512-instr log/exp tube, interp chain, skip")]
    L += ["input in 0", "output out 0", "control cutoff = 0.2", "static a", "static b", "static t", "static x", "static o"]
    ns = 16
    L += ["static s%d" % i for i in range(ns)]
    L.append("macs x, 0, in, 1.0")
    for c in range(63):
        s = "s%d" % (c % ns)
        k = 1 + (c % 3)
        L += [
            "log a, x, 3, 0",
            "exp b, a, 7, 0",
            "interp %s, %s, cutoff, b" % (s, s),
            "macs t, x, 0, 0",
            "skip ccr, ccr, 6, %d" % k,
            "macs x, %s, b, 0.5" % s,
            "macsn x, x, a, 0.25",
            "macs x, x, in, 0.1",
        ]
    L += ["interp o, o, cutoff, x", "macs t, o, x, 0.25", "macsn t, t, a, 0.125", "acc3 o, o, t, 0", "interp o, o, cutoff, t"]
    L.append("macs out, 0, o, 1.0")
    L.append("end")
    return "\n".join(L)


def config5(pairs=4):
    """512 instructions: reverb-style network over an 8192-sample xTRAM with `pairs` read/write pairs."""
    L = [_HEADER.format(name="cfg5_reverb", comment="config 5: 512-instr reverb, 8192-sample xtram")]
    L += ["xtramsize 8192 ", "input in 0", "output out 0", "control damp = 0.3", "control decay = 0.45", "control diff = 0.6",
          "static u", "static v", "static m"]
    L += ["static d%d" % j for j in range(pairs)]
    L += ["static w%d" % j for j in range(pairs)]
    L += ["static lp%d" % j for j in range(pairs)]
    ny = 40
    L += ["static y%d" % i for i in range(ny)]
    body = []
    for j in range(pairs):
        body.append("xdelay read, d%d, at, 0" % j)
    for j in range(pairs):
        body.append("interp lp%d, lp%d, damp, d%d" % (j, j, j))
    # Householder-style mix: m = sum(lp)/2 ; w_j = in*0.25 + decay*(lp_j - m)
    body.append("acc3 m, lp0, lp1, lp2")
    body.append("macs m, m, lp3, 1.0" if pairs > 3 else "macs m, m, 0, 0")
    body.append("macs m, 0, m, 0.5")
    for j in range(pairs):
        body.append("macsn w%d, lp%d, m, 1.0" % (j, j))
        body.append("macs w%d, 0, w%d, decay" % (j, j))
        body.append("macs w%d, w%d, in, 0.25" % (j, j))
    total_fixed = len(body) + pairs + 2  # + writes + out + end
    n_fill = 512 - total_fixed
    # diffusion: chains of first-order all-pass sections v = u - g*y ; u' = y + g*v ; y = v (3 instrs),
    # interleaved with damping INTERPs and an ACC3 tap sum.
    fill = []
    k = 0
    src = ["w%d" % j for j in range(pairs)]
    while len(fill) + 5 <= n_fill:
        y = "y%d" % (k % ny)
        t = src[k % pairs]
        fill.append("macsn v, %s, %s, diff" % (t, y))
        fill.append("macs u, %s, v, diff" % y)
        fill.append("interp %s, %s, damp, v" % (y, y))
        fill.append("macs %s, 0, u, 0.7" % t)
        fill.append("acc3 m, m, u, 0" if k % 4 == 3 else "macs m, m, u, 0.05")
        k += 1
    while len(fill) < n_fill:
        fill.append("macs m, m, u, 0.01")
    body += fill
    for j in range(pairs):
        body.append("xdelay write, w%d, at, 0" % j)
    body.append("macs out, 0, m, 0.5")
    body.append("end")
    L += body
    return "\n".join(L)


def tram_bound(pairs=16):
    """Not a BASELINE configuration: a delay-line-dominated program (16 xTRAM reads + 16 writes around 34 cheap
    instructions, 136 algorithmic bytes per instance-sample) that puts the HBM side of the roofline to the test."""
    L = [_HEADER.format(name="tram_bound", comment="memory-bound probe: 16 xtram taps")]
    L += ["xtramsize 8192 ", "input in 0", "output out 0", "static m"]
    L += ["static d%d" % j for j in range(pairs)] + ["static w%d" % j for j in range(pairs)]
    body = ["xdelay read, d%d, at, 0" % j for j in range(pairs)]
    body.append("macs m, 0, in, 0.5")
    for j in range(pairs):
        body.append("macs w%d, m, d%d, 0.45" % (j, (j + 1) % pairs))
    body.append("macs out, 0, w0, 0.5")
    body += ["xdelay write, w%d, at, 0" % j for j in range(pairs)]
    return "\n".join(L + body + ["end"])


def config5_dane(pairs=4):
    """config5's 512-instruction reverb in the kX / DANE convention (opt-in FX_OPT_TRAM_DANE, NOT reference behaviour): the
    four read taps sit at different positions of the line, the four writes at positions 0..3 - multi-tap delay lines."""
    text = config5(pairs)
    taps = (1187, 2909, 4523, 7919, 1601, 3301, 5003, 6997)
    for j in range(pairs):
        text = text.replace("xdelay read, d%d, at, 0" % j, "xdelay read, d%d, at, %d" % (j, taps[j]))
        text = text.replace("xdelay write, w%d, at, 0" % j, "xdelay write, w%d, at, %d" % (j, j))
    return text


def wide_chains(width=12):
    """Not a BASELINE configuration: `width` parallel one-pole chains advancing side by side - width + 1 rows cross every cut
    of a pipelined program (wide packets: short rings, tests/test_gpu_stages.py, tools/stage_probe.py)."""
    W = width
    return ("input in 0\noutput out 0\ncontrol k = 0.25\n" + "".join("static a%d\n" % i for i in range(W))
            + "".join("static s%d_%d\n" % (i, j) for i in range(W) for j in range(6))
            + "".join("macs a%d, in, in, 0.%02d\n" % (i, i + 1) for i in range(W))
            + "".join("interp s%d_%d, s%d_%d, k, a%d\nmacs a%d, s%d_%d, in, 0.05\n" % (i, j, i, j, i, i, i, j) for j in range(6) for i in range(W))
            + "macs out, 0, a0, 1.0\n" + "".join("macs out, out, a%d, 0.1\n" % i for i in range(1, W)) + "end")


def mixed_stages(sections=3):
    """Not a BASELINE configuration: a delay line with feedback and noise (stage 0), filter sections, a SKIP with its shadow and a
    LOG / EXP pair per section - everything a stage can hold (tests/test_gpu_stages.py, tools/stage_probe.py)."""
    L = ["itramsize 11 ", "input in 0", "output out 0", "control k = 0.25", "static noise", "static rd", "static a", "static t", "static u", "static w"]
    L += ["static s%d" % i for i in range(10 * sections)]
    L += ["idelay read, rd, at, 0", "macs a, in, rd, 0.5", "macs a, a, noise, 0.125", "idelay write, a, at, 0"]
    L += ["interp s0, s0, k, a", "macs t, s0, a, 0.05"]
    for q in range(sections):
        b = 10 * q
        L += ["interp s%d, s%d, k, t\nmacs t, s%d, a, 0.05" % (b + i, b + i, b + i) for i in range(1, 4)]
        L += ["macs u, t, 0, 0", "skip ccr, ccr, 6, 2", "macs t, t, 0.5, 0.5", "macs out, out, t, 0.1", "log w, u, 3, 0", "exp u, w, 5, 0"]
        L += ["interp s%d, s%d, k, t\nmacs t, s%d, u, 0.05" % (b + i, b + i, b + i) for i in range(4, 10)]
    L += ["macs out, out, t, 0.5", "end"]
    return "\n".join(L)


# programs of tools/stage_probe.py and of the stage tests that are no BASELINE configuration
PROBE_PROGRAMS = {"wide12": lambda: wide_chains(12), "wide4": lambda: wide_chains(4), "mixed_stages": mixed_stages}

# programs that need an option of the library (FX_OPT_*) before loading
CONFIG_OPTIONS = {"config5_dane": 1}

CONFIGS = {
    "config1_shipped": config1_shipped,
    "config1_logtube": config1_logtube,
    "config2": config2,
    "config3": config3,
    "config4": config4,
    "config5": config5,
    "tram_bound": tram_bound,
    "config5_dane": config5_dane,
}

# instances BASELINE.json quotes per config: CONFIG_TOTAL_INSTANCES = the whole job (config 5: 2 097 152, "sharded across 8" - a FIXED
# total: strong scaling), CONFIG_INSTANCES = one GPU's share of it when all the GPUs the config names are there (config 5: 1/8)
CONFIG_INSTANCES = {"config1_shipped": 1, "config1_logtube": 1, "config2": 4096, "config3": 65536, "config4": 262144, "config5": 262144, "tram_bound": 262144, "config5_dane": 262144}
CONFIG_TOTAL_INSTANCES = dict(CONFIG_INSTANCES, config5=2097152, tram_bound=2097152, config5_dane=2097152)


def count_instructions(text):
    ops = ("macs", "macsn", "macints", "macintw", "acc3", "macmv", "macw", "macwn", "skip", "andxor", "tstneg", "limit",
           "limitn", "log", "exp", "interp", "idelay", "xdelay", "end")
    n = 0
    for line in text.split("\n"):
        w = line.split(";")[0].strip().split(" ")[0].lower()
        if w in ops:
            n += 1
    return n


def write_all(directory):
    os.makedirs(directory, exist_ok=True)
    for name, fn in CONFIGS.items():
        with open(os.path.join(directory, name + ".da"), "wb") as fh:
            fh.write(fn().encode())


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "programs")
    write_all(out)
    for name, fn in CONFIGS.items():
        print(name, count_instructions(fn()))
