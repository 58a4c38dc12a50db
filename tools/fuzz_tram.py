#!/usr/bin/env python3
"""Differential fuzz of the delay-line paths of the translated tier against the oracle: programs that START with a group of
TRAM reads (the ones issued a sample ahead), tiny lines (collisions between an early read and a later write are the norm),
balanced and unbalanced read/write counts, write offsets, reads in the middle, several short blocks (cold entries, the
last-sample stream, cursor distances that change from launch to launch), optionally the opt-in DANE model - with literal tap
positions (`dane`), with tap registers given per-instance values in whole samples, negative and far beyond the line included
(`dane_lanes`: the translated program calls the interpreter's tap handlers), or per-instance DANE addresses with interpolated
reads (`dane_shift`: gathered per lane by generated code).  FX_KERNEL selects the tier as everywhere.

    python tools/fuzz_tram.py [first_seed] [count] [dane | dane_lanes | dane_shift]
"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fx8010-emulator-core_amd/python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402

import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402
from pyoracle import Oracle  # noqa: E402


def random_tram_program(rng, dane=False):
    isize, xsize = int(rng.integers(1, 13)), int(rng.integers(1, 30))
    regs = ["r%d" % i for i in range(6)]
    L = ["input in 0", "output out 0", "control c = 0.4", "itramsize %d " % isize, "xtramsize %d " % xsize] + ["static %s" % r for r in regs] + ["static acc"]
    n_lead = int(rng.integers(1, 5))
    lead = []
    for k in range(n_lead):
        op = "idelay" if rng.random() < 0.5 else "xdelay"
        pos = int(rng.integers(0, 12)) if dane else 0
        lead.append("%s read, r%d, at, %d" % (op, k, pos))
    L += lead
    body = []
    for k in range(int(rng.integers(2, 12))):
        a, b, d = (str(rng.choice(regs + ["in", "acc"])) for _ in range(3))
        kind = rng.random()
        if kind < 0.5:
            body.append("macs %s, %s, %s, %.3f" % (rng.choice(regs + ["acc"]), a, b, rng.uniform(-0.9, 0.9)))
        elif kind < 0.65:
            body.append("interp %s, %s, c, %s" % (rng.choice(regs + ["acc"]), a, b))
        elif kind < 0.8:
            op = "idelay" if rng.random() < 0.5 else "xdelay"
            body.append("%s write, %s, at, %d" % (op, a, int(rng.integers(0, 12)) if dane else int(rng.integers(0, 2))))
        elif kind < 0.9:
            op = "idelay" if rng.random() < 0.5 else "xdelay"
            body.append("%s read, %s, at, %d" % (op, rng.choice(regs), int(rng.integers(0, 12)) if dane else 0))
        else:
            body.append("acc3 acc, %s, %s, %s" % (a, b, d))
    L += body
    L += ["macs out, acc, r0, 0.5", "end"]
    return "\n".join(L)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    mode = sys.argv[3] if len(sys.argv) > 3 else ""
    dane = mode in ("dane", "dane_lanes", "dane_shift")
    options = [A.OPT_TRAM_DANE] if dane else []
    if mode == "dane_shift":
        options += [A.OPT_TRAM_ADDR_SHIFT, A.OPT_TRAM_INTERP]
    N = 70
    failures, kernels, hoisted = [], {}, 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(770000 + seed)
        text = random_tram_program(rng, dane)
        blocks = [int(b) for b in rng.integers(1, 9, size=int(rng.integers(2, 6)))]
        x = P.stimulus(N, sum(blocks)) * np.float32(float(os.environ.get("FX_FUZZ_SCALE", "1")))   # > 1: delay lines hand back values outside their rows' class
        b = A.Batch(N, 1, 0)
        for opt in options:
            b.set_option(opt)
        assert b.load_text(text), (seed, b.errors())
        lanes = {}
        if mode in ("dane_lanes", "dane_shift"):
            # tap registers ("&" + the tap's data register) with a value per instance
            taps = sorted({"&" + ln.split(",")[1].strip() for ln in text.split("\n") if ln.startswith(("idelay", "xdelay"))})
            for name in taps:
                if rng.random() < 0.6:
                    if mode == "dane_lanes":
                        v = rng.integers(-40, 60, size=N).astype(np.float32)
                        wild = rng.random(N) < 0.1
                        v[wild] = rng.choice(np.array([2.0 ** 24 + 2, -2.0 ** 24 - 2, 2147483520.0, -2147483648.0, 4.0e9, np.inf, np.nan], np.float32), size=int(wild.sum()))
                    else:   # DANE addresses: (position * 2048 + fraction) * 2^-31
                        v = ((rng.integers(-40, 60, size=N) * 2048 + rng.integers(0, 2048, size=N) * (rng.random(N) < 0.7)) * 2.0 ** -31).astype(np.float32)
                    lanes[name] = v
                    b.set_register_array(name, v)
        ys, at = [], 0
        for nb in blocks:
            ys.append(b.process_block(x[at:at + nb]))
            at += nb
        y = np.concatenate(ys, axis=0)
        kernels[b.info("kernel")] = kernels.get(b.info("kernel"), 0) + 1
        for n in (0, 63, 69):
            o = Oracle(1)
            for opt in options:
                o.set_option(opt)
            assert o.load_text(text)
            for name, v in lanes.items():
                o.set_register(name, float(v[n]))
            ref = o.process_block(x[:, n].copy())
            ok = np.array_equal(ref.view(np.uint32), np.ascontiguousarray(y[:, n]).view(np.uint32)) and b.instruction_counter_i(n) == o.instruction_counter() \
                and b.ood_flags() == o.ood_flags() and all(b.get_register_bits_i(r, n) == o.get_register_bits(r) for r in ("r0", "r1", "r5", "acc")) and b.get_cursors_i(n) == o.cursors()
            if not ok:
                failures.append((seed, n))
                break
    print("tram fuzz%s: %d programs, kernels %s failures %s" % (" (DANE model%s)" % mode[4:].replace("_", ", ") if dane else "", count, kernels, failures[:10]))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
