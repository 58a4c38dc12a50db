"""Differential fuzz of the stage pipeline (fx_xlate.hpp StageInfo) on the GPU box: random FEED-FORWARD programs - sections with
state registers of their own, values flowing forward through shared temporaries, SKIPs, LOG / EXP, delay lines and noise in the
first section, an output accumulated across sections - run with FX_STAGES = 2 .. 8 on ragged batches over several blocks and
compared bit for bit with the oracle (outputs, registers, instruction counters).
    python tools/fuzz_stages.py [first_seed] [count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fx8010-emulator-core_amd", "python"), os.path.join(ROOT, "oracle")]
try:
    import torch  # noqa: F401
except ImportError:
    pass
import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402
from pyoracle import Oracle  # noqa: E402

LITS = ["0", "0.5", "-0.25", "1.0", "0.125", "0.999", "-1", "0.03", "0.7"]


def random_program(rng, sections, per_section):
    L = ["input in 0", "output out 0", "control k = 0.3", "control g = 0.6", "static noise", "itramsize 13 ", "xtramsize 29 "]
    temps = ["x%d" % i for i in range(4)]
    L += ["static %s" % t for t in temps] + ["static rd", "static rx"]
    regs = []
    for s in range(sections):
        regs.append(["s%d_%d" % (s, i) for i in range(int(rng.integers(1, 4)))])
        L += ["static %s" % r for r in regs[-1]]
    body = []
    written = []   # temporaries defined so far in this sample
    use_tram = rng.uniform() < 0.5
    if use_tram:
        body += ["idelay read, rd, at, 0", "xdelay read, rx, at, 0"]
    for s in range(sections):
        mine = regs[s]
        for i in range(per_section):
            srcs = ["in"] + written + mine + LITS + ["k", "g"] + (["rd", "rx"] if use_tram else []) + (["noise"] if s == 0 else [])
            src = lambda: str(rng.choice(srcs))
            kind = rng.integers(0, 100)
            # destinations: own state, or a temporary (which later sections may read)
            dst = str(rng.choice(mine)) if rng.uniform() < 0.6 else str(rng.choice(temps))
            if kind < 45:
                body.append("%s %s, %s, %s, %s" % (rng.choice(["macs", "macsn", "acc3", "macints"]), dst, src(), src(), src()))
            elif kind < 65:
                body.append("interp %s, %s, %s, %s" % (dst, str(rng.choice(mine)), str(rng.choice(["k", "g", "0.25", "0.5"])), src()))
            elif kind < 75:
                a = str(rng.choice(mine))
                body.append("macs %s, %s, 0, 0" % (a, a))          # saturate: a bounded operand for the table
                body.append("%s %s, %s, %d, 0" % (rng.choice(["log", "exp"]), dst, a, int(rng.integers(1, 20))))
            elif kind < 83 and i + 3 < per_section:
                body.append("macs %s, %s, 0, 0" % (mine[0], src()))
                body.append("skip ccr, ccr, %s, %d" % (rng.choice(["6", "2", "8", "16"]), int(rng.integers(1, 3))))
            elif kind < 90:
                body.append("%s %s, %s, %s, %s" % (rng.choice(["limit", "limitn", "macw", "macintw", "tstneg"]), dst, src(), src(), src()))
            else:
                body.append("macs out, out, %s, 0.1" % src())
            if dst in temps and dst not in written:
                written.append(dst)
        if s == 0 and use_tram:
            body += ["idelay write, %s, at, 0" % mine[0], "xdelay write, in, at, 0"]
    body.append("macs out, out, %s, 0.25" % (written[-1] if written else "in"))
    return "\n".join(L + body + ["end"])


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    N, S = 70, 23
    failures, cut = [], {}
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        text = random_program(rng, int(rng.integers(2, 9)), int(rng.integers(2, 9)))
        K = int(rng.choice([2, 3, 4, 8]))
        os.environ["FX_STAGES"] = str(K)
        # one program in three with a packet ring shorter than the LDS would allow (4, 8 or 16 buffers: rings shorter than the pipeline)
        ring = int(rng.choice([0, 0, 0, 0, 1, 2, 4]))
        if ring:
            os.environ["FX_STAGES_GROUP"] = str(ring)
        else:
            os.environ.pop("FX_STAGES_GROUP", None)
        x = P.stimulus(N, S, first_instance=seed)
        if seed % 5 == 0:   # non-finite words: the taint hand-over between stages
            x = x.copy()
            x[int(rng.integers(0, S)), int(rng.integers(0, N))] = np.nan
            x[int(rng.integers(0, S)), int(rng.integers(0, N))] = np.inf
        b = A.Batch(N, 1, 0)
        if not b.load_text(text):
            print("LOAD", seed, b.errors())
            failures.append(seed)
            continue
        y = [b.process_block(x), b.process_block(x[:7]), b.process_block(x)]
        cut[b.info("waves_per_wg")] = cut.get(b.info("waves_per_wg"), 0) + 1
        names = [l.split()[1] for l in text.split("\n") if l.startswith("static ") and l.split()[1] != "noise"]
        for n in (0, 1, 63, 64, 69):
            o = Oracle(1)
            assert o.load_text(text)
            r = [o.process_block(x[:, n].copy()), o.process_block(x[:7, n].copy()), o.process_block(x[:, n].copy())]
            if o.ood_flags():
                continue
            ok = all(np.array_equal(a.view(np.uint32), np.ascontiguousarray(g[:, n]).view(np.uint32)) for a, g in zip(r, y))
            ok = ok and b.instruction_counter_i(n) == o.instruction_counter()
            ok = ok and all(b.get_register_bits_i(reg, n) == o.get_register_bits(reg) for reg in names + ["ccr", "out", "in"])
            if not ok:
                what = []
                for q, (a, g) in enumerate(zip(r, y)):
                    bad = np.nonzero(a.view(np.uint32) != np.ascontiguousarray(g[:, n]).view(np.uint32))[0]
                    if bad.size:
                        what.append("block %d: %d samples differ, first %d (ref %08x got %08x)" % (q, bad.size, bad[0], a.view(np.uint32)[bad[0]], np.ascontiguousarray(g[:, n]).view(np.uint32)[bad[0]]))
                if b.instruction_counter_i(n) != o.instruction_counter():
                    what.append("counter %d vs %d" % (b.instruction_counter_i(n), o.instruction_counter()))
                what += ["%s %08x vs %08x" % (reg, b.get_register_bits_i(reg, n), o.get_register_bits(reg)) for reg in names + ["ccr", "out", "in"]
                         if b.get_register_bits_i(reg, n) != o.get_register_bits(reg)]
                print("MISMATCH seed %d instance %d stages %d (kernel %d, waves/wg %d): %s\n%s" % (seed, n, K, b.info("kernel"), b.info("waves_per_wg"), "; ".join(what), text if os.environ.get("FX_FUZZ_TEXT") else ""), flush=True)
                failures.append(seed)
                break
    os.environ.pop("FX_STAGES_GROUP", None)
    print("stage fuzz: %d programs, waves per workgroup %s, failures %s" % (count, dict(sorted(cut.items())), failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
