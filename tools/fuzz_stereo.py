"""Random multi-channel programs (2-4 inputs / outputs, the A.IOIndex input quirk of FX8010.cpp:1058,1060) against
the oracle.    python tools/fuzz_stereo.py [first_seed] [count]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fx8010-emulator-core_amd/python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402

import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402
from pyoracle import Oracle  # noqa: E402


def program(rng, ch):
    regs = ["r%d" % i for i in range(int(rng.integers(2, 12)))]
    L = ["input i%d %d" % (c, c) for c in range(ch)] + ["output o%d %d" % (c, c) for c in range(ch)]
    L += ["control g = 0.4", "static noise", "itramsize 5 ", "static rd"] + ["static %s" % r for r in regs]
    ins, outs = ["i%d" % c for c in range(ch)], ["o%d" % c for c in range(ch)]
    ops = ["macs", "macsn", "acc3", "interp", "macw", "limit", "macmv", "tstneg"]
    L.append("idelay read, rd, at, 0")
    for _ in range(int(rng.integers(4, 40))):
        src = lambda: str(rng.choice(regs + ins + ins + outs + ["g", "0.5", "-0.25", "1.0", "rd", "noise", "ccr"]))
        dst = str(rng.choice(regs + outs + outs))
        if rng.uniform() < 0.1:
            L.append("skip ccr, ccr, %s, %d" % (rng.choice(["2", "6", "8"]), rng.integers(0, 3)))
        else:
            L.append("%s %s, %s, %s, %s" % (rng.choice(ops), dst, src(), src(), src()))
    L.append("idelay write, %s, at, 0" % str(rng.choice(regs + ins)))
    for c in range(ch):
        L.append("macs o%d, o%d, %s, 0.5" % (c, c, str(rng.choice(regs + ins))))
    return "\n".join(L + ["end"])


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    bad = []
    for seed in range(first, first + count):
        rng = np.random.default_rng(660000 + seed)
        ch = int(rng.integers(2, 5))
        text = program(rng, ch)
        N, S = int(rng.choice([1, 64, 70, 129])), 14
        b = A.Batch(N, ch, 0)
        if not b.load_text(text):
            print("unloadable", seed, b.errors())
            bad.append(seed)
            continue
        x = np.stack([P.stimulus(N, S, first_instance=1000 * c) for c in range(ch)], axis=1)  # [S, ch, N]
        y1 = b.process_block(x)
        y2 = b.process_block(x)
        for n in sorted(set([0, N - 1])):
            o = Oracle(ch)
            o.load_text(text)
            r1 = o.process_block(np.ascontiguousarray(x[:, :, n]))
            r2 = o.process_block(np.ascontiguousarray(x[:, :, n]))
            if o.ood_flags():
                continue
            g1, g2 = np.ascontiguousarray(y1[:, :, n]), np.ascontiguousarray(y2[:, :, n])
            ok = True
            for r, g in ((r1, g1), (r2, g2)):
                ok = ok and np.array_equal(r.view(np.uint32), g.view(np.uint32))
            ok = ok and b.instruction_counter_i(n) == o.instruction_counter()
            if not ok:
                print("MISMATCH seed", seed, "channels", ch, "instance", n, "kernel", b.info("kernel"))
                bad.append(seed)
                break
    print("stereo fuzz:", count, "programs, failures", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
