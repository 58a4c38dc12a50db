"""Shrink a failing sequence of tools/fuzz_api.py: delete instructions of its program one at a time as long as the sequence still
fails (the host calls stay the same: they come from the seed's random stream).  Prints the reduced program.
    [FX_FUZZ_PANEL=1 ...] python tools/fuzz_api_reduce.py <seed>"""
import contextlib
import io
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_api as F  # noqa: E402

DECL = ("input", "output", "control", "static", "itramsize", "xtramsize", "name", "engine", "comment", "end")


def fails(seed, drop):
    def edit(text):
        lines = text.split("\n")
        return "\n".join(l for i, l in enumerate(lines) if i not in drop)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        try:
            ok = F.run(seed, False, edit)
        except Exception as e:   # (a reduced program the front-end refuses, a register that vanished)
            return False, str(e)
    return (not ok), out.getvalue().strip()


def main():
    seed = int(sys.argv[1])
    captured = {}
    def grab(text):
        captured["text"] = text
        return text
    with contextlib.redirect_stdout(io.StringIO()):
        F.run(seed, False, grab)
    lines = captured["text"].split("\n")
    bad, why = fails(seed, set())
    print("seed", seed, "fails:", bad, why, flush=True)
    if not bad:
        return 1
    sign = " ".join(why.split()[:5])   # "MISMATCH seed N step K": the same failure, not any failure
    drop = set()
    changed = True
    while changed:
        changed = False
        for i, l in enumerate(lines):
            if i in drop or not l.strip() or l.split()[0] in DECL:
                continue
            still, what = fails(seed, drop | {i})
            if still and what.startswith(sign):
                drop.add(i)
                changed = True
    print("reduced program (%d of %d instructions left):" % (sum(1 for i, l in enumerate(lines) if i not in drop and l.strip() and l.split()[0] not in DECL),
                                                              sum(1 for l in lines if l.strip() and l.split()[0] not in DECL)))
    print("\n".join(l for i, l in enumerate(lines) if i not in drop))
    print(fails(seed, drop)[1])
    return 0


if __name__ == "__main__":
    sys.exit(main())
