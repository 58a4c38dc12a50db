"""Random programs x random host call sequences against the oracle (default tier; FX_KERNEL pins another).

    python tools/fuzz_api.py [first_seed] [count]
    python tools/fuzz_api.py <seed> 1 verbose      one sequence, every step printed; before each step all registers, delay memory,
                                                   positions and LFSR words of the checked instances against the oracle

Modes (environment): FX_FUZZ_WILD=1 register values beyond [-1, 1]; FX_FUZZ_SHARDS=n a multi-shard handle; FX_FUZZ_PINNED=1 PCM in
pinned buffers (in place, aliased, overlapping); FX_FUZZ_PANEL=1 four declared controls in operand positions and the builder
thread waited for (control variants come and go); FX_FUZZ_NOCOMPARE=1 the calls only (the stand-in build under ASan).
For a failing seed: tools/fuzz_api_reduce.py shrinks the program; FX_FUZZ_FORCE_S=<step>:<n> gives the plain block of that step n
samples, FX_FUZZ_SPLIT=<step> runs it sample by sample with the whole state compared after each, FX_FUZZ_DUMP=<step> prints the
oracle's registers before it (this is how seed 2605911 was taken apart: profiles/r05_fuzz_campaign_x3.txt).
"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fx8010-emulator-core_amd/python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402
import stress_fuzz  # noqa: E402
from pyoracle import Oracle  # noqa: E402


# FX_FUZZ_NOCOMPARE=1: the call sequences only, nothing compared - for the stand-in build of the library (tests/hipstub: no real
# kernel runs there), where the point is the host engine under AddressSanitizer (tests/test_host_sanitizers.py)
NOCOMPARE = os.environ.get("FX_FUZZ_NOCOMPARE") == "1"


def same(ref, got):
    if NOCOMPARE:
        return True
    ref = np.asarray(ref, dtype=np.float32).reshape(-1)
    got = np.asarray(got, dtype=np.float32).reshape(-1)
    return np.array_equal(ref.view(np.uint32), got.view(np.uint32))  # NaN words included, on every tier


WILD = os.environ.get("FX_FUZZ_WILD") == "1"  # register values beyond [-1, 1] (state that breaks the bounded-row class)
PINNED = os.environ.get("FX_FUZZ_PINNED") == "1"  # PCM in pinned host buffers (fxb_host_alloc): blocks are processed in place (in == out among them); overlapping input and output ranges take the staged copies


PANEL = os.environ.get("FX_FUZZ_PANEL") == "1"  # three more declared controls in operand positions, and the builder thread waited for before half of the blocks: control variants come and go (all controls in rows / only the ones that have moved / folded in)


def with_panel(rng, text):
    """declare c2, c3, c4 and put them where registers or literals stood as A / X / Y operands of arithmetic instructions"""
    lines = text.split("\n")
    out = []
    for line in lines:
        parts = line.split(None, 1)
        if len(parts) == 2 and parts[0] in ("macs", "macsn", "interp", "acc3", "macw", "macwn", "macints", "macintw", "limit", "limitn", "tstneg", "andxor", "macmv") and rng.uniform() < 0.35:
            ops = [o.strip() for o in parts[1].split(",")]
            if len(ops) == 4:
                ops[int(rng.integers(1, 4))] = str(rng.choice(["c2", "c3", "c4"]))
                line = parts[0] + " " + ", ".join(ops)
        out.append(line)
    at = next(i for i, l in enumerate(out) if l.startswith("control c"))
    out[at + 1:at + 1] = ["control c2 = 0.5", "control c3 = 0.25", "control c4 = 0.125"]
    return "\n".join(out)


def value(rng):
    if WILD and rng.uniform() < 0.3:
        return float(np.float32(rng.choice([2.5, -3.0, 1.0000001, 100.0, 1e30, -1e-40])))
    return float(np.float32(rng.uniform(-1.0, 1.0)))


STATS = {"loaded": 0}   # sequences whose program loaded (a generator that produces text the front-end refuses tests nothing)


def run(seed, verbose=False, edit=None):
    """edit: a function text -> text applied to the generated program (tools/fuzz_api_reduce.py deletes instructions with it)"""
    rng = np.random.default_rng(880000 + seed)
    gen = stress_fuzz.random_program2 if seed % 2 else stress_fuzz.random_program
    n_regs = int(rng.integers(3, 30))
    text = gen(rng, int(rng.integers(6, 70)), n_regs)
    if PANEL:
        text = with_panel(rng, text)
    if edit:
        text = edit(text)
    N = int(rng.choice([1, 63, 64, 65, 130, 200]))
    check = sorted(set([0, N - 1, N // 2]))
    shards = int(os.environ.get("FX_FUZZ_SHARDS", "1"))   # > 1: the same call sequence through a multi-shard handle (all shards on device 0)
    b = A.Batch(N, 1, devices=[0] * shards) if shards > 1 and (N + 63) // 64 >= shards else A.Batch(N, 1, 0)
    if not b.load_text(text):
        return True
    STATS["loaded"] += 1
    oracles = {}
    for n in check:
        o = Oracle(1)
        o.load_text(text)
        oracles[n] = o
    x = P.stimulus(N, 1200)
    pos = 0
    pin_in = A.HostBuffer((44, N)) if PINNED else None
    pin_out = A.HostBuffer((40, N)) if PINNED else None

    def process(handle, xs):
        if PANEL and rng.integers(0, 2) and os.environ.get("FX_FUZZ_PANEL_PREPARE") != "0":
            handle.prepare(xs.shape[0], True)      # (what the builder thread was asked for is there: the block below may adopt it)
        if not PINNED:
            return handle.process_block(xs)
        S = xs.shape[0]
        how = rng.integers(0, 4)
        if how == 2:      # the output a few sample periods behind the input in one buffer: overlapping ranges take the staged copies
            pin_in.array[:S] = xs
            return handle.process_block(pin_in.array[:S], pin_in.array[3:3 + S]).copy()
        if how == 3:      # ... in front of it
            pin_in.array[4:4 + S] = xs
            return handle.process_block(pin_in.array[4:4 + S], pin_in.array[:S]).copy()
        pin_in.array[:S] = xs
        if how == 1:
            return handle.process_block(pin_in.array[:S], pin_in.array[:S]).copy()
        return handle.process_block(pin_in.array[:S], pin_out.array[:S]).copy()

    names = ["c", "r0", "r1", "r%d" % (n_regs - 1), "out"] + (["c2", "c3", "c4", "c2", "c"] if PANEL else [])
    if verbose:
        print(text)
        print("N", N, "check", check, "names", names)
    for step in range(30):
        op = rng.integers(0, 15)
        if os.environ.get("FX_FUZZ_DUMP") == str(step):   # debugging: the oracle's registers before this step
            every = [l.split()[1] for l in text.split("\n") if l.startswith(("static ", "control "))] + ["in", "out", "ccr"]
            print("   oracle before step %d, instance %d:" % (step, check[0]), {r: "%08x" % oracles[check[0]].get_register_bits(r) for r in dict.fromkeys(every)},
                  "cursors", oracles[check[0]].cursors(), "tram0", ["%08x" % v for v in np.asarray(oracles[check[0]].tram(0, 8)).view(np.uint32)])
        if verbose:
            every = [l.split()[1] for l in text.split("\n") if l.startswith(("static ", "control "))] + ["out", "ccr"]
            img = b.save_state()
            hdr = np.frombuffer(img[:64].tobytes(), dtype=np.int32)
            image_rows = img[64:64 + int(hdr[6]) * N * 4].view(np.uint32).reshape(int(hdr[6]), N)
            for n in check:
                bad = [(r, "%08x" % b.get_register_bits_i(r, n), "%08x" % oracles[n].get_register_bits(r)) for r in dict.fromkeys(names + every)
                       if b.get_register_bits_i(r, n) != oracles[n].get_register_bits(r)]
                if b.get_cursors_i(n) != oracles[n].cursors():
                    bad.append(("cursors", b.get_cursors_i(n), oracles[n].cursors()))
                nregs = int(hdr[5])
                lfsr = [int(v) for v in image_rows[nregs + 1 + 4: nregs + 1 + 6, n].view(np.int32)]
                if lfsr != oracles[n].lfsr():
                    bad.append(("lfsr", lfsr, oracles[n].lfsr()))
                for which in (0, 1):
                    try:
                        dev, ora = b.get_tram_i(which, n, 64), oracles[n].tram(which, 64)
                        if not np.array_equal(np.asarray(dev).view(np.uint32), np.asarray(ora).view(np.uint32)):
                            bad.append(("tram%d" % which, ["%08x" % v for v in np.asarray(dev).view(np.uint32)[:12]], ["%08x" % v for v in np.asarray(ora).view(np.uint32)[:12]]))
                    except Exception as e:
                        bad.append(("tram%d" % which, str(e)[:80]))
                if bad or b.instruction_counter_i(n) != oracles[n].instruction_counter():
                    print("  BEFORE step", step, "instance", n, "differs:", bad, "ctr", b.instruction_counter_i(n), oracles[n].instruction_counter())
            print("step", step, "op", int(op), "kernel", b.info("kernel"))
        if op == 13:
            # the whole state through an image into a NEW handle (fxb_save_state / fxb_load_state), which carries on: registers, delay
            # memory, positions, LFSR, counters, per-instance rows - and whatever the host side has to re-learn about them
            img = b.save_state()
            nb = A.Batch(N, 1, devices=[0] * shards) if shards > 1 and (N + 63) // 64 >= shards else A.Batch(N, 1, 0)
            if not nb.load_text(text):
                print("RELOAD seed %d step %d" % (seed, step))
                return False
            nb.load_state(img)
            b = nb
            if verbose:
                print("  state image -> new handle")
            continue
        if op == 14:
            b.prepare(int(rng.choice([1, 8, 40, 300])), bool(rng.integers(0, 2)))
            continue
        if op == 12:
            # control tracks: schedules of register values applied inside the next block (fxb_set_register_track), the oracle
            # gets the same values through set_register between its samples
            S = int(rng.choice([1, 5, 16, 33]))
            xs = x[pos:pos + S]
            pos += S
            plans = []
            pool = names[:3] if rng.integers(0, 2) else list(dict.fromkeys(names))   # (a few registers again and again, or any: up to sixteen get a slot)
            for name in rng.choice(pool, size=int(rng.integers(1, min(len(pool), 5) + 1)), replace=False):
                period = int(rng.choice([1, 2, 3, 8]))
                steps = int(rng.integers(1, S // period + 3))
                if rng.integers(0, 2):
                    vals = np.array([value(rng) for _ in range(steps)], dtype=np.float32)
                else:
                    vals = np.array([[value(rng) for _ in range(N)] for _ in range(steps)], dtype=np.float32)
                b.set_register_track(str(name), vals, period)
                plans.append((str(name), period, vals))
                if verbose:
                    print("  track", name, "period", period, "steps", steps, "per-instance" if vals.ndim == 2 else "broadcast", "S", S)
            y = process(b, xs)
            if verbose:
                for name, period, vals in plans:
                    print("   ", name, "schedule for instance", check[0], [float(v if vals.ndim == 1 else v[check[0]]) for v in vals],
                          "register now", b.get_register_i(name, check[0]))
            for n in check:
                ref = np.empty(S, dtype=np.float32)
                for t in range(S):
                    for name, period, vals in plans:
                        if t % period == 0 and t // period < vals.shape[0]:
                            v = vals[t // period]
                            oracles[n].set_register(name, float(v if vals.ndim == 1 else v[n]))
                    ref[t] = oracles[n].process_block(xs[t:t + 1, n].copy())[0]
                if oracles[n].ood_flags():
                    return True
                if not same(ref, y[:, n]):
                    print("MISMATCH (tracks) seed %d step %d instance %d kernel %d" % (seed, step, n, b.info("kernel")))
                    return False
        elif op < 6:
            S = int(rng.choice([1, 3, 8, 16, 40]))
            if os.environ.get("FX_FUZZ_FORCE_S", "").startswith("%d:" % step):   # debugging: "6:2" = step 6 with 2 samples
                S = int(os.environ["FX_FUZZ_FORCE_S"].split(":")[1])
            xs = x[pos:pos + S]
            pos += S
            if os.environ.get("FX_FUZZ_SPLIT") == str(step):   # debugging: the block as S blocks of one sample, the whole state compared after each
                every = [l.split()[1] for l in text.split("\n") if l.startswith(("static ", "control "))] + ["out", "ccr"]
                n = check[0]
                for i in range(S):
                    yi = b.process_block(xs[i:i + 1].copy())
                    ri = oracles[n].process_block(xs[i:i + 1, n].copy())
                    img = b.save_state()
                    hdr = np.frombuffer(img[:64].tobytes(), dtype=np.int32)
                    rows_ = img[64:64 + int(hdr[6]) * N * 4].view(np.uint32).reshape(int(hdr[6]), N)
                    nregs = int(hdr[5])
                    bad = [(r, "%08x" % b.get_register_bits_i(r, n), "%08x" % oracles[n].get_register_bits(r)) for r in dict.fromkeys(every) if b.get_register_bits_i(r, n) != oracles[n].get_register_bits(r)]
                    print("   sample %d of step %d: out device %08x oracle %08x; cursors %s %s; lfsr %s %s; tram0 %s | %s; differing registers %s" % (
                        i, step, int(yi.view(np.uint32)[0, n]), int(ri.view(np.uint32)[0]), b.get_cursors_i(n), oracles[n].cursors(),
                        [int(v) for v in rows_[nregs + 1 + 4: nregs + 1 + 6, n].view(np.int32)], oracles[n].lfsr(),
                        " ".join("%08x" % v for v in np.asarray(b.get_tram_i(0, n, 8)).view(np.uint32)), " ".join("%08x" % v for v in np.asarray(oracles[n].tram(0, 8)).view(np.uint32)), bad))
                return False
            else:
                y = process(b, xs)
            for n in check:
                ref = oracles[n].process_block(xs[:, n].copy())
                if oracles[n].ood_flags():
                    return True  # left the parity domain: nothing to compare from here on
                if not same(ref, y[:, n]):
                    print("MISMATCH seed %d step %d instance %d kernel %d" % (seed, step, n, b.info("kernel")))
                    if verbose:
                        rb, yb = np.asarray(ref, dtype=np.float32).view(np.uint32), np.ascontiguousarray(y[:, n]).view(np.uint32)
                        first = int(np.nonzero(rb != yb)[0][0])
                        print("  block of %d samples, first differing sample %d: oracle %08x device %08x; tier: %s" % (S, first, rb[first], yb[first], b.tier_note()))
                        print("  oracle", " ".join("%08x" % v for v in rb[:8]), "\n  device", " ".join("%08x" % v for v in yb[:8]))
                        every = [l.split()[1] for l in text.split("\n") if l.startswith(("static ", "control "))] + ["out", "ccr"]
                        print("  registers after the block (device, oracle):", [(r, "%08x" % b.get_register_bits_i(r, n), "%08x" % oracles[n].get_register_bits(r))
                                                                               for r in dict.fromkeys(every) if b.get_register_bits_i(r, n) != oracles[n].get_register_bits(r)])
                    return False
        elif op < 8:
            name, v = str(rng.choice(names)), value(rng)
            if verbose:
                print("  set_register", name, v)
            b.set_register(name, v)
            for o in oracles.values():
                o.set_register(name, v)
        elif op < 9:
            name, n, v = str(rng.choice(names)), int(rng.choice(check)), value(rng)
            if verbose:
                print("  set_register_i", name, n, v)
            b.set_register_i(name, n, v)
            oracles[n].set_register(name, v)
        elif op < 10:
            name = str(rng.choice(names))
            vals = np.array([value(rng) for _ in range(N)], dtype=np.float32)
            if verbose:
                print("  set_register_array", name)
            b.set_register_array(name, vals)
            for n in check:
                oracles[n].set_register(name, float(vals[n]))
        elif op < 11:
            n = int(rng.choice(check))
            s1, s2 = int(rng.integers(-2**31, 2**31)), int(rng.integers(-2**31, 2**31))
            b.seed_noise_i(n, s1, s2)
            oracles[n].seed_noise(s1, s2)
        else:
            for n in check:
                for r in names + ["ccr"]:
                    gb, rb = b.get_register_bits_i(r, n), oracles[n].get_register_bits(r)
                    if gb != rb and not NOCOMPARE:
                        print("REGISTER seed %d step %d %s[%d] %08x %08x kernel %d" % (seed, step, r, n, gb, rb, b.info("kernel")))
                        return False
    for n in check:
        if verbose:
            print("counter", n, b.instruction_counter_i(n), oracles[n].instruction_counter())
        if b.instruction_counter_i(n) != oracles[n].instruction_counter() and not NOCOMPARE:
            print("COUNTER seed %d instance %d" % (seed, n))
            return False
    return True


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    if len(sys.argv) > 3 and sys.argv[3] == "verbose":
        return 0 if run(first, True) else 1
    bad = [s for s in range(first, first + count) if not run(s)]
    print("api fuzz:", count, "sequences (%d programs loaded), failures" % STATS["loaded"], bad)
    return 1 if bad or STATS["loaded"] < count // 2 else 0


if __name__ == "__main__":
    sys.exit(main())
