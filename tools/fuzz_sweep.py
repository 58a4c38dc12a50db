"""Wide differential fuzz of the default tier against the oracle (the GPU test suite runs 96 of these programs).

    python tools/fuzz_sweep.py [first_seed] [count]

FX_FUZZ_LANES=1: registers the program only reads - literals and controls, the table numbers of LOG / EXP among them - get a value
per instance before the first block (what fxb_set_register_array is for: N objects with N settings), table numbers now and then
outside 0..31.
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


def same(ref, got):
    ref = np.asarray(ref, dtype=np.float32).reshape(-1)
    got = np.asarray(got, dtype=np.float32).reshape(-1)
    return np.array_equal(ref.view(np.uint32), got.view(np.uint32))  # NaN words included, on every tier


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    N, S = 70, 10
    x = P.stimulus(N, S) * np.float32(float(os.environ.get("FX_FUZZ_SCALE", "1")))   # > 1: inputs beyond the LOG/EXP tables
    if os.environ.get("FX_FUZZ_NAN"):   # non-finite words sprinkled over the input: NaNs of both signs with payloads, a signalling one, +-Inf
        words = np.array([0x7FC00000, 0xFFC00000, 0x7FC12345, 0xFF800001, 0x7F800000, 0xFF800000, 0x7FA00000], dtype=np.uint32).view(np.float32)
        r = np.random.default_rng(99)
        hit = r.random(x.shape) < float(os.environ["FX_FUZZ_NAN"])
        x = x.copy()
        x[hit] = words[r.integers(0, words.size, size=int(hit.sum()))]
    failures, kernels, reasons = [], {}, {}
    for seed in range(first, first + count):
        rng = np.random.default_rng(500000 + seed)
        gen = stress_fuzz.random_program2 if seed % 2 else stress_fuzz.random_program
        max_regs = int(os.environ.get("FX_FUZZ_REGS", "50"))   # up to ~250: the larger VGPR builds; 700: beyond 224 rows the LDS interpreter (<= 640 rows), then the HIP C++ kernel
        text = gen(rng, int(rng.integers(4, 100 if max_regs <= 50 else (400 if max_regs <= 300 else 900))), int(rng.integers(2, max_regs)))
        b = A.Batch(N, 1, 0)
        if not b.load_text(text):
            continue
        lanes = {}
        if os.environ.get("FX_FUZZ_LANES"):
            written, read, tables = set(), [], []
            for ln in text.split("\n"):
                tok = [t.strip() for t in ln.replace(",", " ").split()]
                if len(tok) == 5 and tok[0] not in ("idelay", "xdelay", "skip"):
                    written.add(tok[1])
                    read += tok[2:5]
                    if tok[0] in ("log", "exp"):
                        tables.append(tok[3])
            candidates = sorted({r for r in read if r not in written and r not in ("in", "ccr", "noise", "0")})
            for name in sorted(set(tables)):
                if name in written:
                    continue
                v = rng.integers(0, 32, size=N).astype(np.float32)
                odd = rng.random(N) < 0.08
                v[odd] = rng.choice(np.array([-1.0, 32.0, 31.5, 1.0e10, -0.25, np.nan], np.float32), size=int(odd.sum()))
                lanes[name] = v
            for name in candidates:
                if name not in lanes and rng.random() < 0.3:
                    lanes[name] = rng.uniform(-1.0, 1.0, size=N).astype(np.float32)
            for name, v in lanes.items():
                if b.set_register_array(name, v) != 0:
                    lanes = None
                    break
            if lanes is None:
                continue
        y1 = b.process_block(x)
        y2 = b.process_block(x)
        kernels[b.info("kernel")] = kernels.get(b.info("kernel"), 0) + 1
        if b.info("kernel") < 9:
            reasons[b.tier_note()] = reasons.get(b.tier_note(), 0) + 1
        for n in (0, 1, 37, 64, 69):
            o = Oracle(1)
            o.load_text(text)
            for name, v in lanes.items():
                o.set_register(name, float(v[n]))
            r1 = o.process_block(x[:, n].copy())
            r2 = o.process_block(x[:, n].copy())
            if o.ood_flags() and not os.environ.get("FX_FUZZ_OOD"):   # FX_FUZZ_OOD=1: out-of-domain behaviour is compared too
                continue
            if o.ood_flags() and b.ood_flags() & o.ood_flags() != o.ood_flags():
                failures.append(seed)
                print("OOD FLAGS seed", seed, "instance", n, "oracle", o.ood_flags(), "batch (OR over instances)", b.ood_flags(), flush=True)
                break
            ok = same(r1, y1[:, n]) and same(r2, y2[:, n]) and b.instruction_counter_i(n) == o.instruction_counter()
            ok = ok and b.get_register_bits_i("ccr", n) == o.get_register_bits("ccr")
            if not ok:
                failures.append(seed)
                print("MISMATCH seed", seed, "instance", n, "kernel", b.info("kernel"), flush=True)
                break
        del b
    unexpected = {r: c for r, c in reasons.items() if r.startswith("interpreter") and "multi-pass program" not in r and "register file in LDS" not in r}
    if int(os.environ.get("FX_FUZZ_REGS", "50")) > 300:
        unexpected = {}   # (programs of 900 instructions: code beyond the hole or a branch's reach is a legitimate reason)
    if not os.environ.get("FX_KERNEL") and not os.environ.get("FX_INST_PER_LANE") and unexpected:
        # the default tier ran some programs on the interpreter for another reason than passes over the program (generated code
        # is one pass): a translation failed (the batch falls back silently)
        print("NOTE: interpreter fallbacks in default mode:", unexpected)
        failures.append("interpreter fallback")
    if reasons and not os.environ.get("FX_KERNEL"):
        print("below the translated tier:", sorted(reasons.items(), key=lambda kv: -kv[1])[:4])
    print("fuzz sweep:", count, "programs, kernels", kernels, "failures", failures)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
