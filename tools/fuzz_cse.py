"""Differential fuzz of the translated tier's product cache: programs over a tiny vocabulary (three registers, three
coefficients, the input, the CCR) so that the same product is asked for again with its register unchanged, rewritten,
rewritten under a SKIP shadow, or first computed inside one (stress_fuzz.product_cache_program).

    python tools/fuzz_cse.py [first_seed] [count]
"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fx8010-emulator-core_amd/python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

import fx8010_amd as A  # noqa: E402
import stress_fuzz  # noqa: E402
from pyoracle import Oracle  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    N, S = 70, 24
    failures, kernels = [], {}
    for seed in range(first, first + count):
        rng = np.random.default_rng(770000 + seed)
        text = stress_fuzz.product_cache_program(rng, int(rng.integers(4, 90)))
        x = (rng.uniform(-1.0, 1.0, size=(S, N)) * rng.choice([1.0, 0.5, 1e-3], size=(1, N))).astype(np.float32)
        x *= np.float32(float(os.environ.get("FX_FUZZ_SCALE", "1")))   # > 1: waves leave the fast streams (and with them the product cache)
        b = A.Batch(N, 1, 0)
        if not b.load_text(text):
            continue
        y1 = b.process_block(x)
        y2 = b.process_block(x)
        kernels[b.info("kernel")] = kernels.get(b.info("kernel"), 0) + 1
        for n in (0, 1, 37, 64, 69):
            o = Oracle(1)
            o.load_text(text)
            r1 = o.process_block(x[:, n].copy())
            r2 = o.process_block(x[:, n].copy())
            ok = np.array_equal(r1.view(np.uint32), y1[:, n].view(np.uint32)) and np.array_equal(r2.view(np.uint32), y2[:, n].view(np.uint32))
            ok = ok and b.instruction_counter_i(n) == o.instruction_counter()
            for reg in ("a", "b", "c", "t", "ccr"):
                ok = ok and b.get_register_bits_i(reg, n) == o.get_register_bits(reg)
            if not ok:
                failures.append(seed)
                print("MISMATCH seed", seed, "instance", n, "kernel", b.info("kernel"), flush=True)
                break
        del b
    print("cse fuzz:", count, "programs, kernels", kernels, "failures", failures)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
