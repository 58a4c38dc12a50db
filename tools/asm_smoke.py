import sys, os
sys.path.insert(0,'oracle'); sys.path.insert(0,'fx8010-emulator-core_amd/python')
import numpy as np, fx8010_amd as A, fx8010_programs as P
from pyoracle import Oracle
def run(name, text, N, S):
    x = P.stimulus(N, S)
    b = A.Batch(N, 1, 0); assert b.load_text(text)
    y = b.process_block(x)
    print(name, "kernel", b.info("kernel"), "rows", b.info("lds_bytes_per_wg")//256, flush=True)
    bad = 0
    for n in (0, 1, 63, N-1):
        o = Oracle(1); o.load_text(text); ref = o.process_block(x[:, n].copy())
        ok = np.array_equal(ref.view(np.uint32), y[:, n].view(np.uint32)) and b.instruction_counter_i(n) == o.instruction_counter()
        if not ok:
            bad += 1
            d = np.nonzero(ref.view(np.uint32) != y[:, n].view(np.uint32))[0]
            print("  MISMATCH inst", n, "first at", d[:3], ref[d[:3]], y[d[:3], n], b.instruction_counter_i(n), o.instruction_counter())
    print("  ->", "OK" if not bad else "FAIL", "ood", b.ood_flags(), flush=True)
run("cfg1", P.config1_shipped(), 70, 5)
run("cfg2", P.config2(), 70, 33)
run("cfg3", P.config3(), 130, 40)
run("cfg4", P.config4(), 130, 40)
run("cfg5", P.config5(), 130, 40)
