"""Scale stress: every benchmark program at 262144 instances, many launches, sampled instances checked
against the oracle.  (Catches faults that only show with many wavefronts per SIMD.)

    python tools/stress_scale.py [launches]
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

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for name in ("config4", "config2", "config3", "config5"):
    N, S = 262144, 16
    text = P.CONFIGS[name]()
    b = A.Batch(N, 1, 0)
    assert b.load_text(text)
    x = P.stimulus(N, S)
    for i in range(launches):
        y = b.process_block(x)
    bad = []
    for n in (0, 63, 64, 4097, N // 2 + 3, N - 1):
        o = Oracle(1)
        o.load_text(text)
        for i in range(launches):
            ref = o.process_block(x[:, n].copy())
        if not np.array_equal(ref.view(np.uint32), y[:, n].view(np.uint32)):
            bad.append(n)
    print(name, "kernel", b.info("kernel"), "launches", launches, "bad", bad, "ood", b.ood_flags(), flush=True)
    assert not bad
    del b
print("stress ok")
