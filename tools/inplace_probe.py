#!/usr/bin/env python3
"""Which direction bounds a block processed in place on pinned host buffers?  32-sample blocks of config5 through
fxb_process_block_dev + fxb_sync with the input / the output in device memory or in pinned host memory (the kernel addresses either).
    python tools/inplace_probe.py [instances ...]"""
import ctypes as C
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "fx8010-emulator-core_amd", "python"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402

lib = A.load()
for n in [int(v) for v in sys.argv[1:]] or [65536, 131072, 196608]:
    b = A.Batch(n, 1, 0)
    assert b.load_text(P.config5())
    b.prepare(32, True)
    x = P.stimulus(n, 32)
    xh = torch.empty((32, n), dtype=torch.float32).pin_memory(); xh.numpy()[...] = x
    yh = torch.empty((32, n), dtype=torch.float32).pin_memory()
    xd = torch.from_numpy(x).cuda(); yd = torch.empty_like(xd)
    torch.cuda.synchronize()
    for name, i, o in (("device -> device", xd, yd), ("host   -> device", xh, yd), ("device -> host  ", xd, yh), ("host   -> host  ", xh, yh)):
        ts = []
        for k in range(600):
            t0 = time.perf_counter_ns()
            assert lib.fxb_process_block_dev(b._h, C.c_void_p(i.data_ptr()), C.c_void_p(o.data_ptr()), 32, None) == 0
            assert lib.fxb_sync(b._h) == 0
            ts.append((time.perf_counter_ns() - t0) * 1e-3)
        ts = np.sort(np.array(ts[100:]))
        print("N=%7d  %s  median %7.1f us  p99 %7.1f  (%.1f GB/s per direction in use)" % (n, name, ts[len(ts) // 2], ts[int(len(ts) * 0.99)], 32 * n * 4 / (ts[len(ts) // 2] * 1e-6) / 1e9))
    b.close()
