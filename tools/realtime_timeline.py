#!/usr/bin/env python3
"""Timeline of one real-time block: where do the copy-in, the kernels and the copy-out of a host-fed 32-sample block lie in time?
Run under rocprofv3 --kernel-trace --memory-copy-trace (tools/realtime_timeline.sh); prints nothing itself but a marker.
    python tools/realtime_timeline.py <instances> [blocks]"""
import ctypes as C
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "fx8010-emulator-core_amd", "python"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 40
b = A.Batch(n, 1, 0)
assert b.load_text(P.config5())
x = torch.empty((32, n), dtype=torch.float32).pin_memory()
x.numpy()[...] = P.stimulus(n, 32)
y = torch.empty((32, n), dtype=torch.float32).pin_memory()
b.prepare(32, True)
lib = A.load()
xp, yp = C.cast(x.data_ptr(), C.POINTER(C.c_float)), C.cast(y.data_ptr(), C.POINTER(C.c_float))
for k in range(blocks):
    assert lib.fxb_process_block(b._h, xp, yp, 32) == 0
print("done", n, blocks)
