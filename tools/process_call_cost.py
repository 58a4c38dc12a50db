#!/usr/bin/env python3
"""Cost of the reference's own calling style on this library: one fx_process() per sample period (one launch + two copies per
call, batch of 1).  Prints microseconds per call and the emulated-instruction rate, next to the block interface on the same
single instance - so that nobody mistakes per-sample calls for the accelerated path."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fx8010-emulator-core_amd", "python"))
import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402

out = {}
for name in ("config1_shipped", "config1_logtube", "config5"):
    text = P.CONFIGS[name]()
    fd, path = tempfile.mkstemp(suffix=".da")
    with os.fdopen(fd, "wb") as fh:
        fh.write(text.encode())
    s = A.Single(1)
    assert s.load_file(path)
    x = P.stimulus(1, 4096)[:, 0].copy()
    for i in range(50):
        s.process(x[i:i + 1])
    n = 2000
    t0 = time.perf_counter()
    for i in range(n):
        s.process(x[i:i + 1])
    per_call = (time.perf_counter() - t0) / n
    instr = P.count_instructions(text)
    t0 = time.perf_counter()
    for _ in range(5):
        s.process_block(x)
    per_block = (time.perf_counter() - t0) / 5
    out[name] = {"instr_per_sample": instr, "us_per_process_call": round(per_call * 1e6, 1), "mips_per_sample_calls": round(instr / per_call / 1e6, 3),
                 "ms_per_4096_sample_block_call": round(per_block * 1e3, 3), "mips_block_calls_one_instance": round(instr * 4096 / per_block / 1e6, 1)}
    os.unlink(path)
print(json.dumps(out, indent=1))
