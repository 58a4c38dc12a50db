#!/usr/bin/env python3
"""Which words of tests/golden/nonfinite.json (x86 reference) does the GPU path reproduce / not reproduce?  Per tier."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fx8010-emulator-core_amd", "python"))
import fx8010_amd as A  # noqa: E402

cases = json.load(open(os.path.join(ROOT, "tests", "golden", "nonfinite.json")))
for tier in ("default", "asm", "hip"):
    os.environ.pop("FX_KERNEL", None)
    if tier != "default":
        os.environ["FX_KERNEL"] = tier
    total = 0
    for c in cases:
        x1 = np.frombuffer(bytes.fromhex(c["input"]), dtype=np.uint32).view(np.float32)
        want = np.frombuffer(bytes.fromhex(c["output"]), dtype=np.uint32)
        N = 67
        b = A.Batch(N, 1, 0)
        assert b.load_text(c["program"])
        x = np.repeat(x1.reshape(-1, 1), N, axis=1).copy()
        y = b.process_block(x)
        got = np.ascontiguousarray(y[:, 5]).view(np.uint32)
        bad = np.nonzero(want != got)[0]
        regs = [(r, "%08x" % v, "%08x" % b.get_register_bits_i(r, 5)) for r, v in c["registers"].items() if b.get_register_bits_i(r, 5) != v]
        cnt = b.instruction_counter_i(5) == c["counter"]
        if bad.size or regs or not cnt:
            print(tier, c["name"], [(int(i), "in %08x" % x1.view(np.uint32)[i], "ref %08x" % want[i], "gpu %08x" % got[i]) for i in bad[:8]], regs, cnt)
        total += bad.size
    print(tier, "differing output words:", total)
