"""Cost of the translated code per VALU instruction class: programs made of one instruction shape, 262144 instances.

    python tools/op_cost.py
"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fx8010-emulator-core_amd/python"))
import numpy as np  # noqa: E402

import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402

HDR = "input in 0\noutput out 0\n" + "".join("static r%d\n" % i for i in range(8))
SHAPES = {
    "macs r,r,in,c      (mul, add, med3)": "macs r%d, r%d, in, 0.3",
    "macs r,0,r,c       (mul, add)": "macs r%d, 0, r%d, 0.3",
    "acc3 r,r,in,c      (add, add, med3)": "acc3 r%d, r%d, in, 0.3",
    "interp r,r,c,r'    (mul, cvt, cvt, fma64, cvt)": "interp r%d, r%d, 0.3, r7",
    "interp r,r,c,in    (same + med3)": "interp r%d, r%d, 0.3, in",
}
N, S, REP = 262144, 128, 256
x = P.stimulus(N, S)
for name, shape in SHAPES.items():
    body = "\n".join(shape % (i % 7, i % 7) for i in range(REP))
    text = HDR + "macs r7, 0, in, 0.5\n" + body + "\nmacs out, 0, r0, 0.5\nend"
    b = A.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    b.process_block(x)
    b.process_block(x)
    ms = b.last_kernel_ms()
    fe = A.FrontEnd(1)
    fe.load_text(text)
    _, listing = fe.translate(0, 0)
    valu = sum(1 for l in listing.split("\n") if l.startswith("v_"))
    per_instr_ns = ms * 1e6 / (S * (REP + 3)) / (N / 64 / 1024)  # ns of SIMD time per emulated instruction of one wave
    print("%-52s %6.3f ms  %5d VALU/sample  %.2f ns per wave-instruction  (%.2f ns per VALU)" % (name, ms, valu, per_instr_ns, ms * 1e6 / S / valu / (N / 64 / 1024)))
    del b
