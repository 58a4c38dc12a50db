#!/usr/bin/env python3
"""Does the size of the generated loop matter?  config5's reverb with its diffusion fill cut to P instructions (the mix of the
fill is the same at every length), 262 144 instances = 4 wavefronts per SIMD, device-resident stimulus: emulated instructions
per second and generated code bytes against P.  (The instruction cache of a CU pair holds 64 KB.)

    python tools/code_size_probe.py [P ...]
"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fx8010-emulator-core_amd/python"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402


OPS = ("macs", "macsn", "macints", "macintw", "acc3", "macmv", "macw", "macwn", "skip", "andxor", "tstneg", "limit", "limitn", "log", "exp", "interp",
       "idelay", "xdelay", "end")


def program(n_instr):
    lines = P.CONFIGS["config5"]().split("\n")
    first = next(i for i, ln in enumerate(lines) if ln.startswith("macsn v,"))
    last = next(i for i, ln in enumerate(lines) if ln.startswith("xdelay write"))
    total = sum(1 for ln in lines if ln.split() and ln.split()[0] in OPS)
    cut = max(0, min(total - n_instr, last - first))
    cut -= cut % 5    # whole all-pass sections
    return "\n".join(lines[:last - cut] + lines[last:]), total - cut


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [512, 384, 256, 192, 128, 64]
    N, S = 262144, 1024
    x = torch.empty((S, N), dtype=torch.float32, device="cuda").uniform_(-0.9, 0.9)
    y = torch.empty_like(x)
    for want in sizes:
        text, n_instr = program(want)
        b = A.Batch(N, 1, 0)
        assert b.load_text(text), b.errors()
        for _ in range(2):
            b.process_block_dev(x.data_ptr(), y.data_ptr(), S)
        b.sync()
        ms = []
        for _ in range(6):
            b.process_block_dev(x.data_ptr(), y.data_ptr(), S)
            b.sync()
            ms.append(b.last_kernel_ms())
        t = float(np.median(ms))
        print("P = %3d instructions: kernel %7.3f ms, %6.2f e12 instr/s, code %6d bytes, VALU per instruction %.3f, kernel id %d" % (
            n_instr, t, n_instr * S * N / (t * 1e-3) / 1e12, b.info("xlate_code_bytes"), b.info("xlate_valu") / n_instr, b.info("kernel")), flush=True)
        del b


if __name__ == "__main__":
    main()
