#!/usr/bin/env python3
"""When do the wavefronts of a launch finish?  In the DIAGNOSTICS build of the library (make -C fx8010-emulator-core_amd/csrc diag;
fx_knobs.hpp) FX_XLATE_ENDSTAMP=1 makes the generated code store the 100 MHz clock, behind the last sample, into word [wavefront]
of a buffer of the handle's own (fxb_diag_read_end_stamps, csrc/fx_diag.h - never into an output element): the spread of these
stamps over the 4 096 wavefronts of the 1/8 shard says how long SIMDs drain while the launch waits for its last wavefront.

    python tools/wave_end_probe.py [config] [instances] [samples]      (builds and loads the diagnostics library itself)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fx8010-emulator-core_amd/python"))
DIAG = os.path.join(ROOT, "fx8010-emulator-core_amd", "csrc", "build", "diag", "libfx8010_amd.so")
if not os.path.exists(DIAG):
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "fx8010-emulator-core_amd", "csrc"), "diag"])
os.environ["FX8010_AMD_LIB"] = DIAG          # before the binding is imported: it reads the variable once
os.environ["FX_XLATE_ENDSTAMP"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "config5"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    lib = A.load()
    assert lib.fxb_diag_build() == 1, "not the diagnostics build"
    x = torch.empty((S, N), dtype=torch.float32, device="cuda").uniform_(-0.9, 0.9)
    y = torch.empty_like(x)
    b = A.Batch(N, 1, 0)
    assert b.load_text(P.CONFIGS[config]()), b.errors()
    for _ in range(3):
        b.process_block_dev(x.data_ptr(), y.data_ptr(), S)
    b.sync()
    ms = b.last_kernel_ms()
    words = (N + 63) // 64
    raw = np.zeros(words, dtype=np.uint32)
    lib.fxb_diag_read_end_stamps.restype = ctypes.c_int
    lib.fxb_diag_read_end_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    got = lib.fxb_diag_read_end_stamps(b._h, raw.ctypes.data, words)
    assert got == words, (got, b.last_error() if hasattr(b, "last_error") else "")
    stamps = raw.astype(np.int64)
    t = (stamps - stamps.min()) * 1e-5          # ms after the first wavefront to finish (100 MHz ticks)
    t = np.where(t > 1e4, t - 2.0 ** 32 * 1e-5, t)
    t -= t.min()
    q = np.percentile(t, [0, 1, 10, 25, 50, 75, 90, 99, 100])
    print("%s, %d instances (%d wavefronts, %.1f per SIMD), %d samples: kernel %.3f ms (kernel id %d)" % (config, N, stamps.size, stamps.size / 1024.0, S, ms, b.info("kernel")))
    print("wavefronts finish, ms after the first one: percentiles 0 / 1 / 10 / 25 / 50 / 75 / 90 / 99 / 100 = " + " / ".join("%.3f" % v for v in q))
    print("mean %.3f ms before the last one = %.1f %% of the launch during which the average wavefront slot is already empty" % (t.max() - t.mean(), 100.0 * (t.max() - t.mean()) / ms))


if __name__ == "__main__":
    main()
