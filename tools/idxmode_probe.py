#!/usr/bin/env python3
"""Runs tools/idxmode_probe.S (see its header) on device 0 and prints, per variant, how many lane-iterations saw a
plain VALU instruction execute under a stale VGPR index mode right after s_set_gpr_idx_off.

    python tools/idxmode_probe.py [iterations per wave, default 2000000] [waves, default 8192] [control]
"""
import ctypes
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LLVM = "/opt/rocm/lib/llvm/bin"


def build():
    src, obj, co = (os.path.join(HERE, "idxmode_probe" + e) for e in (".S", ".o", ".hsaco"))
    if not os.path.exists(co) or os.path.getmtime(co) < os.path.getmtime(src):
        subprocess.run([LLVM + "/clang", "-x", "assembler-with-cpp", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", src, "-o", obj], check=True)
        subprocess.run([LLVM + "/ld.lld", "-shared", obj, "-o", co], check=True)
    return co


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    waves = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    co = build()
    hip = ctypes.CDLL("libamdhip64.so")

    def chk(r, what):
        if r != 0:
            raise RuntimeError("%s failed: %d" % (what, r))

    chk(hip.hipSetDevice(0), "hipSetDevice")
    mod, fn, dbuf = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    chk(hip.hipModuleLoad(ctypes.byref(mod), co.encode()), "hipModuleLoad")
    chk(hip.hipModuleGetFunction(ctypes.byref(fn), mod, b"idxmode_probe"), "hipModuleGetFunction")
    nbytes = waves * 256
    chk(hip.hipMalloc(ctypes.byref(dbuf), ctypes.c_size_t(nbytes)), "hipMalloc")
    chk(hip.hipMemset(dbuf, 0xFF, ctypes.c_size_t(nbytes)), "hipMemset")

    class Args(ctypes.Structure):
        _fields_ = [("out", ctypes.c_void_p), ("iters", ctypes.c_uint32), ("control", ctypes.c_uint32)]

    control = 1 if (len(sys.argv) > 3 and sys.argv[3] == "control") else 0
    a = Args(dbuf.value, iters, control)
    size = ctypes.c_size_t(ctypes.sizeof(a))
    cfg = (ctypes.c_void_p * 5)(1, ctypes.cast(ctypes.byref(a), ctypes.c_void_p), 2, ctypes.cast(ctypes.byref(size), ctypes.c_void_p), 3)
    chk(hip.hipModuleLaunchKernel(fn, waves, 1, 1, 64, 1, 1, 0, None, None, cfg), "launch")
    chk(hip.hipDeviceSynchronize(), "sync")
    host = (ctypes.c_uint32 * (waves * 64))()
    chk(hip.hipMemcpy(host, dbuf, ctypes.c_size_t(nbytes), 2), "hipMemcpy")
    per_variant = [0, 0, 0, 0]
    unwritten = 0
    for w in range(waves):
        for lane in range(64):
            v = host[w * 64 + lane]
            if v == 0xFFFFFFFF:
                unwritten += 1
            else:
                per_variant[w & 3] += v
    print(json.dumps({"mode": "positive control: no s_set_gpr_idx_off, every lane-iteration must count" if control else "probe", "iterations_per_wave": iters, "waves": waves, "waves_per_simd": waves / 1024.0,
                      "stale_index_mode_events": {"dst_after_idx_off": per_variant[0], "dst_after_fp64_then_idx_off": per_variant[1],
                                                  "dst_setpc_then_idx_off": per_variant[2], "src0_after_idx_off_fp64_consumer": per_variant[3]},
                      "lanes_unwritten": unwritten}))


if __name__ == "__main__":
    main()
