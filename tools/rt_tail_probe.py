#!/usr/bin/env python3
"""Where does the tail of the real-time block time come from - the GPU or the host?  (tools/realtime_capacity.py reports
median / p99 / p99.9 of call -> return; the budget is judged at p99.9.)  Per block: the call's wall time and the kernel's own
time between its two HIP events; printed: the percentiles of both and of their difference (launch + wake-up of the waiting
host thread), for device-resident and host-fed (pinned, in place) PCM.

    python tools/rt_tail_probe.py [instances_device] [instances_host] [blocks]
"""
import ctypes as C
import gc
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "fx8010-emulator-core_amd", "python"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)


def pct(a, q):
    import numpy as np
    a = np.sort(np.asarray(a))
    return float(a[min(len(a) - 1, int(np.ceil(q * len(a))) - 1)])


def run(torch, A, progs, rt, n, mode, blocks, warm=400, paced=False):
    lib = A.load()
    b = A.Batch(n, 1, 0)
    assert b.load_text(progs.config5()), b.errors()
    ring = [progs.stimulus(n, rt.BLOCK, first_sample=k * rt.BLOCK) for k in range(rt.RING)]
    if mode == "host":
        xin = [torch.empty((rt.BLOCK, n), dtype=torch.float32).pin_memory() for _ in ring]
        for t, r in zip(xin, ring):
            t.numpy()[...] = r
        yout = torch.empty((rt.BLOCK, n), dtype=torch.float32).pin_memory()
        xp = [C.cast(t.data_ptr(), C.POINTER(C.c_float)) for t in xin]
        yp = C.cast(yout.data_ptr(), C.POINTER(C.c_float))
    else:
        xin = [torch.from_numpy(r).cuda() for r in ring]
        yout = torch.empty((rt.BLOCK, n), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        xp = [C.c_void_p(t.data_ptr()) for t in xin]
        yp = C.c_void_p(yout.data_ptr())
    h = b._h
    b.prepare(rt.BLOCK, True)
    wall, kern = [], []
    gc.collect()
    gc.disable()
    start = time.perf_counter_ns()
    late = 0
    for k in range(warm + blocks):
        if paced:   # one block per 666.667 us, as an audio interface would ask for them (a block that overran delays the next)
            due = start + int(k * rt.BUDGET_US * 1e3)
            if time.perf_counter_ns() > due + 1000:
                late += k >= warm
            while time.perf_counter_ns() < due:
                pass
        t0 = time.perf_counter_ns()
        if k % rt.SLIDER_EVERY == 0:
            lib.fxb_set_register(h, b"decay", C.c_float(rt.SLIDER[(k // rt.SLIDER_EVERY) % 4]))
        if mode == "host":
            rc = lib.fxb_process_block(h, xp[k % rt.RING], yp, rt.BLOCK)
        else:
            rc = lib.fxb_process_block_dev(h, xp[k % rt.RING], yp, rt.BLOCK, None) or lib.fxb_sync(h)
        t1 = time.perf_counter_ns()
        assert rc == 0
        if k >= warm:
            wall.append((t1 - t0) * 1e-3)
            kern.append(b.last_kernel_ms() * 1e3)
    gc.enable()
    rest = [w - k for w, k in zip(wall, kern)]
    print("%-6s N=%7d%s  %s" % (mode, n, "  paced (%d blocks started late)" % late if paced else "", "  ".join("%s: median %.1f p99 %.1f p99.9 %.1f max %.1f" % (name, pct(v, .5), pct(v, .99), pct(v, .999), max(v))
                                                   for name, v in (("call", wall), ("kernel", kern), ("call - kernel", rest)))), flush=True)
    # are the slow calls the slow kernels?
    import numpy as np
    w, kk = np.asarray(wall), np.asarray(kern)
    slow = w >= pct(wall, .99)
    print("       the slowest 1%% of the calls: kernel median %.1f, call - kernel median %.1f (all calls: %.1f, %.1f)" % (
        float(np.median(kk[slow])), float(np.median((w - kk)[slow])), float(np.median(kk)), float(np.median(w - kk))), flush=True)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        np.save(os.path.join(out, "rt_tail_%s_%d%s.npy" % (mode, n, "_paced" if paced else "")), np.stack([w, kk]).astype(np.float32))
    b.close()


def main():
    import torch
    import fx8010_amd as A
    import fx8010_programs as progs
    import realtime_capacity as rt
    nd = int(sys.argv[1]) if len(sys.argv) > 1 else 393216
    nh = int(sys.argv[2]) if len(sys.argv) > 2 else 147456
    blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
    print("environment:", {k: v for k, v in os.environ.items() if k.startswith(("ROC_", "HSA_", "HIP_", "GPU_", "AMD_"))}, flush=True)
    run(torch, A, progs, rt, nd, "device", blocks)
    run(torch, A, progs, rt, nh, "host", blocks)


if __name__ == "__main__":
    main()
