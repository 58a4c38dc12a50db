#!/usr/bin/env python3
"""How many emulated FX8010s does one MI355X carry in REAL TIME - measured the way the reference measures itself.

The reference's only timing idea (source/main.cpp:99-155, include/FX8010.h:37-38): one AUDIOBLOCKSIZE = 32-sample block at
48 kHz must be done within 666.667 us, while a slider moves (main.cpp:107-114: setRegisterValue("volume", ...) with the values
0.1 / 0.25 / 0.5 / 1.0).  Here: program = config5 (the 512-instruction reverb of BASELINE configs[4], 8192-sample xTRAM), N
instances per block call, PCM **host-fed from pinned buffers** through fxb_process_block (copy in, kernel, copy out, done when
the output is in host memory), after fxb_prepare; the declared control `decay` takes the harness's four values in turn, a new
one every 8th block (a fill of one register row - no code is generated in the timed region: checked).  Per N: median / p99 /
p99.9 / max of the block time, the kernel's own time, the PCIe rate, and whether p99.9 <= 666.667 us; the answer is the
largest such N.  The same with device-resident PCM (fxb_process_block_dev + fxb_sync per block: what a host that already has
its audio on the GPU pays).  After the timed region the device's LAST block is compared bit for bit with the CPU oracle, which
replays every block of the run for sampled instances (same PCM, same slider schedule).

    python tools/realtime_capacity.py [--blocks 5000] [--instances 4096,...] [--json profiles/r05_realtime.json]
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "fx8010-emulator-core_amd", "python"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

BLOCK = 32                      # AUDIOBLOCKSIZE, /root/reference/include/FX8010.h:38
SAMPLERATE = 48000              # /root/reference/include/FX8010.h:37
BUDGET_US = BLOCK / SAMPLERATE * 1e6   # 666.667 us: "Erlaubtes Zeitfenster ohne Dropouts", source/main.cpp:155
SLIDER = (0.1, 0.25, 0.5, 1.0)  # source/main.cpp:80
SLIDER_EVERY = int(os.environ.get("FX_RT_SLIDER_EVERY", "8"))   # a new value every 8th block (the environment can slow the slider down: a slider that
                                # rests longer than 8192 sample periods cools down and is folded into the code again, on the builder thread)
RING = 8                        # distinct PCM blocks, fed in turn


def percentiles(us):
    import numpy as np
    a = np.sort(np.asarray(us, dtype=np.float64))
    pick = lambda q: float(a[min(len(a) - 1, int(np.ceil(q * len(a))) - 1)])
    return {"median_us": round(pick(0.5), 1), "p99_us": round(pick(0.99), 1), "p999_us": round(pick(0.999), 1), "max_us": round(float(a[-1]), 1),
            "mean_us": round(float(a.mean()), 1), "blocks": int(len(a)), "over_budget": int((a > BUDGET_US).sum())}


def measure(torch, A, progs, n, blocks, warm, mode, check=4, config="config5", control="decay", shards=1):
    """one N, one mode ("host": pinned host PCM through fxb_process_block; "device": resident PCM, launch + sync per block).
    shards > 1 (host mode): the handle is made of that many shards ON THE SAME GPU (fxb_create_on_devices with the ordinal
    repeated): each shard has its own host thread and stream and copies its own columns of the caller's [S][N] buffers, so the
    copy-in of one shard, the kernel of another and the copy-out of a third overlap - the block's PCIe time is no longer
    serialised around its kernel."""
    import numpy as np
    from pyoracle import Oracle

    text = progs.CONFIGS[config]()
    lib = A.load()
    b = A.Batch(n, 1, 0) if shards <= 1 else A.Batch(n, 1, devices=[0] * shards)
    if not b.load_text(text):
        raise RuntimeError("load failed: %s" % b.errors())
    ring = [progs.stimulus(n, BLOCK, first_sample=k * BLOCK) for k in range(RING)]
    if mode == "host":
        xin = [torch.empty((BLOCK, n), dtype=torch.float32).pin_memory() for _ in range(RING)]
        for t, r in zip(xin, ring):
            t.numpy()[...] = r
        yout = torch.empty((BLOCK, n), dtype=torch.float32).pin_memory()
        xp = [C.cast(t.data_ptr(), C.POINTER(C.c_float)) for t in xin]
        yp = C.cast(yout.data_ptr(), C.POINTER(C.c_float))
    else:
        xin = [torch.from_numpy(r).cuda() for r in ring]
        yout = torch.empty((BLOCK, n), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        xp = [C.c_void_p(t.data_ptr()) for t in xin]
        yp = C.c_void_p(yout.data_ptr())
    h = b._h
    b.prepare(BLOCK, True)                       # the code for 32-sample blocks (and the variant with the controls in rows) before the stream starts
    key = control.encode()

    def block(k):
        if k % SLIDER_EVERY == 0:
            rc = lib.fxb_set_register(h, key, C.c_float(SLIDER[(k // SLIDER_EVERY) % len(SLIDER)]))
            assert rc == 0, rc
        if mode == "host":
            rc = lib.fxb_process_block(h, xp[k % RING], yp, BLOCK)
        else:
            rc = lib.fxb_process_block_dev(h, xp[k % RING], yp, BLOCK, None)
            rc = rc or lib.fxb_sync(h)
        if rc != 0:
            raise RuntimeError("block %d failed (%d): %s" % (k, rc, b.last_error()))

    for k in range(warm):                        # first control touch, the tuner's trials, clocks
        block(k)
    b.prepare(BLOCK, True)
    builds0 = (b.info("xlate_builds"), b.info("xlate_background_builds"))
    times, kernel = [], []
    gc.collect()
    gc.disable()
    try:
        t_start = time.perf_counter()
        for k in range(warm, warm + blocks):
            t0 = time.perf_counter_ns()
            block(k)                             # (the slider write of every 8th block is inside the block's time: it is the caller's)
            times.append((time.perf_counter_ns() - t0) * 1e-3)
            if k % 64 == 0:
                kernel.append(b.last_kernel_ms() * 1e3)
        wall = time.perf_counter() - t_start
    finally:
        gc.enable()
    builds1 = (b.info("xlate_builds"), b.info("xlate_background_builds"))
    total = warm + blocks
    y_last = (yout.numpy() if mode == "host" else yout.cpu().numpy()).copy()
    # parity: the oracle replays every block of the run for sampled instances
    picks = sorted(set([0, min(63, n - 1), n // 2, n - 1]))[:check]
    ok = True
    for inst in picks:
        o = Oracle(1)
        assert o.load_text(text)
        ref = None
        cols = [np.ascontiguousarray(r[:, inst]) for r in ring]
        for k in range(total):
            if k % SLIDER_EVERY == 0:
                o.set_register(control, SLIDER[(k // SLIDER_EVERY) % len(SLIDER)])
            ref = o.process_block(cols[k % RING])
        ok = ok and bool(np.array_equal(ref.view(np.uint32), np.ascontiguousarray(y_last[:, inst]).view(np.uint32)))
        ok = ok and b.instruction_counter_i(inst) == o.instruction_counter()
    res = percentiles(times)
    kk = sorted(v for v in kernel if v > 0)
    # what the slider costs: the blocks that start with a control write beside the others
    moved = [t for k, t in zip(range(warm, total), times) if k % SLIDER_EVERY == 0]
    plain = [t for k, t in zip(range(warm, total), times) if k % SLIDER_EVERY != 0]
    if moved and plain:
        pm, pp = percentiles(moved), percentiles(plain)
        res.update({"control_blocks": {"median_us": pm["median_us"], "p99_us": pm["p99_us"], "max_us": pm["max_us"], "blocks": pm["blocks"]},
                    "other_blocks": {"median_us": pp["median_us"], "p99_us": pp["p99_us"], "p999_us": pp["p999_us"], "max_us": pp["max_us"], "blocks": pp["blocks"]}})
    res.update({
        "instances": n, "mode": mode, "shards_on_the_gpu": shards, "budget_us": round(BUDGET_US, 3), "within_budget_p999": res["p999_us"] <= BUDGET_US,
        "kernel_us_median": round(kk[len(kk) // 2], 1) if kk else None,
        "pcie_GBps_each_way_at_median": round(BLOCK * n * 4 / (res["median_us"] * 1e-6) / 1e9, 2) if mode == "host" else None,
        "realtime_factor_at_median": round(BUDGET_US / res["median_us"], 2),
        "emulated_mips_sustained": round(progs.count_instructions(text) * BLOCK * n * blocks / wall / 1e6, 1),
        "translations_in_timed_region": [builds1[0] - builds0[0], builds1[1] - builds0[1]],
        "tier": b.tier_note(), "parity_instances": len(picks), "parity_ok": ok, "blocks_replayed_by_oracle": total,
    })
    b.close()
    del xin, yout
    if mode == "device":
        torch.cuda.empty_cache()
    return res


def capacity(rows):
    """largest N such that it AND every smaller N measured keep their p99.9 block time within the budget (None when even the
    smallest does not): a lucky row above a failing one does not count"""
    best = None
    for r in sorted(rows, key=lambda r: r["instances"]):
        if not (r["within_budget_p999"] and r["parity_ok"]):
            break
        best = r["instances"]
    return best


def run(torch, A, progs, instances, blocks, warm, modes=("host", "device"), log=None, shards=1):
    out = {"what": "32-sample blocks at 48 kHz against %.3f us (the reference's own real-time measure, source/main.cpp:99-155); program config5 (512 instructions, "
                   "8192-sample xTRAM); control `decay` takes 0.1 / 0.25 / 0.5 / 1.0 in turn, one step every 8th block; times are call -> output in host memory "
                   "(host mode, pinned buffers) or call -> fxb_sync (device mode), on the caller's clock" % BUDGET_US,
           "budget_us": round(BUDGET_US, 3), "block_samples": BLOCK, "blocks_per_point": blocks, "warmup_blocks": warm, "rows": []}
    for mode in modes:
        for n in instances:
            r = measure(torch, A, progs, n, blocks, warm, mode, shards=shards if mode == "host" else 1)
            out["rows"].append(r)
            if log:
                log(("%d shards " % shards if shards > 1 and mode == "host" else "") + "%-6s N=%7d  median %7.1f  p99 %7.1f  p99.9 %7.1f  max %8.1f us  kernel %6.1f us  %s  parity %s" % (
                    mode, n, r["median_us"], r["p99_us"], r["p999_us"], r["max_us"], r["kernel_us_median"] or -1,
                    "REAL TIME" if r["within_budget_p999"] else "over budget", "ok" if r["parity_ok"] else "MISMATCH")
                    + ("  (blocks that move the slider: median %.1f, the others %.1f, p99.9 %.1f)" % (
                        r["control_blocks"]["median_us"], r["other_blocks"]["median_us"], r["other_blocks"]["p999_us"]) if "control_blocks" in r else ""))
        out["capacity_%s_fed" % mode] = capacity([r for r in out["rows"] if r["mode"] == mode])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--instances", default="4096,16384,65536,98304,131072,147456,163840,180224,196608")
    ap.add_argument("--device-instances", default="", help="instance counts of the device-resident rows (default: 4096 ... 589824)")
    ap.add_argument("--shards", default="1", help="host-fed rows: shards on the one GPU, e.g. 1,4 (every value is a sweep of its own)")
    ap.add_argument("--no-device", action="store_true", help="skip the device-resident rows")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    import torch  # first: its HIP runtime is the one the library binds to

    import fx8010_amd as A
    import fx8010_programs as progs
    inst = [int(v) for v in args.instances.split(",") if v]
    dev_inst = [int(v) for v in args.device_instances.split(",") if v] or [4096, 65536, 131072, 262144, 393216, 458752, 524288, 557056, 589824]
    log = lambda s: print(s, flush=True)
    out = None
    for k in [int(v) for v in args.shards.split(",") if v]:
        part = run(torch, A, progs, inst, args.blocks, args.warmup, ("host",), log, shards=k)
        part["capacity_host_fed_by_shards"] = {str(k): part["capacity_host_fed"]}
        if out is None:
            out = part
        else:
            out["rows"] += part["rows"]
            out["capacity_host_fed_by_shards"][str(k)] = part["capacity_host_fed"]
            out["capacity_host_fed"] = max([v for v in out["capacity_host_fed_by_shards"].values() if v] or [None], key=lambda v: v or 0)
    out["capacity_device_fed"] = None
    if not args.no_device:
        dev = run(torch, A, progs, dev_inst, args.blocks, args.warmup, ("device",), log)
        out["rows"] += dev["rows"]
        out["capacity_device_fed"] = dev["capacity_device_fed"]
    out["gpu"] = torch.cuda.get_device_name(0)
    print("largest N within %.3f us at p99.9: host-fed %s, device-resident PCM %s" % (BUDGET_US, out["capacity_host_fed"], out["capacity_device_fed"]))
    if args.json:
        with open(args.json, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
