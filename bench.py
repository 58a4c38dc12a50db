#!/usr/bin/env python3
"""bench.py — aggregate emulated DSP MIPS of the MI355X batch FX8010 interpreter.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config config5] [--samples S] [--instances M]

One "step" = one pass of the hot path over one batch of synthetic PCM: fxb_process_block_dev()
for S sample periods on this rank's M instances (inputs already resident in HBM).  The default
workload is the per-GPU shard of BASELINE.json configs[4] — the configuration the north-star
target (>= 1e12 emulated instr/s on 8 GPUs) is quoted on: 262 144 instances/GPU of the 512-instr
reverb with an 8192-sample xTRAM, so that `--gpus 8` is exactly configs[4] (2 097 152 instances,
weak scaling, no collective on the data path).  `--config config2|config3|config4` select the
other single-GPU configurations with their BASELINE.json instance counts.

For N > 1 the driver launches this file under torch.distributed.run (one rank per GPU); ranks
only meet in the barriers around the timed region and in the MAX-reduction of the elapsed time.

Output: ONE JSON line (rank 0) with the contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "fx8010-emulator-core_amd", "python"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="config5", choices=["config2", "config3", "config4", "config5", "tram_bound"])
    ap.add_argument("--samples", type=int, default=4096, help="sample periods per step (block length S; SURVEY 8d: 4096)")
    ap.add_argument("--instances", type=int, default=0, help="instances per GPU (0 = BASELINE.json's count)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (rank 0, N=1 only); 0 disables")
    ap.add_argument("--extra-configs", action="store_true", help="also report untimed single-launch MIPS of the other configs")
    return ap.parse_args()


def device_stimulus(torch, n_inst, n_samples, first_instance, device):
    """fx8010_programs.stimulus() on the device (same counter-based hash, same bits), built in slabs of 256
    sample periods so that the int64 temporaries stay small next to the [S, N] fp32 result."""
    M = 0xFFFFFFFF
    out = torch.empty((n_samples, n_inst), dtype=torch.float32, device=device)
    n = (torch.arange(n_inst, dtype=torch.int64, device=device) + first_instance)[None, :] * 0x9E3779B1 + 0xF8010
    for s0 in range(0, n_samples, 256):
        s1 = min(n_samples, s0 + 256)
        s = torch.arange(s0, s1, dtype=torch.int64, device=device)[:, None]
        x = (n + s * 0x85EBCA77) & M
        x = x ^ (x >> 16)
        x = (x * 0x85EBCA6B) & M
        x = x ^ (x >> 13)
        x = (x * 0xC2B2AE35) & M
        x = x ^ (x >> 16)
        i32 = torch.where(x >= 2 ** 31, x - 2 ** 32, x)
        out[s0:s1] = i32.to(torch.float32) * (2.0 ** -31) * 0.9
    return out


def usable_cores():
    """Host threads this process may really use: affinity, capped by the cgroup CPU quota (the GPU
    box hands one GPU's share of a large host: 16 cores) — oversubscribing would only time the throttle."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def measured_traffic(config, n_inst, n_samples):
    """HBM bytes per launch from committed rocprofv3 PMC passes (profiles/*hbm_traffic*.json: FETCH_SIZE and
    WRITE_SIZE in their own passes, gfx950 read correction applied) when one exists for exactly this workload."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic*.json"))):
        try:
            d = json.load(open(f))
            w = d["workload"]
            if w["config"] == config and w["instances"] == n_inst and w["samples"] == n_samples:
                return float(d["hbm_bytes_per_launch"])
        except (OSError, ValueError, KeyError):
            pass
    return None


def cpu_baseline(text, budget_s):
    """Time the reference (or, without oracle/_ref, the C port) on the host cores: bounded sample."""
    import tempfile

    import numpy as np

    import fx8010_programs as progs
    from pyoracle import Oracle, Reference

    cls, kind = (Reference, "reference") if Reference.available() else (Oracle, "port")
    cores = usable_cores()
    fd, path = tempfile.mkstemp(suffix=".da")
    with os.fdopen(fd, "wb") as fh:
        fh.write(text.encode())
    try:
        stim = progs.stimulus(1, 4096)[:, 0].copy()
        t_cal, instr, _ = cls.bench(path, 4096, cores, stim)
        per_sample = instr / (4096.0 * cores)
        samples = int(max(4096, min(4096 * budget_s / max(t_cal, 1e-6), 50_000_000)))
        secs, _, _ = cls.bench(path, samples, cores, stim)
        # the reference counts instructions in an int (FX8010.h getInstructionCounter): it wraps on long runs,
        # so the total comes from the per-sample count of the short calibration run
        mips = per_sample * samples * cores / secs / 1e6
        # one thread alone, the reference's own calling style (comparable to README's 200 MIPS)
        n1 = max(4096, samples // 8)
        s1, _, _ = cls.bench(path, n1, 1, stim)
        i1 = per_sample * n1
        return {"value": round(mips, 1), "unit": "MIPS", "cores": cores, "kind": kind,
                "sample": "%d host threads x %d process() calls each of the same program (%.0f instr/sample), %.1f s" % (cores, samples, per_sample, secs),
                "single_thread_mips": round(i1 / s1 / 1e6, 1)}
    finally:
        os.unlink(path)


def main():
    args = parse()
    import torch  # first: its HIP runtime is the one the library binds to in this process

    import fx8010_amd
    import fx8010_programs as progs

    import fx8010_shard as shard

    rank, local, world = shard.env_world()
    if world == 1:
        local = 0
    # rehearsal on a box with fewer GPUs than ranks (FX_BENCH_REHEARSAL=1): ranks share the devices round-robin and
    # meet over gloo instead of RCCL, which refuses two ranks on one device - same code path otherwise
    rehearsal = os.environ.get("FX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = shard.init_process_group("gloo" if rehearsal else "nccl", dev)  # RCCL; None for a single process
    reduce_dev = "cpu" if rehearsal else dev

    text = progs.CONFIGS[args.config]()
    n_inst = args.instances or progs.CONFIG_INSTANCES[args.config]
    S = args.samples
    P = progs.count_instructions(text)

    batch = fx8010_amd.Batch(n_inst, 1, local)
    if not batch.load_text(text):
        raise RuntimeError("program failed to load: %s" % batch.errors())
    first_instance, _ = shard.weak_shard(n_inst, rank)  # weak scaling: every GPU owns n_inst instances
    x = device_stimulus(torch, n_inst, S, first_instance, dev)  # [S, N] resident in HBM
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    # a non-default torch stream: the kernel is launched on it through the C ABI, so the
    # torch.cuda.Events below (HIP events on that same stream) bracket exactly the launches
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    def step():
        batch.process_block_dev(x.data_ptr(), y.data_ptr(), S, stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    c0 = batch.instruction_counter()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms = ev0.elapsed_time(ev1) / max(args.steps, 1)  # HIP events on the launch stream
    last_ms = batch.last_kernel_ms()
    executed = batch.instruction_counter() - c0  # reference counting: END/SKIP count, skipped don't
    ood = batch.ood_flags()

    elapsed = shard.reduce_scalar(dist, elapsed, "max", reduce_dev)            # slowest rank
    executed_all = shard.reduce_scalar(dist, executed, "sum", reduce_dev)      # whole job

    if rank == 0:
        mips = executed_all / elapsed / 1e6
        tram_ops = batch.info("tram_ops")
        rows = batch.info("num_rows")
        kid = batch.info("kernel")
        vg = (0, 0, 64, 72, 80, 96, 128, 168, 256)
        if kid == 0:
            kernel_kind = "fx_step_block (HIP C++)"
        elif kid == 1:
            kernel_kind = "fx_interp_lds (gfx950 asm interpreter, LDS register file)"
        elif kid <= 8:
            kernel_kind = "fx_interp_v%d (gfx950 asm interpreter, VGPR register file)" % vg[kid]
        else:
            kernel_kind = ("fx_xlate_v%d (program translated to gfx950 code: %d records inline, %d handler calls, %d saturations elided, %d code bytes)"
                           % (vg[kid - 7],
                              batch.info("xlate_inlined"), batch.info("xlate_called"), batch.info("xlate_unsaturated"), batch.info("xlate_code_bytes")))
        # algorithmic HBM bytes of ONE launch on ONE GPU: PCM in+out, every executed
        # TRAM read/write, and the once-per-block register-file spill/fill
        bytes_per_inst_sample = 4 * (1 + 1) + 4 * tram_ops
        algo_bytes = float(n_inst) * (S * bytes_per_inst_sample + 2 * 4 * (rows + 9))
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "aggregate emulated DSP MIPS (instr x samples x instances / s)",
            "value": round(mips, 1),
            "unit": "MIPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d instances/GPU x %d-instr program, block of %d samples, mono 48 kHz" % (args.config, n_inst, P, S),
                "instances_total": n_inst * world,
                "instr_per_sample_static": P,
                "instr_per_sample_executed": round(executed / float(args.steps * S * n_inst), 3),
                "tram_ops_per_sample": tram_ops,
                "lane_register_rows": rows,
                "parity_domain_flags": ood,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": measured_traffic(args.config, n_inst, S),
                "kernel": kernel_kind,
                "kernel_ms": round(kernel_ms, 4),
                "kernel_ms_last_launch": round(last_ms, 4),
                "algorithmic_bytes_per_launch": algo_bytes,
                "note": "the interpreter is instruction-issue bound (>= 12 emulated instr per algorithmic HBM byte); see DESIGN.md section 5",
                "emulated_instr_per_s_per_gpu": round(executed / (kernel_ms * 1e-3 * args.steps), 1),
            },
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(text, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
