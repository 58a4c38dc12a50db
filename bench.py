#!/usr/bin/env python3
"""bench.py — aggregate emulated DSP MIPS of the MI355X batch FX8010 interpreter.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak] [--config config5] [--samples S]
                    [--instances M] [--sharded] [--no-extras] [--parity-instances P]

One "step" = one pass of the hot path over one batch of synthetic PCM: fxb_process_block_dev()
for S sample periods on this rank's instances (inputs already resident in HBM).  The default
workload is BASELINE.json configs[4] - the configuration the north-star target (>= 1e12 emulated
instr/s on 8 GPUs) is quoted on: 2 097 152 instances of the 512-instr reverb with an 8192-sample
xTRAM each (64 GiB of delay memory), "sharded across 8 MI355X".  That is a FIXED total, so the
default is STRONG scaling: `--gpus N` splits the 2 097 152 instances into N contiguous ranges
(fx8010_shard.shard_range; no collective on the data path) - all of them on the one GPU at N = 1,
262 144 each at N = 8.  `--scaling weak` keeps 262 144 instances on every GPU instead (N = 8 is
the same job either way).  `--config config2|config3|config4` select the single-GPU
configurations with their BASELINE.json instance counts; `--instances M` overrides the count
(strong: of the whole job, weak: per GPU).

For N > 1 the driver launches this file under torch.distributed.run (one rank per GPU); ranks
only meet in the barriers around the timed region and in the MAX-reduction of the elapsed time.
`--sharded` instead drives all N GPUs from ONE process through the library's own multi-device
handle (fxb_create_sharded: one host thread + stream per device).

Output: ONE JSON line (rank 0) with the contract fields plus
  roofline      HBM roofline of the kernel (algorithmic bytes / HIP-event kernel time / 8 TB/s), and next to it
                roofline.valu: the vector-ALU issue roofline that actually bounds this path
  cpu_baseline  the unmodified reference (oracle/_ref) timed on this box's host cores (N = 1 only)
  parity        instances of the timed run compared bit for bit with the CPU oracle after the timed region
  extra         (N = 1, default workload only) the other single-GPU configurations, the 1/8 shard of configs[4]
                (what every GPU of an 8-GPU job runs) and the reference's own real-time measure (32-sample blocks
                against 666.667 us, tools/realtime_capacity.py)
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
SIMDS = 1024           # 256 CUs x 4 SIMD-32
MAX_CLOCK_HZ = 2.4e9   # MI355X_MICROARCH.md: max clock
FAST_VALU_CLOCKS = 2.12  # measured: SIMD clocks per wave64 v_mul_f32 / v_add_f32 with >= 2 wavefronts resident (tools/micro/valu_rate.hip)
NOMINAL_VALU_CLOCKS = 2.0  # MI355X_MICROARCH.md: a wave64 plain fp32 instruction occupies a SIMD-32 for two clocks (the nominal peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="config5", choices=["config2", "config3", "config4", "config5", "tram_bound", "config5_dane"])
    ap.add_argument("--samples", type=int, default=4096, help="sample periods per step (block length S; SURVEY 8d: 4096)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong (default): the config's instance count is the WHOLE job, split over the GPUs (configs[4]: 2 097 152 'sharded across 8'); "
                         "weak: every GPU runs the config's per-GPU share (config5: 262 144) whatever their number")
    ap.add_argument("--instances", type=int, default=0, help="instances (strong: of the whole job; weak: per GPU); 0 = BASELINE.json's count")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (rank 0, N=1 only); 0 disables")
    ap.add_argument("--parity-instances", type=int, default=1024, help="instances checked against the oracle after the timed region (0 disables)")
    ap.add_argument("--no-extras", action="store_true", help="skip the other configurations (they run by default for N=1, default workload)")
    ap.add_argument("--sharded", action="store_true", help="one process, --gpus devices through fxb_create_sharded")
    ap.add_argument("--cpu-baseline", default="reference", choices=["reference", "port"],
                    help="what is timed on the host cores: 'reference' = oracle/_ref/libfxref.so, the UNMODIFIED reference compiled by `make -C oracle ref` "
                         "(build() does that wherever /root/reference exists; the .so is git-ignored and travels to the GPU box) - the policy; "
                         "'port' = oracle/libfxoracle.so, this repo's C restatement (also what a tree without the prebuilt reference falls back to, and says so)")
    return ap.parse_args()


def plan_instances(config, scaling, instances, world, rank, progs, shard):
    """Which instances rank `rank` of `world` owns: (first instance, count, instances of the whole job).
    strong: BASELINE.json's total for the config (configs[4]: 2 097 152), contiguous balanced ranges (ragged totals: counts
    differ by one); weak: the config's per-GPU share on every rank.  Pure: the CPU tests check the totals for N = 1, 2, 4, 8."""
    if scaling == "strong":
        total = int(instances) or progs.CONFIG_TOTAL_INSTANCES[config]
        if total < world:
            raise SystemExit("fewer instances (%d) than GPUs (%d)" % (total, world))
        first, count = shard.shard_range(total, world, rank)
        return first, count, total
    per = int(instances) or progs.CONFIG_INSTANCES[config]
    first, count = shard.weak_shard(per, rank)
    return first, count, per * world


def device_stimulus(torch, n_inst, n_samples, first_instance, device):
    """fx8010_programs.stimulus() on the device (same counter-based hash, same bits), built in slabs of 256
    sample periods so that the int64 temporaries stay small next to the [S, N] fp32 result."""
    M = 0xFFFFFFFF
    out = torch.empty((n_samples, n_inst), dtype=torch.float32, device=device)
    n = (torch.arange(n_inst, dtype=torch.int64, device=device) + first_instance)[None, :] * 0x9E3779B1 + 0xF8010
    slab = max(8, min(256, (1 << 26) // max(n_inst, 1)))
    for s0 in range(0, n_samples, slab):
        s1 = min(n_samples, s0 + slab)
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


def measured_issue_quads(config, n_inst, n_samples, code_hash):
    """Quad-cycles in which a SIMD issued vector instructions for one wavefront during one sample, from a COMMITTED rocprofv3 PMC
    pass of exactly this workload AND this code (profiles/*pmc_valu*.json: SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2, tools/pmc_valu.txt,
    tools/pmc_summary.py) - a constant read from that file, like roofline.traffic; the timing it is set against is this run's.
    The file names the code object it was collected on (bench.code_hash = fxb_info xlate_code_hash of that run): a pass of
    other code - the generated code has changed since - is not used.  -> (quads or None, file or the reason there is none)"""
    import glob

    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_valu*.json")), reverse=True):
        try:
            d = json.load(open(f))
            w = d["bench"]
            if w["config"] == config and w["instances"] == n_inst and w["samples"] == n_samples:
                if w.get("code_hash") == code_hash:
                    return float(d["derived"]["valu_issue_quad_cycles_per_wave_sample"]), os.path.relpath(f, ROOT)
                stale = stale or "stale: %s is a pass of other code (%s, this run %s)" % (os.path.relpath(f, ROOT), w.get("code_hash"), code_hash)
        except (OSError, ValueError, KeyError):
            pass
    return None, stale


def measured_traffic(config, n_inst, n_samples):
    """HBM bytes per launch from a COMMITTED rocprofv3 PMC pass (profiles/*hbm_traffic*.json: FETCH_SIZE and WRITE_SIZE in
    their own passes, gfx950 read correction applied) of exactly this workload - a constant read from that file, not a
    measurement of this run (counters cannot be collected from inside the timed process); its name goes into the line."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic*.json")), reverse=True):
        try:
            d = json.load(open(f))
            w = d["workload"]
            if w["config"] == config and w["instances"] == n_inst and w["samples"] == n_samples:
                return float(d["hbm_bytes_per_launch"]), os.path.relpath(f, ROOT)
        except (OSError, ValueError, KeyError):
            pass
    return None, None


def cpu_baseline(text, budget_s, want="reference"):
    """Time the CPU side on the host cores: bounded sample.  Policy (--cpu-baseline): the unmodified reference, prebuilt as
    oracle/_ref/libfxref.so; a tree that does not have it (a fresh clone without /root/reference) times this repo's C
    restatement instead and says so in the line (kind "port", "fallback")."""
    import tempfile

    import fx8010_programs as progs
    from pyoracle import Oracle, Reference

    fallback = None
    if want == "reference" and not Reference.available():
        fallback = "oracle/_ref/libfxref.so is not in this tree (built by `make -C oracle ref` where /root/reference exists): the C restatement oracle/libfxoracle.so was timed"
        print("bench.py: " + fallback, file=sys.stderr)
    cls, kind = (Reference, "reference") if (want == "reference" and fallback is None) else (Oracle, "port")
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
        return {"value": round(mips, 1), "unit": "MIPS", "cores": cores, "kind": kind, "policy": "--cpu-baseline " + want, "fallback": fallback,
                "sample": "%d host threads x %d process() calls each of the same program (%.0f instr/sample), %.1f s" % (cores, samples, per_sample, secs),
                "single_thread_mips": round(i1 / s1 / 1e6, 1)}
    finally:
        os.unlink(path)


def parity_check(text, y_last, n_inst, first_instance, n_samples, blocks, want, option=0):
    """After the timed region: `want` instances spread over the shard (plus lanes 0 / 63 / the last one) are replayed on the
    CPU oracle - all `blocks` launches of the run, state carried from block to block as on the device - and the device's
    output of the LAST block is compared bit for bit.  y_last: callable instance -> numpy [S] of the device output."""
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    import fx8010_programs as progs
    from pyoracle import Oracle

    if want <= 0:
        return None
    picks = sorted(set([0, min(63, n_inst - 1), n_inst - 1] + [int(i) for i in np.linspace(0, n_inst - 1, num=min(want, n_inst))]))
    t0 = time.perf_counter()

    def one(n):
        o = Oracle(1)
        if option:
            o.set_option(option)
        if not o.load_text(text):
            return n, False
        x = progs.stimulus(1, n_samples, first_instance=first_instance + n)[:, 0].copy()
        ref = None
        for _ in range(blocks):
            ref = o.process_block(x)
        got = y_last(n)
        return n, bool(np.array_equal(ref.view(np.uint32), np.ascontiguousarray(got).view(np.uint32)))

    with ThreadPoolExecutor(max_workers=usable_cores()) as pool:  # the oracle calls release the GIL
        results = list(pool.map(one, picks))
    bad = [n for n, ok in results if not ok]
    return {"parity_checked": len(picks), "parity_ok": not bad, "mismatching_instances": bad[:8],
            "what": "device output of the last timed block vs oracle/ (C restatement of the reference), bit for bit, after replaying all %d blocks" % blocks,
            "seconds": round(time.perf_counter() - t0, 1)}


def kernel_name(batch):
    kid = batch.info("kernel")
    vg = (0, 0, 64, 72, 80, 96, 128, 168, 256)
    if kid == 0:
        return "fx_step_block (HIP C++)"
    if kid == 1:
        return "fx_interp_lds (gfx950 asm interpreter, LDS register file)"
    if kid <= 8:
        return "fx_interp_v%d (gfx950 asm interpreter, VGPR register file)" % vg[kid]
    turns = "; the wavefronts of a SIMD take turns at the top priority" if "by turns" in batch.tier_note() else ""
    return ("fx_xlate_v%d (program translated to gfx950 code: %d records inline, %d handler calls, %d saturations elided, %d code bytes%s)"
            % (vg[kid - 7], batch.info("xlate_inlined"), batch.info("xlate_called"), batch.info("xlate_unsaturated"), batch.info("xlate_code_bytes"), turns))


class ClockSampler:
    """Shader clock and socket power of one GPU while the timed region runs, from the amdgpu hwmon files (None when the
    files cannot be found: the numbers are extra information, never part of `value`)."""

    def __init__(self, torch, device_index):
        import glob
        import threading
        self.freq = self.power = None
        self.samples = []
        self._stop = threading.Event()
        self._thread = None
        try:
            p = torch.cuda.get_device_properties(device_index)
            bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
            for d in glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf):
                f, w = os.path.join(d, "freq1_input"), os.path.join(d, "power1_input")
                if os.path.exists(f):
                    self.freq, self.power = f, (w if os.path.exists(w) else None)
        except Exception:
            pass

    def _read(self, path):
        try:
            with open(path) as fh:
                return float(fh.read().strip())
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            self.samples.append((self._read(self.freq), self._read(self.power) if self.power else None))
            self._stop.wait(0.005)

    def start(self):
        import threading
        if self.freq is None:
            return
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def stop(self):
        """-> (mean shader clock in MHz, mean power in W) over the samples taken, either None when unknown"""
        if self._thread is None:
            return None, None
        self._stop.set()
        self._thread.join()
        # (the first readings still show the clock of the idle chip ramping up: the first fifth of a short run's samples is left out)
        skip = max(1, len(self.samples) // 5) if len(self.samples) >= 3 else 0
        samples = self.samples[skip:]
        f = [a for a, _ in samples if a]
        w = [b for _, b in samples if b]
        return (round(sum(f) / len(f) / 1e6, 1) if f else None), (round(sum(w) / len(w) / 1e6, 1) if w else None)


def run_workload(torch, fx8010_amd, progs, shard, config, n_local, first_instance, scaling, S, steps, warmup, devices, rank, dist, reduce_dev, parity_n):
    """One configuration on this process's device(s): instances [first_instance, first_instance + n_local) of the job
    (--sharded: split once more over this process's devices by the library).  Returns (fields for the JSON line, batch text)."""
    text = progs.CONFIGS[config]()
    P = progs.count_instructions(text)
    sharded = len(devices) > 1
    batch = fx8010_amd.Batch(n_local, 1, devices=devices) if sharded else fx8010_amd.Batch(n_local, 1, devices[0])
    option = getattr(progs, "CONFIG_OPTIONS", {}).get(config, 0)  # opt-in behaviour beyond the reference (config5_dane)
    if option:
        batch.set_option(option)
    if not batch.load_text(text):
        raise RuntimeError("program failed to load: %s" % batch.errors())
    n_inst = batch.shards()[0][2]   # instances on this rank's (first) GPU: what its kernel launch processes
    xs, ys, streams = [], [], []
    for k, (dev_ord, first, count) in enumerate(batch.shards()):
        dev = torch.device("cuda", dev_ord)
        with torch.cuda.device(dev):
            xs.append(device_stimulus(torch, count, S, first_instance + first, dev))  # [S, N_k] resident in HBM
            ys.append(torch.empty_like(xs[-1]))
    for d in devices:
        torch.cuda.synchronize(d)
    # a non-default torch stream: the kernel is launched on it through the C ABI, so the
    # torch.cuda.Events below (HIP events on that same stream) bracket exactly the launches
    dev0 = torch.device("cuda", devices[0])
    tstream = torch.cuda.Stream(device=dev0)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    def step():
        if sharded:
            batch.process_block_dev_shards([t.data_ptr() for t in xs], [t.data_ptr() for t in ys], S)
        else:
            batch.process_block_dev(xs[0].data_ptr(), ys[0].data_ptr(), S, stream)

    def barrier():
        if sharded:
            batch.sync()
        if dist is not None:
            dist.barrier()
        for d in devices:
            torch.cuda.synchronize(d)

    for _ in range(warmup):
        step()
    barrier()
    c0 = batch.instruction_counter()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    sampler = ClockSampler(torch, devices[0])
    sampler.start()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        step()
    ev1.record()
    barrier()
    t1 = time.perf_counter()
    clock_mhz, power_w = sampler.stop()
    elapsed = t1 - t0
    # HIP events on the launch stream (single device); a sharded batch launches from its own threads on its own
    # streams: the slowest shard's last launch, measured by the library's events on that stream
    kernel_ms = batch.last_kernel_ms() if sharded else ev0.elapsed_time(ev1) / max(steps, 1)
    last_ms = batch.last_kernel_ms()
    executed = batch.instruction_counter() - c0  # reference counting: END/SKIP count, skipped don't
    ood = batch.ood_flags()
    # every device's own time: one entry per rank (torchrun) or per shard (--sharded)
    per_gpu_kernel_ms = batch.shard_kernel_ms() if sharded else shard.gather_scalars(dist, kernel_ms, reduce_dev)
    per_gpu_elapsed_s = shard.gather_scalars(dist, elapsed, reduce_dev)
    elapsed = shard.reduce_scalar(dist, elapsed, "max", reduce_dev)        # slowest rank
    executed_all = shard.reduce_scalar(dist, executed, "sum", reduce_dev)  # whole job
    n_job = int(round(shard.reduce_scalar(dist, n_local, "sum", reduce_dev)))
    per_gpu_instances = [c for _, _, c in batch.shards()] if sharded else [int(round(v)) for v in shard.gather_scalars(dist, n_local, reduce_dev)]

    parity = None
    if rank == 0 and parity_n > 0:
        cuts = [(f, c) for _, f, c in batch.shards()]

        def y_last(n):
            for k, (f, c) in enumerate(cuts):
                if f <= n < f + c:
                    return ys[k][:, n - f].cpu().numpy()
            raise IndexError(n)

        parity = parity_check(text, y_last, n_local, first_instance, S, warmup + steps, parity_n, option)

    res = None
    if rank == 0:
        n_dev = len(devices)
        mips = executed_all / elapsed / 1e6
        tram_ops = batch.info("tram_ops")
        rows = batch.info("num_rows")
        # algorithmic HBM bytes of ONE launch on ONE GPU: PCM in+out, every executed
        # TRAM read/write, and the once-per-block register-file spill/fill
        bytes_per_inst_sample = 4 * (1 + 1) + 4 * tram_ops
        algo_bytes = float(n_inst) * (S * bytes_per_inst_sample + 2 * 4 * (rows + 9))
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_source = measured_traffic(config, n_inst, S)
        valu_per_wave_sample = batch.info("xlate_valu")   # (a program cut into stages: the sum over its stages = one sample of one group of 64 instances)
        stages = batch.info("waves_per_wg")
        waves = (n_inst + 63) // 64
        valu = None
        if valu_per_wave_sample > 0:
            per_s = valu_per_wave_sample * waves * S / (kernel_ms * 1e-3)
            peak = SIMDS * MAX_CLOCK_HZ / FAST_VALU_CLOCKS
            peak_nominal = SIMDS * MAX_CLOCK_HZ / NOMINAL_VALU_CLOCKS
            code_hash = "%016x" % batch.info("xlate_code_hash")
            # issue time of the instructions a wave executes per sample, by class (cost table: fx_xlate.cpp Emitter::issueCost,
            # measured by tools/micro/mix_cost.hip), against the SIMD cycles that were available at the clock the chip held
            clocks = batch.info("xlate_valu_clocks")
            hz = (clock_mhz * 1e6) if clock_mhz else MAX_CLOCK_HZ
            busy = clocks * (waves / float(SIMDS)) * S / (kernel_ms * 1e-3 * hz)
            # ... and the same from hardware counters: quad-cycles in which the SIMD issued for a wavefront (dual issue counted once)
            quads, quads_source = measured_issue_quads(config, n_inst, S, code_hash)
            busy_counters = round(quads * 4.0 * (waves * stages / float(SIMDS)) * S / (kernel_ms * 1e-3 * hz), 4) if quads else None
            valu = {"bound": "valu issue", "achieved": round(per_s / 1e9, 2), "peak": round(peak / 1e9, 1), "unit": "G wave-instr/s",
                    "frac": round(per_s / peak, 4), "peak_nominal_2clk": round(peak_nominal / 1e9, 1), "frac_of_nominal_2clk_peak": round(per_s / peak_nominal, 4),
                    "code_hash": code_hash, "valu_per_wave_sample": valu_per_wave_sample, "waves_per_simd": round(waves * stages / float(SIMDS), 3),
                    "stages": stages,
                    "valu_per_emulated_instr": round(valu_per_wave_sample / max(executed / float(steps * S * n_local), 1e-9), 3),
                    "valu_4clock_class_per_wave_sample": batch.info("xlate_valu_slow"),
                    "issue_clocks_per_wave_sample": clocks, "clock_mhz": clock_mhz, "power_w": power_w,
                    "clocks_per_valu_per_simd": round(kernel_ms * 1e-3 * hz / (valu_per_wave_sample * max(waves / float(SIMDS), 1.0) * S), 3),
                    "simd_issue_busy_from_counters": busy_counters, "issue_quad_cycles_source": quads_source,
                    "simd_issue_busy_model": round(min(busy, 1.0), 4), "simd_issue_busy_model_uncapped": round(busy, 4),
                    "note": "peak = 1024 SIMDs x 2.4 GHz / 2.12 clocks, the measured rate of plain fp32 add / mul (peak_nominal_2clk: the data-sheet 2 clocks); "
                            "conversions, fp64, min/max and compares cost 2.6-4.25 clocks each.  simd_issue_busy_from_counters = quad-cycles in which a SIMD "
                            "issued vector instructions (SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2 of a committed PMC pass of this launch shape AND this code "
                            "object: code_hash; null when the committed pass is of other code) x 4 clocks / the SIMD clocks available at the measured clock; "
                            "simd_issue_busy_model = modelled issue clocks of the executed mix / the same SIMD clocks (a four-wavefront calibration: "
                            "capped at 1, the uncapped figure beside it; DESIGN.md section 5); below 2 wavefronts per SIMD a wavefront issues only every ~5th clock"}
        res = {
            "value": round(mips, 1),
            "ms_per_step": round(elapsed / max(steps, 1) * 1e3, 4),
            "config": {
                "workload": "%s: %d instances in all (%s scaling: %s on this GPU) x %d-instr program, block of %d samples, mono 48 kHz"
                            % (config, n_job, scaling, ("all of them" if n_inst == n_job else "%d" % n_inst), P, S),
                "name": config, "instances_per_gpu": n_inst, "samples_per_step": S,
                "instances_total": n_job, "per_gpu_instances": per_gpu_instances,
                "instr_per_sample_static": P,
                "instr_per_sample_executed": round(executed / float(steps * S * n_local), 3),
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
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": kernel_name(batch),
                "kernel_ms": round(kernel_ms, 4),
                "kernel_ms_last_launch": round(last_ms, 4),
                "per_gpu_kernel_ms": [round(v, 4) for v in per_gpu_kernel_ms],
                "per_gpu_kernel_ms_note": ("last launch of each shard (library events on the shard's stream)" if sharded else "mean of the timed launches on each rank's device (HIP events on the launch stream)"),
                "per_gpu_elapsed_s": [round(v, 6) for v in per_gpu_elapsed_s],
                "algorithmic_bytes_per_launch": algo_bytes,
                "note": "the interpreter is instruction-issue bound (>= 12 emulated instr per algorithmic HBM byte): see roofline.valu and DESIGN.md section 5",
                "emulated_instr_per_s_per_gpu": round(executed / n_dev / (kernel_ms * 1e-3 * steps), 1),
                "algorithmic_bytes_note": "instances on this GPU x (samples x (4 B PCM in + 4 B PCM out + 4 B per executed delay-line read / write) + 2 x 4 B x (register rows + 9) once per block)",
                "valu": valu,
            },
        }
        if parity is not None:
            res["parity"] = parity
    torch.cuda.set_stream(torch.cuda.default_stream(dev0))
    del batch, xs, ys
    torch.cuda.empty_cache()
    return res, text


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
    n_visible = max(torch.cuda.device_count(), 1)
    if rehearsal or local >= n_visible:
        # (a launcher that shows every rank only its own GPU - ROCR_VISIBLE_DEVICES per rank - makes that GPU ordinal 0)
        local = local % n_visible
    if args.sharded:
        if world != 1:
            raise SystemExit("--sharded drives all devices from one process: do not launch it under torch.distributed.run")
        devices = [d % n_visible for d in range(args.gpus)] if rehearsal else list(range(args.gpus))
        dist, reduce_dev = None, "cpu"
        torch.cuda.set_device(devices[0])
    else:
        torch.cuda.set_device(local)
        devices = [local]
        dev = torch.device("cuda", local)
        dist = shard.init_process_group("gloo" if rehearsal else "nccl", dev)  # RCCL; None for a single process
        reduce_dev = "cpu" if rehearsal else dev

    n_gpus = args.gpus if args.sharded else world
    S = args.samples
    if args.sharded:   # this process owns the whole job; the library splits it over its devices (fx_shard.cpp Sharded::plan)
        first = 0
        n_local = (args.instances or progs.CONFIG_TOTAL_INSTANCES[args.config]) if args.scaling == "strong" else (args.instances or progs.CONFIG_INSTANCES[args.config]) * n_gpus
    else:
        first, n_local, _ = plan_instances(args.config, args.scaling, args.instances, world, rank, progs, shard)
    res, text = run_workload(torch, fx8010_amd, progs, shard, args.config, n_local, first, args.scaling, S, args.steps, args.warmup, devices, rank, dist,
                             reduce_dev, args.parity_instances)

    if rank == 0:
        out = {
            "metric": "aggregate emulated DSP MIPS (instr x samples x instances / s)",
            "value": res["value"],
            "unit": "MIPS",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32+f64",  # the path computes in IEEE fp32 with fp64 for INTERP / LOG / EXP, as the reference does
            "data": "synthetic",
            "config": res["config"],
            "roofline": res["roofline"],
        }
        if args.sharded:
            out["config"]["host"] = "one process, fxb_create_sharded over %d devices (one host thread + stream each)" % len(devices)
        if "parity" in res:
            out["parity"] = res["parity"]
        single = n_gpus == 1
        if single and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(text, args.cpu_seconds, args.cpu_baseline)
        if single and not args.no_extras and args.config == "config5" and args.scaling == "strong" and not args.instances and S == 4096:
            # the other single-GPU configurations of BASELINE.json at their instance counts, and the 1/8 shard of configs[4]
            # (262 144 instances: what every GPU of the 8-GPU job runs - the N = 8 point of the strong-scaling curve is 8 x this);
            # fewer launches each, same block length
            extra = {}
            # (enough launches for the clock sampler to see the chip at its working clock: config2's launch takes 0.6 ms)
            plan = [("config2", 4096, 400, 256), ("config3", 65536, 40, 256), ("config4", 262144, 8, 256), ("config5_shard_1of8", 262144, 20, 256)]
            for name, n, k, pn in plan:
                cfg = "config5" if name.startswith("config5") else name
                try:
                    r, _ = run_workload(torch, fx8010_amd, progs, shard, cfg, n, 0, "weak" if name.startswith("config5") else "strong", S, k, 1, devices, 0, None, "cpu", pn)
                    extra[name] = {"value": r["value"], "unit": "MIPS", "steps": k, "ms_per_step": r["ms_per_step"], "workload": r["config"]["workload"],
                                   "instr_per_sample_executed": r["config"]["instr_per_sample_executed"],
                                   "kernel_ms": r["roofline"]["kernel_ms"], "hbm_frac": r["roofline"]["frac"],
                                   "valu_frac": (r["roofline"]["valu"] or {}).get("frac"), "waves_per_simd": (r["roofline"]["valu"] or {}).get("waves_per_simd"),
                                   "stages": (r["roofline"]["valu"] or {}).get("stages"),
                                   "valu_per_emulated_instr": (r["roofline"]["valu"] or {}).get("valu_per_emulated_instr"),
                                   "simd_issue_busy_from_counters": (r["roofline"]["valu"] or {}).get("simd_issue_busy_from_counters"),
                                   "issue_quad_cycles_source": (r["roofline"]["valu"] or {}).get("issue_quad_cycles_source"),
                                   "simd_issue_busy_model": (r["roofline"]["valu"] or {}).get("simd_issue_busy_model"),
                                   "code_hash": (r["roofline"]["valu"] or {}).get("code_hash"), "clock_mhz": (r["roofline"]["valu"] or {}).get("clock_mhz"),
                                   "power_w": (r["roofline"]["valu"] or {}).get("power_w"),
                                   "parity_checked": (r.get("parity") or {}).get("parity_checked"), "parity_ok": (r.get("parity") or {}).get("parity_ok")}
                except Exception as e:  # an extra must never take the headline line down
                    extra[name] = {"error": str(e)[:200]}
            # the reference's own measure of speed (source/main.cpp:99-155): 32-sample blocks at 48 kHz against 666.667 us, host-fed
            # from pinned buffers with a slider moving every 8th block - how many instances stay inside the budget at p99.9
            # (tools/realtime_capacity.py; the full sweep with 5 000 blocks per point is profiles/r05_realtime.json)
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import realtime_capacity as rt
                rows = []
                for mode, counts in (("host", (131072, 147456, 163840, 180224)), ("device", (393216, 458752, 524288))):
                    for n in counts:
                        r = rt.measure(torch, fx8010_amd, progs, n, 4000, 400, mode)
                        rows.append({k: r[k] for k in ("instances", "mode", "median_us", "p99_us", "p999_us", "max_us", "kernel_us_median", "within_budget_p999",
                                                        "pcie_GBps_each_way_at_median", "translations_in_timed_region", "parity_ok", "blocks")})
                extra["realtime"] = {"budget_us": round(rt.BUDGET_US, 3), "block_samples": rt.BLOCK, "program": "config5", "control_moved_every_blocks": rt.SLIDER_EVERY,
                                     "capacity_host_fed": rt.capacity([r for r in rows if r["mode"] == "host"]),
                                     "capacity_device_resident": rt.capacity([r for r in rows if r["mode"] == "device"]),
                                     "note": "largest N of the rows whose p99.9 block time (call -> output in host memory / -> fxb_sync) is within the budget", "rows": rows}
            except Exception as e:
                extra["realtime"] = {"error": str(e)[:200]}
            out["extra"] = extra
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
