#!/usr/bin/env python3
"""Summarise rocprofv3 counter passes (tools/pmc_sq.txt) of a bench.py run for the translated / interpreter kernel.

    python tools/pmc_summary.py <dir with pmc_*/..._counter_collection.csv> <samples per launch> [kernel prefix [kernel ms, shader MHz]]
    python tools/pmc_summary.py <dir> --bench <bench.py's JSON line of the same workload, unprofiled> [kernel prefix]

With --bench the samples per launch, the kernel time, the shader clock and - what ties the counters to the code they were
collected on - the code object's fingerprint (roofline.valu.code_hash) are taken from that line and recorded under "bench":
bench.py only uses a committed pass whose fingerprint is the running code's.

Prints one JSON object: counters averaged over the kernel's dispatches, per wavefront and per wave-sample,
plus the shares DESIGN.md section 5 quotes (waiting / issuing as a fraction of a wave's resident time).  Only ratios of SQ
counters are derived here: absolute clocks per instruction come from bench.py, which times the launches and reads the
shader clock while they run (SQ_WAVE_CYCLES x 4 agrees with that for one wave per SIMD - config3 - and reads ~30 % low with
four, so it is not used as a clock).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    bench = None
    if sys.argv[2] == "--bench":
        line = [l for l in open(sys.argv[3]).read().split("\n") if l.startswith("{")][-1]
        b = json.loads(line)
        v = b["roofline"]["valu"] or {}
        bench = {"config": b["config"]["name"], "instances": b["config"]["instances_per_gpu"], "samples": b["config"]["samples_per_step"],
                 "code_hash": v.get("code_hash"), "kernel_ms": b["roofline"]["kernel_ms"], "clock_mhz": v.get("clock_mhz"), "stages": v.get("stages"),
                 "value_mips": b["value"], "file": os.path.basename(sys.argv[3])}
        samples = bench["samples"]
        prefix = sys.argv[4] if len(sys.argv) > 4 else "fx_"
        sys.argv = sys.argv[:2] + [str(samples), prefix] + ([str(bench["kernel_ms"]), str(bench["clock_mhz"])] if bench["clock_mhz"] else [])
    root, samples = sys.argv[1], int(sys.argv[2])
    prefix = sys.argv[3] if len(sys.argv) > 3 else "fx_"
    sums, counts, meta = defaultdict(float), defaultdict(int), {}
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if not k.startswith(prefix) or "reduce" in k or "fill" in k:
                    continue
                sums[row["Counter_Name"]] += float(row["Counter_Value"])
                counts[row["Counter_Name"]] += 1
                meta = {"kernel": k, "grid": int(row["Grid_Size"]), "workgroup": int(row["Workgroup_Size"]), "vgprs": int(row["VGPR_Count"]),
                        "lds_bytes": int(row["LDS_Block_Size"])}
    if not sums:
        print(json.dumps({"error": "no dispatches of %s* under %s" % (prefix, root)}))
        return
    avg = {k: sums[k] / counts[k] for k in sums}
    # wavefronts per launch: the grid's (one launch = grid / 64 wavefronts, exactly).  SQ_WAVES agrees most of the time; in some
    # passes one dispatch's window also counts up to one chip-full (4096) of a neighbouring dispatch's wavefronts while every
    # other counter of it is exact (seen in round 5: 36 864 for 32 768) - it is reported, not used.
    waves = meta["grid"] / 64.0
    per_ws = {k: v / waves / samples for k, v in avg.items() if k != "SQ_WAVES"}
    out = {"workload": meta, "samples_per_launch": samples, "dispatches_averaged": min(counts.values()), "waves": waves,
           "sq_waves_counter_mean": avg.get("SQ_WAVES"),
           "per_wave_sample": {k: round(v, 2) for k, v in sorted(per_ws.items())}}
    d = {}
    wc = per_ws.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA"):
            if k in per_ws:
                d[k.lower() + "_share_of_wave_cycles"] = round(per_ws[k] / wc, 4)
        d["waves_per_simd"] = round(waves / 1024.0, 3)
    # VALU issue from counters (tools/pmc_valu.txt): a SIMD issues vector instructions in quad-cycles, one - or, for the plain
    # fp32 / integer class from two different wavefronts, TWO (SQ_ACTIVE_INST_VALU2 counts those quad-cycles).  Quad-cycles in
    # which a wavefront's SIMD issued for it = instructions - dual-issued quad-cycles; x 4 clocks x wavefronts per SIMD against the
    # clocks of a sample period (bench.py: kernel time x shader clock) = the share of the SIMD's issue slots that were taken.
    if "SQ_ACTIVE_INST_VALU2" in per_ws and "SQ_INSTS_VALU" in per_ws:
        quads = per_ws["SQ_INSTS_VALU"] - per_ws["SQ_ACTIVE_INST_VALU2"]
        d["valu_issue_quad_cycles_per_wave_sample"] = round(quads, 2)
        d["valu_dual_issued_share_of_instructions"] = round(2.0 * per_ws["SQ_ACTIVE_INST_VALU2"] / per_ws["SQ_INSTS_VALU"], 4)
        if len(sys.argv) > 4:   # kernel ms per launch and shader clock in MHz of the run (from its bench line)
            ms, mhz = float(sys.argv[4]), float(sys.argv[5])
            clocks_per_sample = ms * 1e-3 * mhz * 1e6 / samples
            d["clocks_per_sample"] = round(clocks_per_sample, 1)
            d["valu_issue_busy_from_counters"] = round(quads * 4.0 * max(waves / 1024.0, 0.0) / clocks_per_sample, 4)
    out["derived"] = d
    if bench:
        out["bench"] = bench
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
