#!/usr/bin/env python3
"""Summarise rocprofv3 counter passes (tools/pmc_sq.txt) of a bench.py run for the translated / interpreter kernel.

    python tools/pmc_summary.py <dir with pmc_*/..._counter_collection.csv> <samples per launch> [kernel prefix]

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
    waves = avg.get("SQ_WAVES") or meta["grid"] / 64.0
    per_ws = {k: v / waves / samples for k, v in avg.items() if k != "SQ_WAVES"}
    out = {"workload": meta, "samples_per_launch": samples, "dispatches_averaged": min(counts.values()), "waves": waves,
           "per_wave_sample": {k: round(v, 2) for k, v in sorted(per_ws.items())}}
    d = {}
    wc = per_ws.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA"):
            if k in per_ws:
                d[k.lower() + "_share_of_wave_cycles"] = round(per_ws[k] / wc, 4)
        d["waves_per_simd"] = round(waves / 1024.0, 3)
    out["derived"] = d
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
