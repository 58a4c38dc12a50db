#!/bin/bash
# A/B of the LDS layout of the LOG / EXP tables (fx_xlate_emit.hpp LutLdsLayout) on config4, inside ONE call on the
# GPU box: narrow = three ds_read_b64 per LOG / EXP (x1, slope, y1), wide = a ds_read_b64 (x1) + a ds_read_b128 ({slope, y1}).
# The switch (FX_XLATE_LUTWIDE) exists in the DIAGNOSTICS build only; both variants compute the same words (parity checked in
# every run).  Alternating runs, then the SQ counter pass of each.      tools/lut_wide_ab.sh > gpurun_out/r05_lut_wide_ab.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
make -s -j8 -C $ROOT/fx8010-emulator-core_amd/csrc diag || exit 1
export FX8010_AMD_LIB=$ROOT/fx8010-emulator-core_amd/csrc/build/diag/libfx8010_amd.so
OUT=$ROOT/gpurun_out/lut_wide_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for round in 1 2 3; do
  for w in 0 1; do
    FX_XLATE_LUTWIDE=$w python3 $ROOT/bench.py --config config4 --steps 10 --warmup 2 --no-extras --cpu-seconds 0 --parity-instances 64 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); v = d['roofline']['valu']
print('round $round wide=$w: %.4e instr/s  kernel %.3f ms  %s MHz  %s W  valu/wave-sample %d  parity %s' % (d['value'] * 1e6, d['roofline']['kernel_ms'], v['clock_mhz'], v['power_w'], v['valu_per_wave_sample'], d['parity']['parity_ok']))"
  done
done
for w in 0 1; do
  FX_XLATE_LUTWIDE=$w python3 $ROOT/bench.py --config config4 --steps 10 --warmup 2 --no-extras --cpu-seconds 0 --parity-instances 0 > $OUT/bench_$w.json 2>/dev/null
  FX_XLATE_LUTWIDE=$w rocprofv3 -i $ROOT/tools/pmc_sq.txt -d $OUT/pmc_$w -o w$w --output-format csv -- python3 $ROOT/bench.py --config config4 --cpu-seconds 0 --steps 2 --warmup 1 --no-extras --parity-instances 0 > /dev/null 2> $OUT/pmc_$w.err
  python3 $ROOT/tools/pmc_summary.py $OUT/pmc_$w --bench $OUT/bench_$w.json > $OUT/pmc_sq_wide$w.json
  python3 - $OUT/pmc_sq_wide$w.json $w <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); p = d["per_wave_sample"]; r = d["derived"]
print("wide=%s counters per wave-sample: LDS instructions %.1f  bank-conflict cycles %.1f  LDS active %.1f  wait-any %.1f of %.1f wave cycles (%.1f %%)  wait-LDS share %.4f  VALU %.1f" % (
    sys.argv[2], p["SQ_INSTS_LDS"], p["SQ_LDS_BANK_CONFLICT"], p.get("SQ_ACTIVE_INST_LDS", 0), p["SQ_WAIT_ANY"], p["SQ_WAVE_CYCLES"], 100 * r["sq_wait_any_share_of_wave_cycles"],
    r.get("sq_wait_inst_lds_share_of_wave_cycles", 0), p["SQ_INSTS_VALU"]))
PY
  rm -rf $OUT/pmc_$w
done
