#!/bin/bash
# Does the non-temporal hint on the PCM / delay-line accesses change what the chip draws (and so the clock it holds) at config5?
# DIAGNOSTICS build, FX_XLATE_NT=mask (1 TRAM loads, 2 TRAM stores, 4 PCM loads, 8 PCM stores; the release default for the shard: 3),
# alternating runs inside one call.     tools/nt_clock_ab.sh > gpurun_out/r05_nt_clock_ab.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
make -s -j8 -C $ROOT/fx8010-emulator-core_amd/csrc diag || exit 1
export FX8010_AMD_LIB=$ROOT/fx8010-emulator-core_amd/csrc/build/diag/libfx8010_amd.so
cd $ROOT
for round in 1 2; do
  for m in 3 0 15 12; do
    FX_XLATE_NT=$m python3 bench.py --config config5 --scaling weak --steps 40 --warmup 3 --no-extras --cpu-seconds 0 --parity-instances 16 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); v = d['roofline']['valu']
print('round $round NT=%-2s: %.4e instr/s  kernel %.3f ms  %s MHz  %s W  parity %s' % ('$m', d['value'] * 1e6, d['roofline']['kernel_ms'], v['clock_mhz'], v['power_w'], d['parity']['parity_ok']))"
  done
done
