#!/bin/bash
# How much independent work fits into a LOG / EXP's LDS round trip for nothing?  config4 (BASELINE configs[3]) with every LOG / EXP
# padded with n instructions between its reads and their wait (FX_XLATE_LUTPAD / _SLOW) or behind the wait (.._AFTER=1: the
# price of the same instructions when nothing hides them), and with raised wave priority around the reads (FX_XLATE_LUTPRIO).
#   tools/lut_pad_probe.sh > gpurun_out/lut_pad_probe.txt        (on the GPU box, one call: boxes differ by a few per cent)
# (the knobs used here exist only in the DIAGNOSTICS build of the library: fx_knobs.hpp)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
make -s -j8 -C $ROOT/fx8010-emulator-core_amd/csrc diag || exit 1
export FX8010_AMD_LIB=$ROOT/fx8010-emulator-core_amd/csrc/build/diag/libfx8010_amd.so
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() {
  local label=$1; shift
  local v
  v=$(env "$@" python3 bench.py --config config4 --steps 5 --warmup 1 --no-extras --cpu-seconds 0 --parity-instances 64 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['valu']['valu_per_wave_sample'], d.get('parity',{}).get('parity_ok'))")
  echo "$label: MIPS kernel_ms valu/wave-sample parity = $v"
}
run "baseline            " FX_X=0
run "pad 4 plain, hidden " FX_XLATE_LUTPAD=4
run "pad 4 plain, after  " FX_XLATE_LUTPAD=4 FX_XLATE_LUTPAD_AFTER=1
run "pad 8 plain, hidden " FX_XLATE_LUTPAD=8
run "pad 8 plain, after  " FX_XLATE_LUTPAD=8 FX_XLATE_LUTPAD_AFTER=1
run "pad 16 plain, hidden" FX_XLATE_LUTPAD=16
run "pad 16 plain, after " FX_XLATE_LUTPAD=16 FX_XLATE_LUTPAD_AFTER=1
run "pad 2 slow, hidden  " FX_XLATE_LUTPAD_SLOW=2
run "pad 2 slow, after   " FX_XLATE_LUTPAD_SLOW=2 FX_XLATE_LUTPAD_AFTER=1
run "pad 4 slow, hidden  " FX_XLATE_LUTPAD_SLOW=4
run "pad 4 slow, after   " FX_XLATE_LUTPAD_SLOW=4 FX_XLATE_LUTPAD_AFTER=1
run "setprio 1           " FX_XLATE_LUTPRIO=1
run "setprio 3           " FX_XLATE_LUTPRIO=3
run "baseline again      " FX_X=0
