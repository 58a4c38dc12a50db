#!/bin/bash
# Where does config4's time go?  Pieces of the LOG / EXP code taken out one at a time (FX_XLATE_LUTPROBE_WRONG_RESULTS: the results are WRONG,
# only the timing means something): 1 = no branch to the miss path, 4 = LDS reads issued but not waited for, 2 = no LDS reads at
# all.  One call on the GPU box: tools/lut_cost_probe.sh > gpurun_out/lut_cost_probe.txt
# (the knobs used here exist only in the DIAGNOSTICS build of the library: fx_knobs.hpp)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
make -s -j8 -C $ROOT/fx8010-emulator-core_amd/csrc diag || exit 1
export FX8010_AMD_LIB=$ROOT/fx8010-emulator-core_amd/csrc/build/diag/libfx8010_amd.so
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() {
  local label=$1; shift
  local v
  v=$(env "$@" python3 bench.py --config config4 --steps 5 --warmup 1 --no-extras --cpu-seconds 0 --parity-instances 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['valu']['valu_per_wave_sample'])")
  echo "$label: MIPS kernel_ms valu/wave-sample = $v"
}
run "baseline                     " FX_X=0
run "no miss branch               " FX_XLATE_LUTPROBE_WRONG_RESULTS=1
run "reads, no wait               " FX_XLATE_LUTPROBE_WRONG_RESULTS=4
run "reads, no wait, no branch    " FX_XLATE_LUTPROBE_WRONG_RESULTS=5
run "no reads, no wait            " FX_XLATE_LUTPROBE_WRONG_RESULTS=2
run "no reads, no wait, no branch " FX_XLATE_LUTPROBE_WRONG_RESULTS=3
run "baseline again               " FX_X=0
