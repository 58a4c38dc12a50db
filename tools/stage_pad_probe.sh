#!/bin/bash
# Is a stage's sample loop bound by issue slots or by the latency of its dependent chain?  config2 at 4 096 instances in eight
# stages (and unstaged, and config3: one wavefront per SIMD) with n INDEPENDENT instructions added to every sample of every
# stage (FX_XLATE_LOOPPAD plain class, .._SLOW 4-clock class): issue-bound code pays for each of them, latency-bound code does not.
#   tools/stage_pad_probe.sh > gpurun_out/stage_pad_probe.txt      (on the GPU box, one call)
# (the knobs used here exist only in the DIAGNOSTICS build of the library: fx_knobs.hpp)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
make -s -j8 -C $ROOT/fx8010-emulator-core_amd/csrc diag || exit 1
export FX8010_AMD_LIB=$ROOT/fx8010-emulator-core_amd/csrc/build/diag/libfx8010_amd.so
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() {
  local label=$1 cfg=$2; shift 2
  local v
  v=$(env "$@" python3 bench.py --config $cfg --steps 20 --warmup 2 --no-extras --cpu-seconds 0 --parity-instances 16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['valu']['valu_per_wave_sample'], d['roofline']['valu']['stages'], d.get('parity',{}).get('parity_ok'))")
  echo "$label: MIPS kernel_ms valu/group-sample stages parity = $v"
}
for cfg in config2 config3; do
  run "$cfg baseline      " $cfg FX_STAGES_TUNE=0
  run "$cfg +4 plain      " $cfg FX_STAGES_TUNE=0 FX_XLATE_LOOPPAD=4
  run "$cfg +8 plain      " $cfg FX_STAGES_TUNE=0 FX_XLATE_LOOPPAD=8
  run "$cfg +16 plain     " $cfg FX_STAGES_TUNE=0 FX_XLATE_LOOPPAD=16
  run "$cfg +4 slow       " $cfg FX_STAGES_TUNE=0 FX_XLATE_LOOPPAD_SLOW=4
  run "$cfg +8 slow       " $cfg FX_STAGES_TUNE=0 FX_XLATE_LOOPPAD_SLOW=8
done
run "config2 unstaged        " config2 FX_STAGES=1
run "config2 unstaged +16 pl." config2 FX_STAGES=1 FX_XLATE_LOOPPAD=16
run "config2 unstaged +8 slow" config2 FX_STAGES=1 FX_XLATE_LOOPPAD_SLOW=8
