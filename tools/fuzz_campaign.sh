#!/bin/bash
# The differential fuzzers back to back on the GPU box (gpurun): one summary line per run under gpurun_out/<tag>/campaign.txt.
#   tools/fuzz_campaign.sh <tag> [scale] [part/parts]     scale multiplies the program counts (default 1); "1/2" runs every second
#   line starting with the first, "2/2" the others (one gpurun call has 20 minutes); "L:3,9,37" runs those lines of the list below
TAG=${1:-campaign}
K=${2:-1}
PART=${3:-1/1}
PART_I=${PART%%/*}; PART_N=${PART##*/}; LINE=0
case "$PART" in L:*) PICK=",${PART#L:},"; PART_I=$(echo "${PART#L:}" | tr , _);; *) PICK="";; esac   # "L:3,9,37": those lines only
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
run() { # label, env..., command
  local label=$1; shift
  LINE=$((LINE+1))
  if [ -n "$PICK" ]; then case "$PICK" in *",$LINE,"*) ;; *) return 0;; esac
  else [ $(( (LINE - 1) % PART_N + 1 )) -eq $PART_I ] || return 0; fi
  local line log="$OUT/run_${PART_I}_$LINE.log"
  # (each run is bounded by its own `timeout`; the box takes 7 minutes without output for a hang, and a fuzzer prints one line
  # at its end: a heartbeat beside it says which run is on and for how long)
  ( while sleep 60; do echo "$(date +%T) $label running" >> "$OUT/heartbeat_$PART_I.txt"; done ) &
  local beat=$!
  env "$@" > "$log" 2>&1
  kill $beat 2>/dev/null; wait $beat 2>/dev/null
  line=$(tail -1 "$log" | cut -c1-240)
  rm -f "$log"
  echo "$label: $line" | tee -a "$OUT/campaign_$PART_I.txt"
}
: > "$OUT/campaign_$PART_I.txt"
run "sweep default"            timeout -k 10 900 python3 tools/fuzz_sweep.py 2000000 $((6000*K))
run "sweep inputs x3 + ood"    FX_FUZZ_SCALE=3 FX_FUZZ_OOD=1 timeout -k 10 900 python3 tools/fuzz_sweep.py 2100000 $((4000*K))
run "sweep non-finite inputs"  FX_FUZZ_NAN=0.03 FX_FUZZ_OOD=1 timeout -k 10 900 python3 tools/fuzz_sweep.py 2200000 $((4000*K))
run "sweep 250 registers"      FX_FUZZ_REGS=250 timeout -k 10 900 python3 tools/fuzz_sweep.py 2300000 $((1500*K))
run "sweep 700 registers"      FX_FUZZ_REGS=700 timeout -k 10 900 python3 tools/fuzz_sweep.py 3200000 $((600*K))
run "sweep interpreter"        FX_KERNEL=asm timeout -k 10 900 python3 tools/fuzz_sweep.py 2400000 $((2000*K))
run "sweep interpreter (LDS)"  FX_KERNEL=asm_lds timeout -k 10 900 python3 tools/fuzz_sweep.py 2500000 $((1000*K))
run "sweep HIP kernel"         FX_KERNEL=hip timeout -k 10 900 python3 tools/fuzz_sweep.py 2600000 $((2000*K))
run "sweep per-instance operands"  FX_FUZZ_LANES=1 FX_FUZZ_OOD=1 timeout -k 10 900 python3 tools/fuzz_sweep.py 2800000 $((3000*K))
run "... interpreter"          FX_FUZZ_LANES=1 FX_FUZZ_OOD=1 FX_KERNEL=asm timeout -k 10 900 python3 tools/fuzz_sweep.py 2900000 $((1500*K))
run "... interpreter (LDS)"    FX_FUZZ_LANES=1 FX_FUZZ_OOD=1 FX_KERNEL=asm_lds timeout -k 10 900 python3 tools/fuzz_sweep.py 3000000 $((800*K))
run "... HIP kernel"           FX_FUZZ_LANES=1 FX_FUZZ_OOD=1 FX_KERNEL=hip timeout -k 10 900 python3 tools/fuzz_sweep.py 3100000 $((800*K))
run "api"                      timeout -k 10 900 python3 tools/fuzz_api.py 2000000 $((2500*K))
run "api wild registers"       FX_FUZZ_WILD=1 timeout -k 10 900 python3 tools/fuzz_api.py 2100000 $((2000*K))
run "api two shards"           FX_FUZZ_SHARDS=2 timeout -k 10 900 python3 tools/fuzz_api.py 2200000 $((1500*K))
run "api interpreter"          FX_KERNEL=asm timeout -k 10 900 python3 tools/fuzz_api.py 2300000 $((800*K))
run "api HIP kernel"           FX_KERNEL=hip timeout -k 10 900 python3 tools/fuzz_api.py 2400000 $((800*K))
run "stereo"                   timeout -k 10 900 python3 tools/fuzz_stereo.py 2000000 $((800*K))
run "delay lines"              timeout -k 10 900 python3 tools/fuzz_tram.py 2000000 $((3000*K))
run "delay lines x3"           FX_FUZZ_SCALE=3 timeout -k 10 900 python3 tools/fuzz_tram.py 2100000 $((3000*K))
run "delay lines DANE"         timeout -k 10 900 python3 tools/fuzz_tram.py 2200000 $((3000*K)) dane
run "delay lines DANE lanes"   timeout -k 10 900 python3 tools/fuzz_tram.py 2300000 $((2000*K)) dane_lanes
run "delay lines DANE shift"   timeout -k 10 900 python3 tools/fuzz_tram.py 2400000 $((2000*K)) dane_shift
run "DANE interpreter"         FX_KERNEL=asm timeout -k 10 900 python3 tools/fuzz_tram.py 2500000 $((1500*K)) dane_lanes
run "DANE interpreter shift"   FX_KERNEL=asm timeout -k 10 900 python3 tools/fuzz_tram.py 2600000 $((1500*K)) dane_shift
run "DANE interpreter (LDS)"   FX_KERNEL=asm_lds timeout -k 10 900 python3 tools/fuzz_tram.py 2700000 $((1000*K)) dane_lanes
run "DANE HIP kernel"          FX_KERNEL=hip timeout -k 10 900 python3 tools/fuzz_tram.py 2800000 $((1000*K)) dane_lanes
run "product cache"            timeout -k 10 900 python3 tools/fuzz_cse.py 2000000 $((4000*K))
run "product cache x3"         FX_FUZZ_SCALE=3 timeout -k 10 900 python3 tools/fuzz_cse.py 2100000 $((3000*K))
run "at scale 65553 x3"        FX_FUZZ_SCALE=3 FX_FUZZ_OOD=1 timeout -k 10 900 python3 tools/stress_fuzz.py $((400*K)) 65553
run "at scale 262144"          timeout -k 10 900 python3 tools/stress_fuzz.py $((400*K)) 262144
run "benchmark programs"       timeout -k 10 900 python3 tools/stress_scale.py 3
run "stages"                 timeout -k 10 900 python3 tools/fuzz_stages.py 2000000 $((1500*K))
run "stages, interpreter off" FX_STAGES=4 timeout -k 10 900 python3 tools/fuzz_sweep.py 2700000 $((1500*K))
run "api pinned buffers"       FX_FUZZ_PINNED=1 timeout -k 10 900 python3 tools/fuzz_api.py 2500000 $((1500*K))
run "api control panel"        FX_FUZZ_PANEL=1 timeout -k 10 900 python3 tools/fuzz_api.py 2600000 $((2000*K))
run "api control panel, interpreter"  FX_FUZZ_PANEL=1 FX_KERNEL=asm timeout -k 10 900 python3 tools/fuzz_api.py 2700000 $((600*K))
run "api control panel, wild registers"  FX_FUZZ_PANEL=1 FX_FUZZ_WILD=1 timeout -k 10 900 python3 tools/fuzz_api.py 2800000 $((2000*K))
run "api control panel, two shards"      FX_FUZZ_PANEL=1 FX_FUZZ_SHARDS=2 timeout -k 10 900 python3 tools/fuzz_api.py 2900000 $((1000*K))
echo done
