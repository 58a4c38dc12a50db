#!/bin/bash
# Many launches of every single-GPU configuration at its BASELINE size, then the device's LAST block against the oracle after
# replaying ALL blocks on the CPU (bench.py's in-run parity): state carried over hundreds to thousands of launches.
#   tools/soak.sh > gpurun_out/soak.txt     (GPU box; ~4 minutes)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for spec in "config5 600 48" "config4 200 48" "config3 1500 48" "config2 4000 40"; do
  set -- $spec
  timeout -k 10 400 python3 bench.py --config $1 --steps $2 --warmup 1 --parity-instances $3 --no-extras --cpu-seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$1 $2 launches', d['value'], d['parity'])"
done
