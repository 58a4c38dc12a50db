#!/bin/bash
# Stage counts measured over program shapes and batch sizes (tools/stage_probe.py: bit-exact vs the oracle on sampled instances,
# kernel time of one launch of S samples), K = 0 being the library's own choice - what the policy is calibrated with and checked
# against.   tools/stage_policy_probe.sh [S] > gpurun_out/stage_policy.txt     (on the GPU box, one call)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
S=${1:-2048}
for spec in "config2 1024" "config2 4096" "config2 16384" "config2 32768" "config2 49152" "config2 65536" "config2 98304" "config2 131072" \
            "wide12 4096" "wide12 32768" "wide12 98304" "mixed_stages 4096" "mixed_stages 32768" "mixed_stages 98304" "config3 4096" "config4 4096"; do
  set -- $spec
  timeout -k 10 300 python3 tools/stage_probe.py $1 $2 $S 0 1 2 4 8 2>&1 | grep -v amdgpu.ids
done
