#!/bin/bash
# When a GPU call aborts, faults or is killed: keep its record BEFORE fixing anything.
#   tools/keep_fault_evidence.sh <round tag, e.g. r04> <short name> [log files under gpurun_out/ ...]
# copies gpurun's verdict of the last call (gpurun_out/.last_call.json) and the named logs into profiles/<tag>_faults/<name>/,
# with the commit the tree was at.  Do not re-run the failing command to "see it again": find the cause from this record and
# the code, fix, test once.
TAG=${1:?round tag}; NAME=${2:?short name}; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/profiles/${TAG}_faults/$NAME
mkdir -p "$OUT"
cp "$ROOT/gpurun_out/.last_call.json" "$OUT/last_call.json" 2>/dev/null
for f in "$@"; do cp "$ROOT/gpurun_out/$f" "$OUT/" 2>/dev/null || cp "$f" "$OUT/" 2>/dev/null; done
( cd "$ROOT" && git rev-parse HEAD && git status --short | head -40 ) > "$OUT/tree.txt"
ls -la "$OUT"
