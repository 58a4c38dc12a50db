#!/bin/bash
# rocprofv3 timeline (kernels + memory copies) of host-fed 32-sample blocks: tools/realtime_timeline.sh <instances>
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-131072}
OUT=$ROOT/gpurun_out/rt_timeline_$N
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT -o t --output-format csv -- python3 $ROOT/tools/realtime_timeline.py $N 40 > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, os
root = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("fx_"):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel " + r["Kernel_Name"][:16]))
for f in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", "?")) + " " + r.get("Bytes", r.get("Size", "?"))))
ev.sort()
# the last block: everything behind the third-to-last gap of > 100 us without activity... simply print the last 40 events relative to the first of them
tail = ev[-40:]
t0 = tail[0][0]
for s, e, what in tail:
    print("%9.1f us .. %9.1f us  (%7.1f)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, what))
PY
rm -rf $OUT/*/ 2>/dev/null
