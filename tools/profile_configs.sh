#!/bin/bash
# Evidence pass for the single-GPU configurations (run on the GPU box through gpurun):
#   tools/profile_configs.sh <tag> [configs...]
# A config is a bench.py --config name, or label=name:args for another shape of it (config5 alone is the default workload:
# all 2 097 152 instances of configs[4] on the one GPU, strong scaling; `config5_shard=config5:--scaling,weak` is the 1/8 shard
# that every GPU of an 8-GPU job runs) - the label names the files.
# Per config: the bench line without a profiler, rocprofv3 --kernel-trace --stats of the same command,
# and the SQ counter passes (tools/pmc_sq.txt; fewer launches - counters serialise the kernel).
# Everything lands under gpurun_out/<tag>/; copy what is to be judged into profiles/.
set -o pipefail
TAG=${1:-r02}
shift
CONFIGS=${@:-config2 config3 config4 config5}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for spec in $CONFIGS; do
  c=${spec%%=*}; rest=${spec#*=}; [ "$rest" = "$spec" ] && rest=$c
  cfg=${rest%%:*}; more=${rest#*:}; [ "$more" = "$rest" ] && more=""
  more=${more//,/ }
  echo "== $c: bench ($cfg $more)" 
  python3 $ROOT/bench.py --config $cfg $more --cpu-seconds 0 --no-extras > $OUT/${c}_bench.json 2> $OUT/${c}_bench.err || exit 1
  cat $OUT/${c}_bench.json
  echo "== $c: kernel trace"
  rocprofv3 --kernel-trace --stats -d $OUT/${c}_trace -o ${c} --output-format csv -- python3 $ROOT/bench.py --config $cfg $more --cpu-seconds 0 --no-extras --parity-instances 0 > $OUT/${c}_bench_under_rocprof.json 2> $OUT/${c}_trace.err || exit 1
  echo "== $c: HBM traffic counters"
  rocprofv3 -i $ROOT/tools/pmc_hbm.txt --kernel-trace -d $OUT/${c}_hbm -o ${c} --output-format csv -- python3 $ROOT/bench.py --config $cfg $more --cpu-seconds 0 --steps 3 --warmup 1 --no-extras --parity-instances 0 > $OUT/${c}_hbm.json 2> $OUT/${c}_hbm.err || exit 1
  echo "== $c: SQ counters"
  rocprofv3 -i $ROOT/tools/pmc_sq.txt -d $OUT/${c}_pmc -o ${c} --output-format csv -- python3 $ROOT/bench.py --config $cfg $more --cpu-seconds 0 --steps 2 --warmup 1 --no-extras --parity-instances 0 > $OUT/${c}_pmc.json 2> $OUT/${c}_pmc.err || exit 1
  echo "== $c: VALU issue counters (dual-issued quad-cycles, instruction classes)"
  rocprofv3 -i $ROOT/tools/pmc_valu.txt -d $OUT/${c}_valu -o ${c} --output-format csv -- python3 $ROOT/bench.py --config $cfg $more --cpu-seconds 0 --steps 2 --warmup 1 --no-extras --parity-instances 0 > $OUT/${c}_valu.json 2> $OUT/${c}_valu.err || exit 1
  # the summaries that get committed under profiles/ (bench.py reads <tag>_<config>_pmc_valu.json and <tag>_hbm_traffic_<config>.json)
  python3 $ROOT/tools/pmc_summary.py $OUT/${c}_pmc --bench $OUT/${c}_bench.json > $OUT/${TAG}_${c}_pmc_sq.json
  python3 $ROOT/tools/pmc_summary.py $OUT/${c}_valu --bench $OUT/${c}_bench.json > $OUT/${TAG}_${c}_pmc_valu.json
  python3 - $OUT/${c}_bench.json $OUT/${c}_hbm $cfg > $OUT/${TAG}_hbm_traffic_${c}.json <<PY
import json, subprocess, sys
b = json.loads([l for l in open(sys.argv[1]).read().split("\n") if l.startswith("{")][-1])
out = subprocess.run([sys.executable, "$ROOT/tools/hbm_traffic.py", sys.argv[2], sys.argv[3], str(b["config"]["instances_per_gpu"]), str(b["config"]["samples_per_step"]),
                      str(b["roofline"]["algorithmic_bytes_per_launch"])], stdout=subprocess.PIPE, text=True).stdout
d = json.loads(out)
d["code_hash"] = (b["roofline"]["valu"] or {}).get("code_hash")
print(json.dumps(d, indent=1))
PY
  cp $OUT/${c}_bench.json $OUT/${TAG}_${c}_bench.json
  f=$(find $OUT/${c}_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_${c}_kernel_stats.csv
  # (gpurun copies at most 64 MiB back: the raw traces and counter tables stay on the box, the summaries above are what is kept)
  du -sh $OUT/${c}_trace $OUT/${c}_hbm $OUT/${c}_pmc $OUT/${c}_valu 2>/dev/null
  rm -rf $OUT/${c}_trace $OUT/${c}_hbm $OUT/${c}_pmc $OUT/${c}_valu
done
ls $OUT | grep "^${TAG}_"
echo done
