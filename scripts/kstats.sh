#!/bin/bash
# Quick per-kernel averages (rocprofv3 --kernel-trace --stats) of the default bench, for before / after comparisons at
# kernel granularity (step-level A/B cannot resolve 1 us):   bash scripts/kstats.sh TAG [WORKLOAD] [LIB]
TAG=${1:-k}
WL=${2:-B2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
[ -n "$3" ] && export CMLPL_LIB=$3
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_ks
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks -o t -- python3 $ROOT/bench.py --workload $WL --steps 60 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_ks.json 2> /dev/null
F=$(find $OUT/${TAG}_ks -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "cmlpl" in r["Name"] and int(r["Calls"]) >= 60]
tot = 0.0
per = min(int(r["Calls"]) for r in rows)
for r in rows:
    name = r["Name"].replace("void ", "").replace("cmlpl::", "").split("(")[0]
    avg = float(r["AverageNs"]) / 1e3
    tot += avg * int(r["Calls"]) / per
    print(f"  {name:34s} {avg:7.2f} us x{int(r['Calls'])//per}")
print(f"  sum per step {tot:7.1f} us")
PY
