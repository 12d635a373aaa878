#!/bin/bash
# Per-kernel device times (rocprofv3 --kernel-trace --stats) of the SURVEY 8(f) rows N4 / N2 as scripts/bench_next_rows.py
# calls them:   bash scripts/kstats_next_rows.sh TAG      -> gpurun_out/TAG_next_rows_kernel_stats.csv + a table
TAG=${1:-k}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_nr
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_nr -o t -- python3 $ROOT/scripts/bench_next_rows.py n4 n2 > $OUT/${TAG}_nr.txt 2> /dev/null
F=$(find $OUT/${TAG}_nr -name "*kernel_stats.csv" | head -1)
cp $F $OUT/${TAG}_next_rows_kernel_stats.csv
python3 - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "cmlpl" in r["Name"] and any(k in r["Name"] for k in ("ntx_", "us_", "mb_")):
        name = r["Name"].replace("void ", "").replace("cmlpl::", "").split("(")[0]
        print(f"  {name:34s} {float(r['AverageNs']) / 1e3:8.2f} us  x{r['Calls']}")
PY
