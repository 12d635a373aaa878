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
T=$(find $OUT/${TAG}_nr -name "*kernel_trace.csv" | head -1)
python3 $ROOT/scripts/kstats_by_grid.py "$T" ntx_ us_ mb_
