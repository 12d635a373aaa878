#!/bin/bash
# Per-kernel device times of one rank's step at a given world size (scripts/rank_cost.py under rocprofv3 --kernel-trace):
#   bash scripts/kstats_rank.sh TAG W WORKLOAD BT BTU [name-substring ...]
TAG=$1; W=$2; WL=$3; BT=$4; BTU=$5; shift 5
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_rk
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_rk -o t -- python3 $ROOT/scripts/rank_cost.py $W $WL $BT $BTU > $OUT/${TAG}_rk.txt 2> /dev/null
T=$(find $OUT/${TAG}_rk -name "*kernel_trace.csv" | head -1)
python3 $ROOT/scripts/kstats_by_grid.py "$T" "$@"
