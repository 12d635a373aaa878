#!/bin/bash
# SQ counters of one rank's step at a given world size:   bash scripts/pmc_rank.sh TAG W WORKLOAD BT BTU
TAG=$1; W=$2; WL=$3; BT=$4; BTU=$5
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_pm1 $OUT/${TAG}_pm2 $OUT/${TAG}_pm3
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_pm1 -o p -- python3 $ROOT/scripts/rank_cost.py $W $WL $BT $BTU > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/${TAG}_pm2 -o p -- python3 $ROOT/scripts/rank_cost.py $W $WL $BT $BTU > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pm3 -o p -- python3 $ROOT/scripts/rank_cost.py $W $WL $BT $BTU > /dev/null 2>&1
python3 $ROOT/scripts/pmc_instmix_summary.py $OUT/${TAG}_pm1 $OUT/${TAG}_pm2 $OUT/${TAG}_pm3
