#!/bin/bash
# Instruction mix of the step's kernels (which unit bounds a CU when the matrix pipe is idle 70 % of the time):
# VALU / SALU / LDS / VMEM instruction counts and the cycles each unit was issuing, one rocprofv3 --pmc pass per group
# (separate passes, no trace domains).     bash scripts/pmc_instmix.sh TAG [WORKLOAD]
TAG=${1:-mix}
WL=${2:-B2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline"
(rocprofv3 -L 2> /dev/null || rocprofv3-avail list 2> /dev/null) | grep -ao "SQ_[A-Z0-9_]*" | sort -u > $OUT/${TAG}_sq_counters.txt
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_FLAT SQ_INSTS_BRANCH"; do
  i=$((i + 1))
  rm -rf $OUT/${TAG}_mix$i
  rocprofv3 --pmc $grp --output-format csv -d $OUT/${TAG}_mix$i -o p -- python3 $ARGS > /dev/null 2> $OUT/${TAG}_mix$i.err || echo "pass $i failed ($grp)"
done
python3 $ROOT/scripts/pmc_instmix_summary.py $OUT/${TAG}_mix[0-9] > $OUT/${TAG}_instmix.txt
cat $OUT/${TAG}_instmix.txt
