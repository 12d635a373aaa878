#!/bin/bash
# Profiles of the default bench (B2, 128+128) for profiles/: kernel-trace stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in
# separate PMC passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) and the MFMA / wait counters of the kernels.
#   bash scripts/profile_round.sh r02_a        (on the GPU box; writes gpurun_out/<tag>_*, copy what you keep to profiles/)
#   bash scripts/profile_round.sh r03_b4 B4    another workload (kernel stats + PMC only: bench.py's traffic record stays B2's)
set -e
TAG=${1:-rXX}
WL=${2:-B2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $WL --steps 60 --warmup 10 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- python3 $ARGS > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o p -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o p -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_pmc_sq -o p -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_sq2 -o p -- python3 $ARGS > /dev/null 2>&1 || true
cd $ROOT
cp $(find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
python3 scripts/pmc_summary.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write --json $OUT/${TAG}_pmc_traffic.json --workload $WL --n-local 256 --profile profiles/${TAG}_pmc_hbm_traffic.txt > $OUT/${TAG}_pmc_hbm_traffic.txt
python3 scripts/pmc_sq_summary.py $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_sq2 > $OUT/${TAG}_pmc_mfma.txt || true
# the traffic record bench.py quotes (it carries the source hash): in place before the bench line of this round is taken
[ "$WL" = B2 ] && cp $OUT/${TAG}_pmc_traffic.json profiles/pmc_traffic.json
python3 bench.py --workload $WL --steps 200 > $OUT/${TAG}_bench.json 2> /dev/null
