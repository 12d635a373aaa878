#!/bin/bash
# End-to-end rate of the drop-in driver (train.py: loader + step + logging read-backs) next to bench.py's step:
#   bash scripts/train_e2e.sh [SHAPE]      (on the GPU box)
WL=${1:-B2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
b=$(python3 bench.py --workload $WL --steps 200 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
echo "bench.py --workload $WL: $b ms/step"
for mode in "" "--graph"; do
  for ppb in 10 100; do
    out=$(python3 train.py --synthetic $WL --no_eval --num_epochs 20 --num_unlabel 10000 --print_per_batches $ppb $mode 2>/dev/null | grep "^after the first epoch:")
    ms=$(echo $out | awk '{print $(NF-1)}')
    python3 -c "print('train.py --synthetic $WL $mode --print_per_batches $ppb: $out (%.1f %% over the bench step)' % (($ms / $b - 1) * 100))"
  done
done
