# the pair weight-gradient launch's split of the CUs between conv1's and conv2's workgroups (groups per network and kernel
# row), re-swept on the two-piece planes: launch + reduce per step
for pg in "0 0" "32 10" "32 12" "36 10" "36 6" "28 12" "30 10" "34 8" "40 2"; do
  set -- $pg
  CMLPL_WGRAD3_PG1=$1 CMLPL_WGRAD3_PG2=$2 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --breakdown 2>/tmp/e.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PG1=$1 PG2=$2  step %.4f ms' % d['ms_per_step'], end='  ')"
  grep "conv1_wgrad\|conv1_wred" /tmp/e.txt | awk '{printf "%s %s us  ", $1, $2}'; echo
done
