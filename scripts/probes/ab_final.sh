# same box, alternating: round 5's kernels (CMLPL_F16X2=0 CMLPL_ZERO_SKIP=0 CMLPL_BWD_PAIR=0: three-piece products, nothing
# skipped, twin pairing) | two-piece products with nothing skipped | the default
for w in B2 B4 B5 P; do
  for rep in 1 2; do
    for m in "r5:CMLPL_F16X2=0 CMLPL_ZERO_SKIP=0 CMLPL_BWD_PAIR=0" "two-piece-dense:CMLPL_ZERO_SKIP=0 CMLPL_BWD_PAIR=0" "default:CMLPL_NOOP=1"; do
      name=${m%%:*}; envs=${m#*:}
      env $envs python bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w  %-16s %.4f ms/step  %8.0f patches/s' % ('$name', d['ms_per_step'], d['value']))"
    done
  done
done
