# same-box A/B of conv1's tap loops on two fp16 pieces (CMLPL_F16X2=1) against the three-piece bf16 default
for X in ${AB_MODES:-0 1 0 1}; do
  echo "== CMLPL_F16X2=$X"
  CMLPL_F16X2=$X python bench.py --workload ${1:-B2} --steps 300 --warmup 30 --no-cpu-baseline --breakdown 2>/tmp/ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  step %.4f ms   dominant %s  %.2f us  frac %.3f' % (d['ms_per_step'], d['roofline']['kernel'][:34], 1e3 * d['roofline']['ms_per_launch'], d['roofline']['frac']))"
  grep -A 12 "per-kernel" /tmp/ab.err | grep "conv1_fwd\|conv1_dgrad\|conv1_wgrad\|adam\|sum"
done
