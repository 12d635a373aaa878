#!/bin/bash
# A/B/C... of several (library, environment) configurations on ONE box, in alternation (boxes differ by several percent):
#   bash scripts/ab_multi.sh REPS "bench args" "NAME|ENV=.. ENV=..|lib.so" ...
REPS=$1; ARGS=$2; shift 2
for i in $(seq 1 $REPS); do
  for cfg in "$@"; do
    IFS='|' read -r name envs lib <<< "$cfg"
    echo -n "$name: "
    env $envs CMLPL_LIB=$lib python bench.py --steps 200 --no-cpu-baseline $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; o=d['roofline_others']; print('%.4f ms/step  dom %.1f us  others %s' % (d['ms_per_step'], r['ms_per_launch']*1e3, ' '.join('%.1f' % (x['ms_per_launch']*1e3) for x in o)))" || echo failed
  done
done
