#!/bin/bash
# A/B of two TREES (e.g. the previous round's checkout under .ab_r03/ against this one) on ONE box, in alternation:
#   bash scripts/ab_tree.sh REPS "bench args" "NAME|ENV=..|dir" ...      (dir holds bench.py + cmlpl_amd/ with its own library)
REPS=$1; ARGS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in $(seq 1 $REPS); do
  for cfg in "$@"; do
    IFS='|' read -r name envs dir <<< "$cfg"
    echo -n "$name: "
    (cd $ROOT/$dir && env $envs python bench.py --steps 200 --no-cpu-baseline $ARGS 2>/dev/null) | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; o=d['roofline_others']; print('%.4f ms/step  dom %.1f us  others %s' % (d['ms_per_step'], r['ms_per_launch']*1e3, ' '.join('%.1f' % (x['ms_per_launch']*1e3) for x in o)))" || echo failed
  done
done
