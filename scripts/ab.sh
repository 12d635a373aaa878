#!/bin/bash
# A/B of two builds of the library in ONE process sequence on ONE box (boxes differ by several percent):
#   bash scripts/ab.sh cmlpl_amd/libbase.so cmlpl_amd/libcmlpl_hip.so [bench args]
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in $A $B; do
    echo -n "$L: "; CMLPL_LIB=$L python bench.py --steps 200 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/step' % d['ms_per_step'])"
  done
done
