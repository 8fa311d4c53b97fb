#!/bin/bash
# A/B of two builds of the library on ONE GPU box (boxes differ by several percent):
#   tools/ab.sh <libA.so> <libB.so> [rounds]     prints the edge kernel times of alternating runs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in "$A" "$B"; do
    NMRGNN_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
k={r['kernel']: round(r['avg_ms'],3) for r in d['roofline_all'] if r['kernel'].startswith('edge_')}
print('$L', 'ms/step %.3f' % d['ms_per_step'], 'inference %.3f' % d['inference']['ms_per_step'], k)"
  done
done
