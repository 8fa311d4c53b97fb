#!/bin/bash
# A/B of two builds of the library on ONE GPU box: per-kernel averages of the bench step (alternating runs)
#   tools/ab_kernels.sh <libA.so> <libB.so> [rounds]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do
  for L in "$A" "$B"; do
    NMRGNN_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
k={r['kernel']: round(r['avg_ms']*1e3,1) for r in d['roofline_all'][:9]}
print('$L'.split('/')[-1], 'ms/step %.3f' % d['ms_per_step'], k)"
  done
done
