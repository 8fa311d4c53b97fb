#!/bin/bash
# A/B of two builds of the library on ONE GPU box by the hipEvent per-kernel times of the bench step:
#   tools/kernel_ab.sh <prefix of the kernels to print> <libA.so> <libB.so> [rounds]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
P=$1; A=$2; B=$3; R=${4:-2}
for i in $(seq $R); do
  for L in "$A" "$B"; do
    NMRGNN_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | P=$P L=$L python -c "
import json, os, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
k = {r['kernel']: round(r['avg_ms'] * 1e3, 1) for r in d['roofline_all'] if r['kernel'].startswith(os.environ['P'])}
print(os.environ['L'], 'ms/step %.4f' % d['ms_per_step'], k)"
  done
done
