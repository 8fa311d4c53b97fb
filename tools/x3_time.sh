#!/bin/bash
# forward edge kernel time in the bench step, fp32 MFMA vs bf16x3 (training = with z_save, inference = without)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for M in fp32 bf16x3; do
  NG_EDGE_MATH=$M python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
k=[r for r in d['roofline_all'] if r['kernel'].startswith('edge_f')]
print('$M', 'ms/step %.3f' % d['ms_per_step'], 'inference ms %.3f' % d['inference']['ms_per_step'], [(r['kernel'], round(r['avg_ms'],3)) for r in k])"
done
