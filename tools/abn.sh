#!/bin/bash
# A/B/... of several builds of the library on ONE GPU box:  ROUNDS=2 tools/abn.sh <libA.so> <libB.so> ...
# per build and round: edge kernels alone (tools/edge_ab.py), then the bench step with its per-kernel table
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${ROUNDS:-2}
for i in $(seq $R); do
  for L in "$@"; do
    echo -n "$L: "; NMRGNN_HIP_LIB=$PWD/$L python tools/edge_ab.py 2>&1 | tail -1
    NMRGNN_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
k={r['kernel']: round(r['ms_per_step'],3) for r in d['roofline_all'][:9]}
print('   step %.3f ms' % d['ms_per_step'], k)"
  done
done
