#!/bin/bash
# same-box comparison of library variants on the forward window kernel: tools/wv_ab.sh <lib.so> ... ; prints mp_win_fwd's bench
# bracket (us per launch) and the step, two alternating rounds; stamped variants also print their phase table to <lib>.err
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
  for L in "$@"; do
    NMRGNN_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2> gpurun_out/$(basename $L).err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
k={r['kernel']: round(r['avg_ms']*1e3,1) for r in d['roofline_all'][:9]}
print('$L'.split('/')[-1], 'ms/step %.3f' % d['ms_per_step'], 'mp_win_fwd', k.get('mp_win_fwd'), 'bwd_edge', k.get('mp_win_bwd_edge'), 'bwd_node', k.get('mp_win_bwd_node'))"
  done
done
