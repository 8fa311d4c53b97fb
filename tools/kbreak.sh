#!/bin/bash
# per-kernel time table of one bench configuration: tools/kbreak.sh [ENV=VAL ...]
# (run on the GPU box through gpurun)
for kv in "$@"; do export "$kv"; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith('{'):
        continue
    d = json.loads(line)
    print('value %.3fM atoms/s  %.3f ms/step  (profiled sum %.3f)' % (d['value'] / 1e6, d['ms_per_step'], d.get('profiled_ms_per_step', 0)))
    for k in d.get('roofline_all', []):
        print('  %-22s %8.4f ms/step  x%-4g avg %.4f ms  %s' % (k['kernel'], k['ms_per_step'], k['launches_per_step'], k['avg_ms'],
              ('%.3f %s' % (k['frac'], k['bound'])) if 'frac' in k else ''))
"
