#!/bin/bash
# same-box A/B of library builds by rocprof kernel averages: ROUNDS=2 tools/abk.sh <libA.so> <libB.so> ... ; prints the step time
# and the per-kernel average (us) of every kernel above 3 us
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${ROUNDS:-2}
for i in $(seq $R); do
  for L in "$@"; do
    rm -rf gpurun_out/abk_prof
    NMRGNN_HIP_LIB=$PWD/$L rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk_prof -o abk -- python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > gpurun_out/abk.json 2>/dev/null
    python - "$L" <<'PY'
import csv, glob, json, sys
d = json.loads([l for l in open('gpurun_out/abk.json') if l.startswith('{')][0])
f = glob.glob('gpurun_out/abk_prof/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6
print(sys.argv[1], 'step %.3f ms (under rocprof)' % d['ms_per_step'], 'kernel time total %.1f ms' % tot)
print('   ', {r['Name'].replace('(anonymous namespace)::','').split('(')[0].split('::')[-1][:26]: round(float(r['AverageNs']) / 1e3, 1) for r in rows if float(r['AverageNs']) > 3000 and int(r['Calls']) >= 20})
PY
  done
done
