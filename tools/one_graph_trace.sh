# ordered kernel trace of ONE single-graph training step (bench.py: train_one_graph_per_step): start, gap, duration
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/prof_one
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_one -o one -- python tools/one_graph.py 30 > gpurun_out/one_graph_traced.json 2>gpurun_out/one_graph_err.log
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_one/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
s, e = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[s]['Start_Timestamp'])
prev_end = t0
out = []
gaps = busy = 0.0
for r in rows[s:e]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gaps += max(0, st - prev_end) / 1e3
    busy += (en - st) / 1e3
    out.append("%8.1f us  +gap %5.1f  dur %7.1f  %s" % ((st - t0) / 1e3, (st - prev_end) / 1e3, (en - st) / 1e3, r['Kernel_Name'].replace('(anonymous namespace)::', '')[:80]))
    prev_end = en
out.append("launches %d  span %.1f us  kernel time %.1f us  sum of gaps %.1f us (gaps are inflated by the tracer)" % (e - s, (prev_end - t0) / 1e3, busy, gaps))
open('gpurun_out/one_graph_trace.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
