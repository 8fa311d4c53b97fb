cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/prof_frame
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_frame -o frame -- python tools/frame_prof.py > gpurun_out/frame_prof.log 2>&1
find gpurun_out/prof_frame -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03_frame_kernel_stats.csv \;
head -30 gpurun_out/r03_frame_kernel_stats.csv | cut -c1-160
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_frame/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last full forward: find the last embed kernel and print until the head
idx = [i for i, r in enumerate(rows) if 'embed' in r['Kernel_Name']]
s = idx[-2]
t0 = int(rows[s]['Start_Timestamp'])
prev_end = t0
for r in rows[s-3:idx[-1]-3]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%8.1f us  +gap %5.1f  dur %6.1f  %s" % ((st - t0) / 1e3, (st - prev_end) / 1e3, (en - st) / 1e3, r['Kernel_Name'][:70]))
    prev_end = en
PY
