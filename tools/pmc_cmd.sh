#!/bin/bash
# rocprofv3 PMC pass over an arbitrary command.  Usage: tools/pmc_cmd.sh <tag> "<COUNTERS>" <cmd...>
TAG=$1; CTRS=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_${TAG}; rm -rf "$OUT"
rocprofv3 --pmc $CTRS --output-format csv -d "$OUT" -o "$TAG" -- "$@" > gpurun_out/pmc_${TAG}.log 2>&1 || tail -5 gpurun_out/pmc_${TAG}.log
python - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter csv"); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen=set()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"].split("(")[0][:50]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key=(k,row["Dispatch_Id"])
    if key not in seen: seen.add(key); cnt[k]+=1
names = sorted({c for v in agg.values() for c in v})
print("kernel,calls," + ",".join(names))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get(names[0], 0))[:8]:
    print(k + "," + str(cnt[k]) + "," + ",".join(f"{v.get(c,0)/cnt[k]:.4g}" for c in names))
PY
