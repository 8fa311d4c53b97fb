#!/bin/bash
# HBM traffic per launch of every kernel of the bench step (rocprofv3 PMC, separate passes as the
# MI355X guide prescribes).  Writes gpurun_out/pmc_traffic.json — copy it to profiles/.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=gpurun_out/pmc_${C}; rm -rf "$OUT"
  rocprofv3 --pmc $C --output-format csv -d "$OUT" -o t -- \
      python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extras > gpurun_out/pmc_${C}.log 2>&1
done
python - <<'PY'
import csv, glob, json, collections
res = collections.defaultdict(dict)
names = {"edge_fused_bwd_kernel": "edge_fused_bwd",
         "edge_fused_fwd_kernel": "edge_fused_fwd", "edge_fwd_h2_kernel<2>": "edge_fwd_h2", "edge_fwd_h2_kernel<1>": "edge_fwd_h2_rowmajor", "edge_fwd_h2_kernel<0>": "edge_fwd_h2_inference", "edge_bwd_h2_kernel": "edge_bwd_h2",
         "mp_win_fwd_kernel": "mp_win_fwd_eight_or_sixteen_wave", "mp_win16_fwd_kernel": "mp_win_fwd_sixteen_wave", "mp_wave_fwd_kernel": "mp_win_fwd",
         "mp_win_bwd_edge_kernel": "mp_win_bwd_edge", "mp_win_bwd_node_kernel": "mp_win_bwd_node",
         "fc_fwd_kernel": "fc_fused_fwd", "fc_bwd_kernel": "fc_fused_bwd"}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{C}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    tot = collections.defaultdict(float); cnt = collections.Counter(); seen = set()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        tot[k] += float(row["Counter_Value"])
        if (k, row["Dispatch_Id"]) not in seen:
            seen.add((k, row["Dispatch_Id"])); cnt[k] += 1
    for k in tot:
        for pat, short in names.items():
            if pat in k:
                for s in short.split("|"):
                    res[s][C + "_KB"] = tot[k] / cnt[k]
import os, sys
sys.path.insert(0, os.getcwd())
import bench
res["_meta"] = {"commit": os.environ.get("COMMIT", "unknown"), "tool": "tools/pmc_traffic.sh",
                "source_digest": {k: bench.source_digest(k) for k in bench.KERNEL_SOURCES}}
json.dump(res, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
