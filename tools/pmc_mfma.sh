#!/bin/bash
# Matrix-pipe utilisation and effective clock per kernel of the bench step (rocprofv3 PMC only, no traces).
#   MfmaUtil  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMDs)      (share of cycles the MFMA pipe is busy)
#   clock     = GRBM_GUI_ACTIVE / kernel duration                        (DVFS: the chip clocks to its power budget)
# Writes gpurun_out/pmc_mfma.json — copy it to profiles/.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
P=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  P=$((P+1)); OUT=gpurun_out/pmc_mfma_$P; rm -rf "$OUT"
  rocprofv3 --pmc $SET --output-format csv -d "$OUT" -o t -- \
      python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extras > gpurun_out/pmc_mfma_$P.log 2>&1
done
python - <<'PY'
import csv, glob, json, collections
val = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
dur = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_mfma_*/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    if rows:
        print(f, list(rows[0].keys()))
    for r in rows:
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        val[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
        if "Start_Timestamp" in r and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
res = {}
for k, d in val.items():
    e = {c: v / max(1, len(cnt[(k, c)])) for c, v in d.items()}
    n = max(1, len(cnt[(k, "GRBM_GUI_ACTIVE")]))
    if e.get("GRBM_GUI_ACTIVE", 0) > 0:
        e["mfma_util_pct"] = 100.0 * e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / (e["GRBM_GUI_ACTIVE"] / 8.0)   # GRBM counts per XCD
        if dur[k] > 0:
            e["dur_us"] = dur[k] / n / 1e3
            e["clock_GHz"] = e["GRBM_GUI_ACTIVE"] / (dur[k] / n)
    res[k] = e
keep = {k: v for k, v in res.items() if v.get("GRBM_GUI_ACTIVE", 0) > 20000}
import os, sys
sys.path.insert(0, os.getcwd())
import bench
keep["_meta"] = {"commit": os.environ.get("COMMIT", "unknown"), "tool": "tools/pmc_mfma.sh",
                 "source_digest": {k: bench.source_digest(k) for k in bench.KERNEL_SOURCES}}
json.dump(keep, open("gpurun_out/pmc_mfma.json", "w"), indent=1)
keep.pop("_meta")
for k, v in sorted(keep.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:12]:
    print(k[:60], {a: round(b, 2) for a, b in v.items()})
PY
