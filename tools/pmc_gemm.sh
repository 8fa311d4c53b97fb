#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_gemm; rm -rf "$OUT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d "$OUT" -o t -- python tools/eval_bench.py > gpurun_out/pmc_gemm.log 2>&1
OUT2=gpurun_out/pmc_gemm2; rm -rf "$OUT2"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT2" -o t -- python tools/eval_bench.py > gpurun_out/pmc_gemm2.log 2>&1
python - <<'PY'
import csv, glob, collections
val = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_gemm*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_h2" not in k: continue
        if int(r["Grid_Size"]) < 200000: continue
        k = k.split("(")[0]
        val[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, d in val.items():
    e = {c: v / len(cnt[(k, c)]) for c, v in d.items()}
    w = e.get("SQ_WAVE_CYCLES", 1)
    print(k, "mfma busy %.1f%%" % (100 * e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (e.get("GRBM_GUI_ACTIVE", 8) / 8)),
          {c: round(100 * e[c] / w, 1) for c in e if c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT")},
          "conflict/cycle", round(e.get("SQ_LDS_BANK_CONFLICT", 0) / max(e.get("GRBM_GUI_ACTIVE", 8) / 8 * 256, 1), 3) if "GRBM_GUI_ACTIVE" in e else e.get("SQ_LDS_BANK_CONFLICT"))
PY
