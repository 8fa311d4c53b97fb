#!/bin/bash
# PMC passes over the F=256 training step (tools/f256_ab.py): matrix-pipe busy, wait classes, LDS conflicts per kernel.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
P=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  P=$((P+1)); OUT=gpurun_out/pmc_f256_$P; rm -rf "$OUT"
  rocprofv3 --pmc $SET --output-format csv -d "$OUT" -o t -- python tools/f256_ab.py > gpurun_out/pmc_f256_$P.log 2>&1
done
python - <<'PY'
import csv, glob, json, collections
val = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_f256_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        val[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
res = {}
for k, d in val.items():
    e = {c: v / max(1, len(cnt[(k, c)])) for c, v in d.items()}
    if e.get("GRBM_GUI_ACTIVE", 0) > 100000:
        e["mfma_busy_pct"] = 100.0 * e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / (e["GRBM_GUI_ACTIVE"] / 8.0)
        res[k] = e
json.dump(res, open("gpurun_out/pmc_f256.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:14]:
    wc = max(v.get("SQ_WAVE_CYCLES", 1), 1)
    print("%-44s gui %9.0f busy %5.1f%% wait_any %4.2f wait_inst %4.2f active %4.2f | valu %8.0f mfma %8.0f lds %8.0f vmem_rd %7.0f conflicts %9.0f" % (
        k[:44], v["GRBM_GUI_ACTIVE"], v["mfma_busy_pct"], v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc,
        v.get("SQ_ACTIVE_INST_ANY", 0) / wc, v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_MFMA", 0), v.get("SQ_INSTS_LDS", 0),
        v.get("SQ_INSTS_VMEM_RD", 0), v.get("SQ_LDS_BANK_CONFLICT", 0)))
PY
