#!/bin/bash
# LDS bank-conflict cycles per transposing read for the address patterns of tools/ubench/trbank2.hip (24 launches of 256 x 512
# threads, 160,000 ds_read_b64_tr_b16 per wave each): is the conflict count of edge_bwd_h2 (50 M cycles per launch) avoidable?
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/pmc_trbank
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_trbank -o t -- tools/ubench/trbank2 > gpurun_out/pmc_trbank_stdout.txt 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_trbank/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
timing = [l for l in open("gpurun_out/pmc_trbank_stdout.txt") if l.startswith("S ")]
out = ["dispatches come in pairs (warm-up + timed) per address pattern; reads per launch = 256 WG x 8 waves x 160,000 = 327.68 M wave reads"]
for i, d in enumerate(sorted(by)):
    c = by[d]
    if i % 2 == 1:
        n = 256 * 8 * 160000.0
        out.append("%-44s conflict cycles %12.0f = %.2f per wave read; LDS instructions %12.0f; SQ_ACTIVE_INST_LDS %12.0f" % (
            timing[i // 2].strip() if i // 2 < len(timing) else "?", c.get("SQ_LDS_BANK_CONFLICT", 0), c.get("SQ_LDS_BANK_CONFLICT", 0) / n,
            c.get("SQ_INSTS_LDS", 0), c.get("SQ_ACTIVE_INST_LDS", 0)))
open("gpurun_out/pmc_trbank.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
