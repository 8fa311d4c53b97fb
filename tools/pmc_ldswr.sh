#!/bin/bash
# LDS bank-conflict cycles per access for the non-transposing LDS accesses of edge_bwd_h2 (tools/ubench/ldswr.hip)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/pmc_ldswr
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_ldswr -o t -- tools/ubench/ldswr > gpurun_out/pmc_ldswr_stdout.txt 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_ldswr/**/*counter_collection.csv", recursive=True)[0]
by = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
timing = [l.strip() for l in open("gpurun_out/pmc_ldswr_stdout.txt") if l.startswith("pattern")]
names = ["G-image piece write (ds_write_b64, stride 272)", "Z-image piece write (ds_write_b64, stride 264)", "G rows as dZ B operand (ds_read_b128)",
         "fp32 staging write (ds_write_b128, stride 528)", "staging column read (ds_read_b32)", "G-image write, rows 16.. shifted by 16 B (variant)", "G-image pieces as ds_write_b128 after a lane swap (variant)", "control: contiguous ds_write_b64"]
out = ["accesses per launch = 256 WG x 8 waves x 80,000 = 163.84 M wave accesses; dispatches in pairs (warm-up + timed)"]
for i, d in enumerate(sorted(by)):
    if i % 2 == 1:
        c = by[d]; n = 256 * 8 * 80000.0; p = i // 2
        out.append("%-52s %-38s conflict cycles per access %.2f" % (names[p], timing[p] if p < len(timing) else "?", c.get("SQ_LDS_BANK_CONFLICT", 0) / n))
open("gpurun_out/pmc_ldswr.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
