"""Do the committed PMC files (profiles/pmc_traffic.json, pmc_mfma.json) describe the kernel sources at HEAD?  Prints one line per
kernel of bench.KERNEL_SOURCES; exit code 1 when any profiled kernel's source digest differs (tools/round_profiles.sh runs this
last; the round's evidence must be regenerated after the last commit that touches csrc/)."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
bad = 0
for fn in ("pmc_traffic.json", "pmc_mfma.json"):
    path = os.path.join(R, "profiles", fn)
    if not os.path.exists(path):
        print(fn, "MISSING"); bad += 1; continue
    meta = json.load(open(path)).get("_meta", {})
    dg = meta.get("source_digest", {})
    for k in bench.KERNEL_SOURCES:
        ok = dg.get(k) == bench.source_digest(k)
        print(f"{fn:18s} {k:18s} {'ok' if ok else 'STALE'}   (collected at {meta.get('commit', '?')})")
        bad += 0 if ok else 1
sys.exit(1 if bad else 0)
