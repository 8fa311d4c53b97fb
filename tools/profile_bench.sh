#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the default bench workload.
# Usage: tools/profile_bench.sh <tag>     -> gpurun_out/prof_<tag>/..., summary copied to profiles/<tag>_kernel_stats.csv
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_${TAG}
rm -rf "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o "$TAG" -- \
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-extras > gpurun_out/bench_${TAG}.log 2>&1
find "$OUT" -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
head -25 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
grep '"metric"' gpurun_out/bench_${TAG}.log | cut -c1-250
