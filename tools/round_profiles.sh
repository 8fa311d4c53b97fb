#!/bin/bash
# Everything the round's evidence is made of, in one gpurun call.  Usage: COMMIT=<sha> bash tools/round_profiles.sh <tag>
TAG=${1:-r02b}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
# PMC passes first: the bench line below prints their traffic / pipe numbers only when they describe the current kernel sources
bash tools/profile_bench.sh ${TAG} > gpurun_out/${TAG}_prof.log 2>&1
bash tools/pmc_traffic.sh > gpurun_out/${TAG}_pmc_traffic.log 2>&1
bash tools/pmc_mfma.sh > gpurun_out/${TAG}_pmc_mfma.log 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/pmc_mfma.json profiles/
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
# the same step at the reference's default width: per-kernel table, rocprof stats, PMC
python tools/f256_ab.py > gpurun_out/${TAG}_f256_table.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_f256 -o f256 -- python tools/f256_ab.py > /dev/null 2>&1
find gpurun_out/prof_${TAG}_f256 -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_f256_kernel_stats.csv \;
bash tools/pmc_f256.sh > gpurun_out/${TAG}_pmc_f256.txt 2>&1
bash tools/pmc_traffic_f256.sh > gpurun_out/${TAG}_pmc_traffic_f256.log 2>&1
cp gpurun_out/pmc_traffic_f256.txt gpurun_out/${TAG}_pmc_traffic_f256.txt; cp gpurun_out/pmc_traffic_f256.json gpurun_out/${TAG}_pmc_traffic_f256.json
cp gpurun_out/pmc_f256.json gpurun_out/${TAG}_pmc_f256.json
python tools/aggbench.py > gpurun_out/${TAG}_aggbench.txt 2>&1
cp gpurun_out/aggbench.json gpurun_out/${TAG}_aggbench.json
python tools/eval_bench.py > gpurun_out/${TAG}_eval_bench.txt 2>&1
python tools/graph_frame.py > gpurun_out/${TAG}_graph_frame.txt 2>&1
python tools/edge_ab.py > gpurun_out/${TAG}_edge_ab.txt 2>&1
python tools/frame_loop.py > gpurun_out/${TAG}_frame_loop.txt 2>&1
bash tools/frame_trace.sh > gpurun_out/${TAG}_frame_trace.txt 2>&1
tools/ubench/mfma_fill > gpurun_out/${TAG}_mfma_fill.txt 2>&1 || true
# round 4: ordered traces of the batched step and of the one-graph step, savedmodel errors (every printed row)
bash tools/step_trace.sh > /dev/null 2>&1; cp gpurun_out/step_trace.txt gpurun_out/${TAG}_step_trace.txt
bash tools/one_graph_trace.sh > gpurun_out/${TAG}_one_graph_trace.txt 2>&1
python -m pytest tests/test_gpu_savedmodel.py -m gpu -q -s 2>&1 | grep -o '\[[a-z0-9]*/[a-z]*\].*' > gpurun_out/${TAG}_savedmodel_errors.txt
# round 5: small calls eager vs replayed, the edge-table step, list-builder times
python tools/small_calls.py > gpurun_out/${TAG}_small_calls.txt 2>&1
python tools/sort_time.py > gpurun_out/${TAG}_gl_sort.txt 2>&1
NG_BENCH_EDGE_TABLE=1 bash tools/step_trace.sh > /dev/null 2>&1; cp gpurun_out/step_trace.txt gpurun_out/${TAG}_step_trace_edge_table.txt
tail -3 gpurun_out/${TAG}_bench.err; head -c 400 gpurun_out/${TAG}_bench.json; echo; tail -4 gpurun_out/${TAG}_eval_bench.txt
python tools/check_profiles.py > gpurun_out/${TAG}_profile_digests.txt 2>&1; grep -c STALE gpurun_out/${TAG}_profile_digests.txt
