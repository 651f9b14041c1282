#!/bin/bash
# rocprofv3 evidence for the HNSW search kernels as shipped: one 1M x 768 graph built on the box, then kernel trace, FETCH_SIZE and WRITE_SIZE
# passes (each its own run) of the same 16 384-query search.  Usage: gpurun -- bash tools/gpu_session_hnsw_prof.sh <tag>
set -u
TAG=${1:-rd3s}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 600 python tools/bench_hnsw.py --rows 1000000 --queries 16384 --build-threads 16 --save-graph /tmp/g.npz --gpu-only --out gpurun_out/${TAG}_hnsw_gpu_only.json > /tmp/first.log 2>&1
tail -c 300 /tmp/first.log
CMD="python $R/tools/bench_hnsw.py --rows 1000000 --queries 16384 --graph /tmp/g.npz --gpu-only"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/trace -o $TAG -- $CMD > /tmp/prof_trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof/pmc_fetch -o $TAG -- $CMD > /tmp/prof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof/pmc_write -o $TAG -- $CMD > /tmp/prof_write.log 2>&1
cd "$R"
tail -2 /tmp/prof_trace.log
python tools/summarize_prof2.py /tmp/prof ${TAG}_hnsw hnsw_search_kernel "rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE -- $CMD" | head -50
cp profiles/${TAG}* gpurun_out/ 2>/dev/null
