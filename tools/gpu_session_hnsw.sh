#!/bin/bash
# One gpurun call: SQ8 + HNSW tests, the 1M x 768 HNSW leg (float + SQ8, reference engines beside), then rocprofv3 passes of the search kernel
# on the same graph (kernel trace; FETCH_SIZE and WRITE_SIZE in their own passes).  Usage: gpurun -- bash tools/gpu_session_hnsw.sh <tag> [rows]
set -u
TAG=${1:-r2k}
ROWS=${2:-1000000}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sq8.py tests/test_gpu_hnsw.py -x -q > gpurun_out/${TAG}_tests.log 2>&1
tail -5 gpurun_out/${TAG}_tests.log
timeout 1500 python tools/bench_hnsw.py --rows $ROWS --queries 16384 --build-threads 16 --save-graph /tmp/g.npz --out gpurun_out/${TAG}_hnsw.json \
	> gpurun_out/${TAG}_hnsw.log 2>&1
tail -c 1500 gpurun_out/${TAG}_hnsw.log
CMD="python $R/tools/bench_hnsw.py --rows $ROWS --queries 16384 --graph /tmp/g.npz --gpu-only"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/trace -o $TAG -- $CMD > /tmp/prof_trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof/pmc_fetch -o $TAG -- $CMD > /tmp/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof/pmc_write -o $TAG -- $CMD > /tmp/prof_write.log 2>&1
cd "$R"
tail -3 /tmp/prof_trace.log /tmp/prof_fetch.log
python tools/summarize_prof2.py /tmp/prof ${TAG}_hnsw hnsw_search_kernel "rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE -- $CMD" | head -60
