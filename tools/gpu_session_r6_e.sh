#!/bin/bash
# Round 6: tests of the mailbox (both classes) and the sharded Map, then single-query variants on ONE 10M x 768 graph: mailbox / launch, rows
# on 4 KB boundaries (RXGPU_ROW_ALIGN), with python threads for the concurrency figure.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hnsw_server.py tests/test_gpu_sharded_hnsw.py tests/test_gpu_preflight.py -x -q -m gpu > gpurun_out/rd6e_tests.log 2>&1; tail -4 gpurun_out/rd6e_tests.log
timeout 1800 python tools/bench_hnsw_single.py --rows ${1:-10000000} --only mailbox_plain,mailbox_rows_page_aligned,launch_plain --out gpurun_out/rd6e_single_10m.json > gpurun_out/rd6e_single_10m.log 2>&1
grep -E "single-query|hops" gpurun_out/rd6e_single_10m.log | cut -c1-420
