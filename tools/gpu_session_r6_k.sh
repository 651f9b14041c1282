#!/bin/bash
# Round 6, last session: the ANN disk cache of the HNSW Map over a device list (new tests) + the ANN cache suite, then the headline loop
# alone (roofline.traffic now names the whole-corpus launches of the committed PMC passes).  Usage: gpurun -- bash tools/gpu_session_r6_k.sh
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_sharded_hnsw.py tests/test_ann_cache.py tests/test_gpu_knn_seam.py -q -m gpu -x > gpurun_out/rd6k_tests.log 2>&1
tail -15 gpurun_out/rd6k_tests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --batch 0 --hnsw-rows 0 --hybrid-docs 0 --ft-packed-words 0 \
  --full-json gpurun_out/rd6k_bench_full.json > gpurun_out/rd6k_bench_line.json 2> gpurun_out/rd6k_bench.err
tail -c 300 gpurun_out/rd6k_bench.err; head -c 1800 gpurun_out/rd6k_bench_line.json
