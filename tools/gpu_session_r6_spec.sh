#!/bin/bash
# Round 6: the look-ahead (speculative) team search: parity tests, then variant-against-variant latency at 1M x 768.
set -u
TAG=${1:-rd6sp}; ROWS=${2:-1000000}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_hnsw_server.py tests/test_gpu_hnsw.py tests/test_gpu_hnsw_sorted.py tests/test_gpu_hnsw_visited.py tests/test_gpu_sharded_map.py tests/test_gpu_concurrency.py \
  tests/test_gpu_knn_seam.py tests/test_gpu_cpp_engine.py -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; tail -3 gpurun_out/${TAG}_tests.log
timeout 900 python tools/fuzz_hnsw.py > gpurun_out/${TAG}_fuzz.log 2>&1; tail -3 gpurun_out/${TAG}_fuzz.log
timeout 900 python tools/bench_hnsw_single.py --rows $ROWS --out gpurun_out/${TAG}_single.json > gpurun_out/${TAG}_single.log 2>&1; grep -E "single-query|hops|trips" gpurun_out/${TAG}_single.log | cut -c1-400
timeout 900 python tools/bench_hnsw.py --rows $ROWS --queries 4096 --cpu-queries 256 --recall-queries 1000 --no-sq8 --map-threads 1,16,64,256 --map-per-thread 64 --out gpurun_out/${TAG}_hnsw.json > gpurun_out/${TAG}_hnsw.log 2>&1
python - <<PY
import json
d = json.load(open('gpurun_out/${TAG}_hnsw.json')); g = d['gpu']
print('batch q/s', round(g['queries_per_sec']), 'kernel-only', round(g['queries_per_sec_kernel_only']), 'single ms', round(g.get('map_single_query_latency_ms', 0), 3),
      'map', [(t['threads'], round(t['queries_per_sec']), t.get('posted')) for t in g.get('map_threads', [])],
      'cpu 1/all', round(d['cpu_baseline']['value']), round(d['cpu_baseline']['all_cores']['value']), 'equal', d.get('equal_to_reference_frac'))
PY
