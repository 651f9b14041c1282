#!/bin/bash
# Round 6: the resident HNSW search kernel.  Tests of the mailbox path + the HNSW suites that now run through it, then the 1M x 768 Map legs.
#   gpurun -- bash tools/gpu_session_r6_server.sh <tag> [rows] [tests]
set -u
TAG=${1:-rd6s}
ROWS=${2:-1000000}
TESTS=${3:-1}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
if [ "$TESTS" = "1" ]; then
  timeout 900 python -m pytest tests/test_gpu_hnsw_server.py -x -q -m gpu > gpurun_out/${TAG}_server_tests.log 2>&1
  tail -15 gpurun_out/${TAG}_server_tests.log
  timeout 1500 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_hnsw_sorted.py tests/test_gpu_hnsw_visited.py tests/test_gpu_sq8.py tests/test_gpu_sharded_hnsw.py \
    tests/test_gpu_sharded_map.py tests/test_gpu_concurrency.py tests/test_gpu_knn_seam.py tests/test_gpu_cpp_engine.py -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1
  grep -E "passed|failed|error" gpurun_out/${TAG}_tests.log | tail -3
fi
if [ "$ROWS" != "0" ]; then
COMMON="--rows $ROWS --queries 4096 --cpu-queries 256 --recall-queries 1000 --no-sq8 --map-threads 1,4,16,64,256 --map-per-thread 64"
timeout 1500 python tools/bench_hnsw.py $COMMON --out gpurun_out/${TAG}_hnsw.json > gpurun_out/${TAG}_hnsw.log 2>&1
python - <<PY
import json
d = json.load(open('gpurun_out/${TAG}_hnsw.json'))
g = d['gpu']
print('batch q/s', round(g['queries_per_sec']), 'single ms', round(g.get('map_single_query_latency_ms', 0), 3),
      'map', [(t['threads'], round(t['queries_per_sec']), t.get('posted')) for t in g.get('map_threads', [])],
      'cpu 1/all', round(d.get('cpu_baseline', {}).get('value', 0)), round(d.get('cpu_baseline', {}).get('all_cores', {}).get('value', 0)),
      'equal', d.get('equal_to_reference_frac'), 'recall', round(d['recall_at_k_vs_exact'], 3))
PY
fi
