#!/bin/bash
# Round 6, HNSW at the planner's concurrency: one gpurun call = HNSW parity tests (the small launches of the tests run the team kernel),
# then the 1M x 768 leg of tools/bench_hnsw.py three times over ONE saved graph — default (four wavefronts per search for small launches,
# zero-copy queries / results), RXGPU_HNSW_TEAM=1 (one wavefront per search), RXGPU_HNSW_ZERO_COPY=0 (the copies).
#   gpurun -- bash tools/gpu_session_r6_hnsw.sh <tag> [tests] [rows]
set -u
TAG=${1:-rd6h}
TESTS=${2:-1}
ROWS=${3:-1000000}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
if [ "$TESTS" = "1" ]; then
  timeout 1500 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_hnsw_sorted.py tests/test_gpu_hnsw_visited.py tests/test_gpu_sq8.py tests/test_gpu_sharded_hnsw.py \
    tests/test_gpu_sharded_map.py tests/test_gpu_concurrency.py tests/test_gpu_ft_sharded.py -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1
  grep -E "passed|failed|error" gpurun_out/${TAG}_tests.log | tail -3
fi
COMMON="--rows $ROWS --queries 4096 --cpu-queries 256 --recall-queries 1000 --no-sq8 --map-threads 1,4,16,64,256 --map-per-thread 64"
timeout 1500 python tools/bench_hnsw.py $COMMON --save-graph /tmp/g.npz --out gpurun_out/${TAG}_hnsw_default.json > gpurun_out/${TAG}_hnsw_default.log 2>&1
# (the Map legs need the Map: the saved graph only skips nothing here — every variant rebuilds; kept for the rocprof passes)
for V in "TEAM=1" "ZERO_COPY=0"; do
  env RXGPU_HNSW_$V timeout 1500 python tools/bench_hnsw.py $COMMON --out gpurun_out/${TAG}_hnsw_${V%%=*}.json > gpurun_out/${TAG}_hnsw_${V%%=*}.log 2>&1
done
python - <<PY
import json, glob
for f in sorted(glob.glob('gpurun_out/${TAG}_hnsw_*.json')):
    d = json.load(open(f))
    g = d['gpu']
    print(f.split('_hnsw_')[1][:-5], 'batch q/s', round(g['queries_per_sec']), 'single ms', round(g.get('map_single_query_latency_ms', 0), 3),
          'map', [(t['threads'], round(t['queries_per_sec']), round(t['avg_batch'], 1)) for t in g.get('map_threads', [])],
          'cpu 1/all', round(d.get('cpu_baseline', {}).get('value', 0)), round(d.get('cpu_baseline', {}).get('all_cores', {}).get('value', 0)),
          'equal', d.get('equal_to_reference_frac'), 'recall', round(d['recall_at_k_vs_exact'], 3), 'build s', round(d['build']['seconds']))
PY
