#!/bin/bash
# Round 6: 256 mailbox slots by default: server tests, then the Map legs at 1M (and 10M with an argument).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_hnsw_server.py tests/test_gpu_concurrency.py tests/test_gpu_hybrid.py -x -q -m gpu > gpurun_out/rd6j_tests.log 2>&1; tail -2 gpurun_out/rd6j_tests.log | cut -c1-200
for ROWS in 1000000 ${1:-}; do
timeout 1500 python tools/bench_hnsw.py --rows $ROWS --queries 4096 --cpu-queries 256 --recall-queries 1000 --no-sq8 --map-threads 16,64,256,512 --map-per-thread 64 \
  --out gpurun_out/rd6j_hnsw_$ROWS.json > gpurun_out/rd6j_hnsw_$ROWS.log 2>&1
python - <<PY
import json
d = json.load(open('gpurun_out/rd6j_hnsw_$ROWS.json')); g = d['gpu']
print($ROWS, 'single ms', round(g.get('map_single_query_latency_ms', 0), 3), 'cpu 1/all', round(d['cpu_baseline']['value']), round(d['cpu_baseline']['all_cores']['value']))
for t in g.get('map_threads', []):
    print('  T', t['threads'], round(t['queries_per_sec']), 'posted', t.get('posted'), 'ms on device', round(t.get('posted_ms_on_device') or 0, 3), 'ms at caller', round(t.get('posted_ms_at_caller') or 0, 3))
PY
done
