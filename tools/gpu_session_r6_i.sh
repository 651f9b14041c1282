#!/bin/bash
# Round 6: the link block of a candidate that an insertion puts at the head of the line is requested at once (latency forms): parity + fuzz,
# then one query at a time at 1M (and, with an argument, the Map legs at 10M).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_hnsw_server.py tests/test_gpu_hnsw.py tests/test_gpu_hnsw_sorted.py tests/test_gpu_hnsw_visited.py tests/test_gpu_sharded_hnsw.py tests/test_gpu_concurrency.py tests/test_gpu_knn_seam.py tests/test_gpu_cpp_engine.py tests/test_gpu_sq8.py -x -q -m gpu > gpurun_out/rd6i_tests.log 2>&1; tail -3 gpurun_out/rd6i_tests.log | cut -c1-200
grep -q " passed" gpurun_out/rd6i_tests.log && ! grep -q "failed" gpurun_out/rd6i_tests.log || exit 1
timeout 300 python tools/fuzz_hnsw.py --seconds 40 > gpurun_out/rd6i_fuzz.log 2>&1; tail -1 gpurun_out/rd6i_fuzz.log
grep -q "fuzz ok" gpurun_out/rd6i_fuzz.log || exit 1
timeout 900 python tools/bench_hnsw_single.py --rows 1000000 --only mailbox_plain,launch_plain --out gpurun_out/rd6i_single_1m.json > gpurun_out/rd6i_single_1m.log 2>&1; grep -E "single-query|hops" gpurun_out/rd6i_single_1m.log | cut -c1-200
if [ "${1:-0}" = "10m" ]; then
timeout 1500 python tools/bench_hnsw.py --rows 10000000 --queries 4096 --cpu-queries 256 --recall-queries 1000 --no-sq8 --map-threads 1,16,64,256 --map-per-thread 64 \
  --out gpurun_out/rd6i_hnsw_10m.json > gpurun_out/rd6i_hnsw_10m.log 2>&1
python - <<'PY'
import json
d = json.load(open('gpurun_out/rd6i_hnsw_10m.json')); g = d['gpu']
print('10M single ms', round(g.get('map_single_query_latency_ms', 0), 3), 'cpu 1/all', round(d['cpu_baseline']['value']), round(d['cpu_baseline']['all_cores']['value']), 'batch', round(g['queries_per_sec']))
for t in g.get('map_threads', []):
    print('  T', t['threads'], round(t['queries_per_sec']), 'posted', t.get('posted'), 'ms on device', round(t.get('posted_ms_on_device') or 0, 3), 'ms at caller', round(t.get('posted_ms_at_caller') or 0, 3))
PY
fi
