#!/bin/bash
# Round 6: parity tests of everything HNSW (mailbox, look-ahead, link blocks that come along, sharded SQ8 / streaming), fuzz, the single-query
# variants at 1M, then configs[2] at 10M with the Map legs.  Stops at the first failing stage.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_hnsw_server.py tests/test_gpu_hnsw.py tests/test_gpu_hnsw_sorted.py tests/test_gpu_hnsw_visited.py tests/test_gpu_sharded_hnsw.py tests/test_gpu_sharded_map.py \
  tests/test_gpu_sq8.py tests/test_gpu_concurrency.py tests/test_gpu_knn_seam.py tests/test_gpu_cpp_engine.py -x -q -m gpu > gpurun_out/rd6d_tests.log 2>&1
tail -4 gpurun_out/rd6d_tests.log
grep -q " passed" gpurun_out/rd6d_tests.log && ! grep -q "failed" gpurun_out/rd6d_tests.log || exit 1
timeout 300 python tools/fuzz_hnsw.py --seconds 45 > gpurun_out/rd6d_fuzz.log 2>&1; tail -2 gpurun_out/rd6d_fuzz.log
grep -q "fuzz ok" gpurun_out/rd6d_fuzz.log || exit 1
timeout 900 python tools/bench_hnsw_single.py --rows 1000000 --out gpurun_out/rd6d_single_1m.json > gpurun_out/rd6d_single_1m.log 2>&1; grep -E "single-query|trips" gpurun_out/rd6d_single_1m.log | cut -c1-300
timeout 1500 python tools/bench_hnsw.py --rows 10000000 --queries 4096 --cpu-queries 256 --recall-queries 1000 --no-sq8 --map-threads 1,16,64,256 --map-per-thread 64 \
  --out gpurun_out/rd6d_hnsw_10m.json > gpurun_out/rd6d_hnsw_10m.log 2>&1
python - <<'PY'
import json
d = json.load(open('gpurun_out/rd6d_hnsw_10m.json')); g = d['gpu']
print('10M batch q/s', round(g['queries_per_sec']), 'kernel-only', round(g['queries_per_sec_kernel_only']), 'single ms', round(g.get('map_single_query_latency_ms', 0), 3),
      'map', [(t['threads'], round(t['queries_per_sec']), t.get('posted')) for t in g.get('map_threads', [])],
      'cpu 1/all', round(d['cpu_baseline']['value']), round(d['cpu_baseline']['all_cores']['value']), 'equal', d.get('equal_to_reference_frac'))
PY
