#!/bin/bash
# round 4, session e: hashed visited sets (tests, 1M x 768 A/B against the bitset), the load-order fix and the resident sessions
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ft_seam.py tests/test_gpu_hybrid.py tests/test_gpu_hybrid_fuse.py tests/test_gpu_concurrency.py tests/test_gpu_hnsw_visited.py tests/test_gpu_hnsw_sorted.py tests/test_gpu_hnsw.py tests/test_gpu_sq8.py -q 2>&1 | tail -12 > gpurun_out/r4e_tests.txt
cat gpurun_out/r4e_tests.txt
timeout 900 python tools/bench_hnsw.py --rows 1000000 --queries 16384 --build-threads 16 --save-graph /tmp/g1m.npz --out gpurun_out/r4e_hnsw_1m_hash.json > /tmp/b1.log 2>&1; tail -2 /tmp/b1.log | cut -c1-300
RXGPU_HNSW_VISITED=bitset timeout 600 python tools/bench_hnsw.py --rows 1000000 --queries 16384 --graph /tmp/g1m.npz --gpu-only --no-map-legs --out gpurun_out/r4e_hnsw_1m_bitset.json > /tmp/b2.log 2>&1; tail -2 /tmp/b2.log | cut -c1-300
python - <<'PY'
import json
for tag in ('hash', 'bitset'):
    try:
        d = json.load(open(f'gpurun_out/r4e_hnsw_1m_{tag}.json')); g = d['gpu']
        print(tag, 'qps', round(g['queries_per_sec']), 'kernel-only', round(g['queries_per_sec_kernel_only']), 'frac', round(g['roofline']['frac'], 3), 'evals', round(g['distance_evals_per_query'], 1),
              'redo', g.get('redo_launches'), 'ties', g.get('tie_reruns'), 'equal', d.get('equal_to_reference_frac'), 'lat', g.get('map_single_query_latency_ms'), 'sq8', round(d['sq8']['gpu']['queries_per_sec']) if d.get('sq8') else None)
    except Exception as e:
        print(tag, 'failed', repr(e))
PY
