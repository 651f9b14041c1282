#!/bin/bash
# round 4, session n: the pruned scan's 16-rows-per-wavefront form for the tile-blocked shadow (parity + timing against the row-major shadow),
# and the HNSW suites once more at the closing commit (helper queue reset ordered before the batch)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pruned.py tests/test_gpu_hnsw_visited.py tests/test_gpu_hnsw_sorted.py tests/test_gpu_hnsw.py tests/test_gpu_sq8.py -q -m gpu -x 2>&1 | tail -4
timeout 600 python tools/bench_gemm_ab.py --metrics ip,l2,cosine --rounds 1 --iters 2 --pruned --modes split_ring_blocked_shadow,split_ring_rowmajor_shadow --out gpurun_out/r4n_pruned_ab.json 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for line in sys.stdin:
    try:
        m, rest = line.split(' ', 1); d = json.loads(rest)
        print(m, 'pruned blocked', round(d['pruned_scan_ms_blocked'], 4), 'rowmajor', round(d['pruned_scan_ms_rowmajor'], 4), 'gemm blocked', round(d['split_ring_blocked_shadow']['best_ms'], 3), 'rowmajor', round(d['split_ring_rowmajor_shadow']['best_ms'], 3))
    except Exception as e:
        print(line[:200])
"
