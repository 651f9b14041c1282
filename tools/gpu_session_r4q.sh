#!/bin/bash
# round 4, session q: large HNSW batches in two halves (tests), the batched brute-force tests after the kernel clean-up, the 1M bench leg with and without
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_hnsw_visited.py tests/test_gpu_hnsw.py tests/test_gpu_sq8.py tests/test_gpu_batched.py -q -m gpu -x 2>&1 | tail -3
B="python tools/bench_hnsw.py --rows 1000000 --queries 16384 --no-map-legs"
timeout 300 $B --build-threads 16 --cpu-queries 32 --save-graph /tmp/g1m.npz --out gpurun_out/r4q_hnsw_1m_split.json > /tmp/k1.log 2>&1; echo "rc=$?"
RXGPU_HNSW_SPLIT_UPLOAD=0 timeout 200 $B --graph /tmp/g1m.npz --gpu-only --out gpurun_out/r4q_hnsw_1m_nosplit.json > /tmp/k2.log 2>&1; echo "rc=$?"
timeout 200 $B --graph /tmp/g1m.npz --gpu-only --out gpurun_out/r4q_hnsw_1m_split_again.json > /tmp/k3.log 2>&1; echo "rc=$?"
python - <<'PY'
import json
for tag in ('split', 'nosplit', 'split_again'):
    try:
        d = json.load(open(f'gpurun_out/r4q_hnsw_1m_{tag}.json')); g = d['gpu']; s = (d.get('sq8') or {}).get('gpu', {})
        print(tag, 'qps', round(g['queries_per_sec']), 'kernel-only', round(g['queries_per_sec_kernel_only']), 'kernel_ms', round(g['kernel_ms_total'], 3), 'equal', d.get('equal_to_reference_frac'),
              'sq8', round(s.get('queries_per_sec') or 0), round(s.get('queries_per_sec_kernel_only') or 0))
    except Exception as e:
        print(tag, 'failed', repr(e))
PY
