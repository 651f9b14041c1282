#!/bin/bash
# round 4, session i: BASELINE configs[2] at its true size (10M x 768 cosine HNSW) — hashed visited sets (default above 4.2M nodes) against the
# bitset + memset path on the SAME graph, with the reference engine beside the first run
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
free -g | head -2
timeout 1500 python tools/bench_hnsw.py --rows 10000000 --queries 16384 --build-threads 16 --no-map-legs --no-sq8 --cpu-queries 128 --save-graph /tmp/g10m.npz --out gpurun_out/r4i_hnsw_10m_hash.json > /tmp/h1.log 2>&1; echo "rc=$?"; tail -c 400 /tmp/h1.log
RXGPU_HNSW_VISITED=bitset timeout 600 python tools/bench_hnsw.py --rows 10000000 --queries 16384 --graph /tmp/g10m.npz --gpu-only --no-map-legs --no-sq8 --out gpurun_out/r4i_hnsw_10m_bitset.json > /tmp/h2.log 2>&1; echo "rc=$?"; tail -c 300 /tmp/h2.log
RXGPU_HNSW_PREFETCH=0 timeout 600 python tools/bench_hnsw.py --rows 10000000 --queries 16384 --graph /tmp/g10m.npz --gpu-only --no-map-legs --no-sq8 --out gpurun_out/r4i_hnsw_10m_hash_nopre.json > /tmp/h3.log 2>&1; echo "rc=$?"
python - <<'PY'
import json
for tag in ('hash', 'bitset', 'hash_nopre'):
    try:
        d = json.load(open(f'gpurun_out/r4i_hnsw_10m_{tag}.json')); g = d['gpu']
        print(tag, 'qps', round(g['queries_per_sec']), 'kernel-only', round(g['queries_per_sec_kernel_only']), 'frac', round(g['roofline']['frac'], 3), 'evals', round(g['distance_evals_per_query'], 1),
              'redo', g.get('redo_launches'), g.get('redo_ms'), 'ties', g.get('tie_reruns'), 'equal', d.get('equal_to_reference_frac'), 'recall', d.get('recall_at_k_vs_exact'))
    except Exception as e:
        print(tag, 'failed', repr(e))
PY
