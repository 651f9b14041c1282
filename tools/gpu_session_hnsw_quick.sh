#!/bin/bash
# HNSW search kernel iteration loop: parity tests, then the 1M x 768 graph built once and searched (float + SQ8 vs the reference engines).
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo} && export TMPDIR=/tmp && mkdir -p gpurun_out
TAG=${1:-hq}
timeout 600 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_sq8.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -4
timeout 900 python tools/bench_hnsw.py --rows 1000000 --queries 16384 --build-threads 16 --no-map-legs --out gpurun_out/${TAG}_hnsw_1m.json > /tmp/b.log 2>&1
python - <<PY
import json
d = json.load(open('gpurun_out/${TAG}_hnsw_1m.json')); g = d['gpu']
print('float', round(g['queries_per_sec_kernel_only']), 'frac', round(g['roofline']['frac'], 3), 'equal', d.get('equal_to_reference_frac'), 'recall', d['recall_at_k_vs_exact'])
s = d.get('sq8')
if s: print('sq8', round(s['gpu']['queries_per_sec_kernel_only']), 'equal', s['equal_to_reference_frac'])
PY
