#!/bin/bash
# One gpurun call: the query-stationary nomination kernel (tests, then the batched leg of the headline bench with the glds kernel beside it) and
# the sorted-list HNSW search (tests, 1M x 768 leg).  Usage: gpurun -- bash tools/gpu_session_round3c.sh <tag>
set -u
TAG=${1:-rd3i}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 90 python - <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from reindexer_amd import capi
rng = np.random.default_rng(3)
bad = 0
for d, n in ((768, 5000), (512, 3000), (128, 4000)):
    rows = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((256, d), dtype=np.float32)
    with capi.VectorIndex(1, d, n) as ix:
        ix.upload_rows(0, rows, None)
        bd, br, bc = ix.search_knn(q, 10)
        for i in range(0, 256, 37):
            sd, sr, sc = ix.search_knn(q[i:i + 1], 10)
            bad += int(not (np.array_equal(sr[0], br[i]) and np.array_equal(sd[0].view(np.uint32), bd[i].view(np.uint32))))
print("GEMM PROBE mismatches", bad)
sys.exit(1 if bad else 0)
PY
rc=$?
echo "gemm probe exit $rc"
if [ $rc -ne 0 ]; then export RXGPU_GEMM_QREG=0; echo "query-stationary kernel OFF for the rest of the session"; fi
timeout 900 python -m pytest tests/test_gpu_batched.py -q -m gpu -x > gpurun_out/${TAG}_tests_batched.log 2>&1
tail -6 gpurun_out/${TAG}_tests_batched.log
timeout 900 python -m pytest tests/test_gpu_hnsw_sorted.py tests/test_ann_cache.py tests/test_gpu_hnsw.py tests/test_gpu_sq8.py -q -m gpu > gpurun_out/${TAG}_tests_hnsw.log 2>&1
tail -8 gpurun_out/${TAG}_tests_hnsw.log
for mode in 1 0; do
  RXGPU_GEMM_QREG=$mode timeout 400 python bench.py --no-cpu --hnsw-rows 0 --hybrid-docs 0 --ft-packed-words 0 --steps 5 --warmup 2 --batch-iters 5 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_qreg$mode.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench_qreg$mode.json').read())
b=d.get('batched',{})
print('BATCHED qreg=$mode', d.get('value'), json.dumps({k:b.get(k) for k in ('ms_per_batch','queries_per_sec','rescore_ms','equals_batch1_rows','equals_batch1_dist_bits')}), json.dumps(b.get('roofline'))[:500])
PY
done
for m in l2 cosine; do
  timeout 300 python bench.py --metric $m --no-cpu --hnsw-rows 0 --hybrid-docs 0 --ft-packed-words 0 --steps 3 --warmup 1 --batch-iters 5 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_$m.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench_$m.json').read())
b=d.get('batched',{})
print('BATCHED $m', json.dumps({k:b.get(k) for k in ('ms_per_batch','queries_per_sec','rescore_ms','equals_batch1_rows','equals_batch1_dist_bits')}), json.dumps(b.get('roofline'))[:300])
PY
done
timeout 900 python tools/bench_hnsw.py --rows 1000000 --queries 16384 --build-threads 16 --out gpurun_out/${TAG}_hnsw.json > gpurun_out/${TAG}_hnsw.log 2>&1
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_hnsw.json"))
g = d["gpu"]
print("HNSW", {k: g.get(k) for k in ("queries_per_sec", "queries_per_sec_kernel_only", "kernel_ms_total", "tie_reruns", "tie_rerun_ms", "redo_ms", "map_single_query_latency_ms", "heap_kernel")})
print("roofline", g["roofline"]["frac"], "equal", d.get("equal_to_reference_frac"), "recall", d.get("recall_at_k_vs_exact"))
s = d.get("sq8", {}).get("gpu", {})
print("SQ8", {k: s.get(k) for k in ("queries_per_sec", "queries_per_sec_kernel_only", "tie_reruns", "tie_rerun_ms")}, d.get("sq8", {}).get("equal_to_reference_frac"))
PY
