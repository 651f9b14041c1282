#!/bin/bash
# One gpurun call for the sorted-list HNSW search: a guarded probe first (small graph, list vs heaps, under a short timeout — a kernel that
# hangs must not take the session with it), then the test files, the 1M x 768 leg with the heap kernel beside it, and the rocprofv3 passes.
# Usage: gpurun -- bash tools/gpu_session_hnsw_sorted.sh <tag> [rows]
set -u
TAG=${1:-rd3h}
ROWS=${2:-1000000}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
probe() {
timeout 120 python - <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from reindexer_amd import hostapi
rng = np.random.default_rng(5)
n, d = 4000, 128
rows = rng.standard_normal((n, d), dtype=np.float32)
m = hostapi.GpuHnswMap(0, d, n, M=16, ef_construction=100)
m.add(rows, np.arange(n, dtype=np.uint64))
mode = os.environ.get("RXGPU_HNSW_SORTED", "1")
bad = 0
for qi in range(32):
    q = rng.standard_normal(d, dtype=np.float32)
    for k, ef in ((10, 128), (10, 16), (100, 200)):
        os.environ["RXGPU_HNSW_SORTED"] = mode
        a = m.search_knn(q, k, ef)
        os.environ["RXGPU_HNSW_SORTED"] = "0"
        b = m.search_knn(q, k, ef)
        bad += int(not (np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))))
print("PROBE mode", mode, "mismatches", bad, "tie reruns", m.tie_reruns())
sys.exit(1 if bad else 0)
PY
}
probe; rc=$?
echo "probe (DPP shift) exit $rc"
if [ $rc -ne 0 ]; then
  export RXGPU_HNSW_SORTED=2
  probe; rc=$?
  echo "probe (bpermute shift) exit $rc"
  if [ $rc -ne 0 ]; then echo "sorted list broken in both forms: heaps only for the rest"; export RXGPU_HNSW_SORTED=0; fi
fi
echo "RXGPU_HNSW_SORTED=${RXGPU_HNSW_SORTED:-unset}" | tee gpurun_out/${TAG}_mode.txt
timeout 1200 python -m pytest tests/test_gpu_hnsw_sorted.py tests/test_ann_cache.py tests/test_gpu_hnsw.py tests/test_gpu_sq8.py -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1
tail -15 gpurun_out/${TAG}_tests.log
timeout 1200 python tools/bench_hnsw.py --rows $ROWS --queries 16384 --build-threads 16 --save-graph /tmp/g.npz --out gpurun_out/${TAG}_hnsw.json \
	> gpurun_out/${TAG}_hnsw.log 2>&1
tail -c 600 gpurun_out/${TAG}_hnsw.log
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_hnsw.json"))
g = d["gpu"]
print("HNSW", {k: g.get(k) for k in ("queries_per_sec", "queries_per_sec_kernel_only", "kernel_ms_total", "tie_reruns", "tie_rerun_ms", "redo_ms", "map_single_query_latency_ms", "heap_kernel")})
print("roofline", g["roofline"]["frac"], "equal", d.get("equal_to_reference_frac"), "recall", d.get("recall_at_k_vs_exact"))
s = d.get("sq8", {}).get("gpu", {})
print("SQ8", {k: s.get(k) for k in ("queries_per_sec", "queries_per_sec_kernel_only", "tie_reruns", "tie_rerun_ms")}, d.get("sq8", {}).get("equal_to_reference_frac"))
print("stream", d.get("streaming_session"))
PY
CMD="python $R/tools/bench_hnsw.py --rows $ROWS --queries 16384 --graph /tmp/g.npz --gpu-only"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/trace -o $TAG -- $CMD > /tmp/prof_trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof/pmc_fetch -o $TAG -- $CMD > /tmp/prof_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof/pmc_write -o $TAG -- $CMD > /tmp/prof_write.log 2>&1
cd "$R"
tail -2 /tmp/prof_trace.log /tmp/prof_fetch.log
python tools/summarize_prof2.py /tmp/prof ${TAG}_hnsw hnsw_search_kernel "rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE -- $CMD" | head -40
cp profiles/${TAG}* gpurun_out/ 2>/dev/null
