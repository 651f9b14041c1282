#!/bin/bash
# round 4, session l: helper workgroups beside an HNSW batch (overflowing restarts run during the batch) + restart area 256 — tests, the 1M A/B on
# one graph, then BASELINE configs[2] at its true size (10M x 768) with and without them; packed upload in one piece again
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hnsw_visited.py tests/test_gpu_hnsw_sorted.py tests/test_gpu_hnsw.py tests/test_gpu_sq8.py tests/test_gpu_ft_packed.py tests/test_gpu_concurrency.py -q -m gpu -x 2>&1 | tail -5
timeout 300 python tools/bench_ft_packed.py --out gpurun_out/r4l_ft_packed.json > /tmp/pk.log 2>&1; echo "packed rc=$?"; python -c "import json; d=json.load(open('gpurun_out/r4l_ft_packed.json'))['device']; print('packed wall', d['seconds'], 'count', d['kernels']['count_ms'], 'write', d['kernels']['write_ms'])"
B="python tools/bench_hnsw.py --rows 1000000 --queries 16384 --no-map-legs"
timeout 600 $B --build-threads 16 --cpu-queries 64 --save-graph /tmp/g1m.npz --out gpurun_out/r4l_hnsw_1m_helpers.json > /tmp/k1.log 2>&1; echo "rc=$?"
RXGPU_HNSW_HELPER=0 timeout 300 $B --graph /tmp/g1m.npz --gpu-only --no-sq8 --out gpurun_out/r4l_hnsw_1m_nohelpers.json > /tmp/k2.log 2>&1; echo "rc=$?"
RXGPU_HNSW_RESTART_CAND=600 timeout 300 $B --graph /tmp/g1m.npz --gpu-only --no-sq8 --out gpurun_out/r4l_hnsw_1m_helpers_cap600.json > /tmp/k3.log 2>&1; echo "rc=$?"
timeout 300 $B --graph /tmp/g1m.npz --gpu-only --no-sq8 --out gpurun_out/r4l_hnsw_1m_helpers_again.json > /tmp/k4.log 2>&1; echo "rc=$?"
show() {
python - "$@" <<'PY'
import json, sys
for tag in sys.argv[1:]:
    try:
        d = json.load(open(f'gpurun_out/r4l_hnsw_{tag}.json')); g = d['gpu']
        print(tag, 'qps', round(g['queries_per_sec']), 'kernel-only', round(g['queries_per_sec_kernel_only']), 'kernel_ms', round(g['kernel_ms_total'], 3), 'redo', g.get('redo_launches'), round(g.get('redo_ms') or 0, 3),
              'ties', g.get('tie_reruns'), 'equal', d.get('equal_to_reference_frac'), 'sq8', round((d.get('sq8') or {}).get('gpu', {}).get('queries_per_sec') or 0), 'sq8 equal', (d.get('sq8') or {}).get('equal_to_reference_frac'))
    except Exception as e:
        print(tag, 'failed', repr(e))
PY
}
show 1m_helpers 1m_nohelpers 1m_helpers_cap600 1m_helpers_again
B="python tools/bench_hnsw.py --rows 10000000 --queries 16384 --no-map-legs"
timeout 1500 $B --build-threads 16 --cpu-queries 128 --save-graph /tmp/g10m.npz --out gpurun_out/r4l_hnsw_10m_helpers.json > /tmp/h1.log 2>&1; echo "rc=$?"; tail -c 200 /tmp/h1.log
RXGPU_HNSW_HELPER=0 timeout 600 $B --graph /tmp/g10m.npz --gpu-only --out gpurun_out/r4l_hnsw_10m_nohelpers.json > /tmp/h2.log 2>&1; echo "rc=$?"
show 10m_helpers 10m_nohelpers
