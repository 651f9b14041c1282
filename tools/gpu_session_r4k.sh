#!/bin/bash
# round 4, session k: the split-ring bf16 nomination kernel — parity tests of both kernels, then the A/B on the headline corpus
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_ft_packed.py -q -m gpu -x 2>&1 | tail -4
timeout 300 python tools/bench_ft_packed.py --out gpurun_out/r4k_ft_packed.json > /tmp/pk.log 2>&1; echo "packed rc=$?"; python -c "import json; d=json.load(open('gpurun_out/r4k_ft_packed.json'))['device']; print('packed wall', d['seconds'], 'count', d['kernels']['count_ms'], 'write', d['kernels']['write_ms'])"
timeout 900 python tools/bench_gemm_ab.py --out gpurun_out/r4k_gemm_ab.json 2>&1 | tail -8
timeout 300 python tools/bench_gemm_ab.py --batch 128 --metrics ip --out gpurun_out/r4k_gemm_ab_b128.json 2>&1 | tail -3
# HNSW 1M: heap area of an in-kernel restart (LDS per workgroup decides how many searches a CU holds: 600 entries -> 15, 320 -> 20) now that an
# overflowing restart costs an LDS re-run instead of a global-heap launch; the bitset memset beside the query upload
B="python tools/bench_hnsw.py --rows 1000000 --queries 16384 --no-map-legs --no-sq8"
timeout 600 $B --build-threads 16 --cpu-queries 64 --save-graph /tmp/g1m.npz --out gpurun_out/r4k_hnsw_1m_cap600.json > /tmp/k1.log 2>&1; echo "rc=$?"
for C in 320 256 448; do
  RXGPU_HNSW_RESTART_CAND=$C timeout 300 $B --graph /tmp/g1m.npz --gpu-only --out gpurun_out/r4k_hnsw_1m_cap$C.json > /tmp/k2.log 2>&1; echo "rc=$?"
done
timeout 300 $B --graph /tmp/g1m.npz --gpu-only --out gpurun_out/r4k_hnsw_1m_cap600_again.json > /tmp/k3.log 2>&1; echo "rc=$?"
python - <<'PY'
import json
for tag in ('cap600', 'cap320', 'cap256', 'cap448', 'cap600_again'):
    try:
        d = json.load(open(f'gpurun_out/r4k_hnsw_1m_{tag}.json')); g = d['gpu']
        print(tag, 'qps', round(g['queries_per_sec']), 'kernel-only', round(g['queries_per_sec_kernel_only']), 'kernel_ms', round(g['kernel_ms_total'], 3), 'redo', g.get('redo_launches'), g.get('redo_ms'),
              'ties', g.get('tie_reruns'), 'equal', d.get('equal_to_reference_frac'))
    except Exception as e:
        print(tag, 'failed', repr(e))
PY
