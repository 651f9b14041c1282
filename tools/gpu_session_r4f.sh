#!/bin/bash
# round 4, session f: link-block prefetch + hashed visited sets A/B at 1M x 768; FT kernels at 5 / 4 workgroups per CU
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_hnsw_visited.py tests/test_gpu_hnsw_sorted.py tests/test_gpu_hnsw.py tests/test_gpu_sq8.py tests/test_gpu_ft_batch.py tests/test_gpu_bm25.py tests/test_gpu_ft_terms.py tests/test_gpu_ft_phrases.py tests/test_gpu_ft_synonyms.py tests/test_gpu_fuzz.py -q 2>&1 | tail -8 > gpurun_out/r4f_tests.txt
cat gpurun_out/r4f_tests.txt
timeout 900 python tools/bench_hnsw.py --rows 1000000 --queries 16384 --build-threads 16 --save-graph /tmp/g1m.npz --gpu-only --no-map-legs --out gpurun_out/r4f_hnsw_hash_pre.json > /tmp/b1.log 2>&1
RXGPU_HNSW_VISITED=bitset timeout 600 python tools/bench_hnsw.py --rows 1000000 --queries 16384 --graph /tmp/g1m.npz --gpu-only --no-map-legs --out gpurun_out/r4f_hnsw_bitset_pre.json > /tmp/b2.log 2>&1
RXGPU_HNSW_VISITED=bitset RXGPU_HNSW_PREFETCH=0 timeout 600 python tools/bench_hnsw.py --rows 1000000 --queries 16384 --graph /tmp/g1m.npz --gpu-only --no-map-legs --out gpurun_out/r4f_hnsw_bitset_nopre.json > /tmp/b3.log 2>&1
RXGPU_HNSW_PREFETCH=0 timeout 600 python tools/bench_hnsw.py --rows 1000000 --queries 16384 --graph /tmp/g1m.npz --gpu-only --no-map-legs --out gpurun_out/r4f_hnsw_hash_nopre.json > /tmp/b4.log 2>&1
python - <<'PY'
import json
for tag in ('hash_pre', 'bitset_pre', 'bitset_nopre', 'hash_nopre'):
    try:
        d = json.load(open(f'gpurun_out/r4f_hnsw_{tag}.json')); g = d['gpu']
        print(tag, 'qps', round(g['queries_per_sec']), 'kernel-only', round(g['queries_per_sec_kernel_only']), 'frac', round(g['roofline']['frac'], 3), 'evals', round(g['distance_evals_per_query'], 1),
              'redo', g.get('redo_launches'), 'ties', g.get('tie_reruns'), 'sq8', round(d['sq8']['gpu']['queries_per_sec_kernel_only']) if d.get('sq8') else None)
    except Exception as e:
        print(tag, 'failed', repr(e))
PY
cd /tmp
for shape in "sparse 1,1 0.04,0.01 64" "dense 1,1,1 0.2,0.05,0.01 16"; do
  set -- $shape
  rm -rf /tmp/prof_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o p -- python $R/tools/bench_bm25.py --ops $2 --fracs $3 --queries 256 --batch $4 --batch-only > $R/gpurun_out/r4f_prof_$1.log 2>&1
  grep batch_only $R/gpurun_out/r4f_prof_$1.log
  f=$(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/r4f_kernel_stats_$1.csv
  head -8 "$f" | cut -c1-120
done
