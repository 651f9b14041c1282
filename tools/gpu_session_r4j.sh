#!/bin/bash
# round 4, session j: (1) tests of the changes since session i (LDS visited set, LDS re-run tier, packed pipeline); (2) packed leg; (3) single-query
# HNSW latency A/B at 1M; (4) 10M x 768 HNSW on one graph: default (hash set), hash-set sizes, bitset, SQ8, and the FETCH / WRITE counters of the default
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hnsw_visited.py tests/test_gpu_hnsw_sorted.py tests/test_gpu_hnsw.py tests/test_gpu_ft_packed.py tests/test_gpu_sq8.py -q -m gpu -x 2>&1 | tail -5
timeout 300 python tools/bench_ft_packed.py --out gpurun_out/r4j_ft_packed.json > /tmp/pk.log 2>&1; echo "packed rc=$?"; tail -c 600 /tmp/pk.log
timeout 600 python tools/probe_hnsw_latency.py --rows 1000000 --threads 16 --ab > gpurun_out/r4j_hnsw_latency_1m.txt 2>/tmp/lat.err; echo "latency rc=$?"; cat gpurun_out/r4j_hnsw_latency_1m.txt; tail -3 /tmp/lat.err
B="python tools/bench_hnsw.py --rows 10000000 --queries 16384 --no-map-legs"
timeout 1500 $B --build-threads 16 --cpu-queries 128 --save-graph /tmp/g10m.npz --out gpurun_out/r4j_hnsw_10m.json > /tmp/h1.log 2>&1; echo "rc=$?"; tail -c 300 /tmp/h1.log
for L in 14 15 16; do
  RXGPU_HNSW_VISITED=hash RXGPU_HNSW_VISITED_LOG2=$L timeout 600 $B --graph /tmp/g10m.npz --gpu-only --no-sq8 --out gpurun_out/r4j_hnsw_10m_hash$L.json > /tmp/h2.log 2>&1; echo "rc=$?"
done
RXGPU_HNSW_VISITED=bitset timeout 600 $B --graph /tmp/g10m.npz --gpu-only --no-sq8 --out gpurun_out/r4j_hnsw_10m_bitset.json > /tmp/h3.log 2>&1; echo "rc=$?"
python - <<'PY'
import json
for tag in ('', '_hash14', '_hash15', '_hash16', '_bitset'):
    try:
        d = json.load(open(f'gpurun_out/r4j_hnsw_10m{tag}.json')); g = d['gpu']
        print(tag or 'default', 'qps', round(g['queries_per_sec']), 'kernel-only', round(g['queries_per_sec_kernel_only']), 'frac', round(g['roofline']['frac'], 3), 'evals', round(g['distance_evals_per_query'], 1),
              'redo', g.get('redo_launches'), g.get('redo_ms'), 'ties', g.get('tie_reruns'), 'equal', d.get('equal_to_reference_frac'), 'sq8', (d.get('sq8') or {}).get('gpu', {}).get('queries_per_sec'))
    except Exception as e:
        print(tag, 'failed', repr(e))
PY
cd /tmp
P="python $R/tools/bench_hnsw.py --rows 10000000 --queries 16384 --no-map-legs --graph /tmp/g10m.npz --gpu-only --no-sq8"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_j/trace -o j -- $P --out /tmp/x1.json > /tmp/p1.log 2>&1; echo "trace rc=$?"
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_j/fetch -o j -- $P --out /tmp/x2.json > /tmp/p2.log 2>&1; echo "fetch rc=$?"
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_j/write -o j -- $P --out /tmp/x3.json > /tmp/p3.log 2>&1; echo "write rc=$?"
cd $R
for f in $(find gpurun_out/prof_j -name "*counter_collection.csv"); do (head -1 "$f"; grep hnsw "$f") > "$f.rx" && mv "$f.rx" "$f"; done
find gpurun_out/prof_j -name "*.csv" -size +4M -delete
find gpurun_out/prof_j -name "*stats.csv" | head; for f in $(find gpurun_out/prof_j -name "*kernel_stats.csv"); do head -6 $f; done
