set -u
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
timeout 600 python tools/bench_hnsw.py --rows 1000000 --queries 16384 --build-threads 16 --save-graph /tmp/g.npz --gpu-only --no-sq8 > /tmp/b0.json 2>/dev/null
python -c "
import json; d=json.load(open('/tmp/b0.json')); g=d['gpu']; print('default', g['queries_per_sec_kernel_only'], g['launches'], g['redo_launches'], g['redo_ms'])"
for cap in 1024 768 512 384 256; do
RXGPU_HNSW_LDS_CAND_CAP=$cap timeout 300 python tools/bench_hnsw.py --rows 1000000 --queries 16384 --graph /tmp/g.npz --gpu-only --no-sq8 > /tmp/b.json 2>/dev/null
python -c "
import json; d=json.load(open('/tmp/b.json')); g=d['gpu']; print('cap', $cap, g['queries_per_sec_kernel_only'], g['launches'], g['redo_launches'], g['redo_ms'], d['recall_at_k_vs_exact'])"
done
