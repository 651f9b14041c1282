cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
timeout 600 python tools/bench_hnsw.py --rows 1000000 --queries 16384 --build-threads 16 --save-graph /tmp/g.npz --gpu-only --out /tmp/first.json > /tmp/first.log 2>&1
for cap in 0 384 600 780 1024; do
  RXGPU_HNSW_RESTART_CAND=$cap timeout 300 python tools/bench_hnsw.py --rows 1000000 --queries 16384 --graph /tmp/g.npz --gpu-only --out /tmp/cap_$cap.json > /tmp/cap_$cap.log 2>&1
  python -c "
import json
g = json.load(open('/tmp/cap_$cap.json'))['gpu']
print('CAP $cap', {k: g.get(k) for k in ('queries_per_sec', 'queries_per_sec_kernel_only', 'kernel_ms_total', 'tie_reruns', 'tie_rerun_ms', 'redo_ms', 'redo_launches')})" | tee -a gpurun_out/rd3p_restart_caps.txt
done
