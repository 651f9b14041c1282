# A/B of the fragment-pipelined bf16 nomination GEMM (RXGPU_GEMM_PIPE=0: the kernel without it), same box, interleaved
set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_batched.py -x -q 2>&1 | tail -4
for m in ip l2; do
for pipe in 0 1 0 1; do
  RXGPU_GEMM_PIPE=$pipe timeout 300 python bench.py --metric $m --no-cpu --hnsw-rows 0 --hybrid-docs 0 --ft-packed-words 0 --steps 3 --warmup 1 --batch-iters 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
b=d.get('batched',{})
r=b.get('roofline',{})
print('METRIC $m PIPE $pipe gemm_ms', r.get('avg_ms'), 'frac', r.get('frac'), 'hbm', (r.get('hbm') or {}).get('frac'), 'ms_per_batch', b.get('ms_per_batch'), 'eq', b.get('equals_batch1_rows'), b.get('equals_batch1_dist_bits'))
"
done
done
