for pipe in 0; do
for ex in 2 5 6 2 5 6; do
  RXGPU_GEMM_PIPE=$pipe RXGPU_GEMM_EXPERIMENT=$ex timeout 300 python bench.py --metric ip --no-cpu --hnsw-rows 0 --hybrid-docs 0 --ft-packed-words 0 --steps 3 --warmup 1 --batch-iters 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
b=d.get('batched',{})
r=b.get('roofline',{})
print('RES PIPE $pipe EXP $ex gemm_ms', r.get('avg_ms'), 'eq', b.get('equals_batch1_rows'))
"
done
done
