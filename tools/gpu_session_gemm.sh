#!/bin/bash
# One gpurun call for the batched nomination GEMM: parity tests of the batched path, then the driver-shaped bench without the CPU / HNSW /
# hybrid legs (headline + batched + pruned + prefilter legs).  Usage: gpurun -- bash tools/gpu_session_gemm.sh <tag>
set -u
TAG=${1:-r3q}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_batched.py -x -q > gpurun_out/${TAG}_tests.log 2>&1
tail -5 gpurun_out/${TAG}_tests.log
timeout 300 python bench.py --no-cpu --hnsw-rows 0 --hybrid-docs 0 --steps 5 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.err
python - <<PY
import json
try:
    r = json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1])
    print('headline', r['value'], r['roofline']['frac'])
    print('batched', json.dumps(r.get('batched'))[:1500])
except Exception as e:
    print('no bench line', e)
PY
