#!/bin/bash
# round 4, closing session at the last commit: smoke, the whole -m gpu suite, the driver-shaped bench line + rocprofv3 evidence of the same command,
# then the BM25 train at several batch sizes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/gpu_session_final.sh rd4y
timeout 300 python tools/bench_bm25.py --ops 1,1,1 --docs 5000000 --queries 32 --batch 4,8,16,32,64 --out gpurun_out/rd4y_bm25_terms_1_1_1_batched.json > /tmp/bm.log 2>&1; echo "bm25 rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/rd4y_bm25_terms_1_1_1_batched.json'))
print('single', d['gpu']['term_pass_ms_per_merge'])
for b in d.get('batched_trains', []):
    print(b['queries_per_train'], round(b['kernel_ms_per_merge'] * 1e3, 2), 'us', b['identical_to_single_merge'])
PY
