#!/bin/bash
# round 4, session b: the batched BM25 train (grid.y = query) — every FT test, then the 3 x 3 merge and the hybrid leg
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ft_batch.py tests/test_gpu_bm25.py tests/test_gpu_ft_terms.py tests/test_gpu_ft_phrases.py tests/test_gpu_ft_synonyms.py tests/test_gpu_ft_seam.py tests/test_gpu_fuzz.py tests/test_gpu_hybrid.py tests/test_gpu_hybrid_fuse.py tests/test_gpu_concurrency.py -x -q 2>&1 | tail -15 > gpurun_out/r4b_ft_tests.txt
cat gpurun_out/r4b_ft_tests.txt
timeout 300 python tools/bench_bm25.py --ops 1,1,1 --queries 32 --batch 1,2,4,8,16,32 --out gpurun_out/r4b_bm25_terms_1_1_1.json > /dev/null 2> gpurun_out/r4b_bm25.err; tail -3 gpurun_out/r4b_bm25.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r4b_bm25_terms_1_1_1.json'))
print('single', d['gpu']['term_pass_ms_per_merge'], d['gpu']['roofline']['frac'], d.get('parity'))
for t in d.get('batched_trains',[]): print(t['queries_per_train'], round(t['kernel_ms_per_merge'],4), round(t['roofline']['frac'],3), round(t['roofline']['frac_20B_per_posting'],3), t['identical_to_single_merge'], round(t['merges_per_sec_wall']))
P
timeout 600 python tools/bench_hybrid.py --out gpurun_out/r4b_hybrid.json > /dev/null 2> gpurun_out/r4b_hybrid.err; tail -3 gpurun_out/r4b_hybrid.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r4b_hybrid.json'))
print(json.dumps(d['ft_half'])[:1500]); print(json.dumps(d['gpu'])[:600]); print(d.get('parity'))
P
