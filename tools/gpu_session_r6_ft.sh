#!/bin/bash
# Round 6, sparse ft_fast train: one gpurun call = FT parity tests, phase stamps of a single sparse two-term merge, trains of 64 merges
# (bench_bm25 --batch 64, each checked against the single merge), the hybrid leg's FT half (bench_hybrid at a small vector dimension: the
# FT half does not depend on it) and a rocprofv3 kernel trace of nothing but trains.
#   gpurun -- bash tools/gpu_session_r6_ft.sh <tag> [tests]     tests = 0 skips pytest
set -u
TAG=${1:-rd6a}
TESTS=${2:-1}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
if [ "$TESTS" = "1" ]; then
  timeout 1500 python -m pytest tests/test_gpu_bm25.py tests/test_gpu_ft_terms.py tests/test_gpu_ft_batch.py tests/test_gpu_ft_phrases.py tests/test_gpu_ft_synonyms.py \
    tests/test_gpu_ft_areas.py tests/test_gpu_ft_sharded.py tests/test_gpu_ft_seam.py tests/test_gpu_hybrid.py tests/test_gpu_hybrid_fuse.py -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1
  tail -5 gpurun_out/${TAG}_tests.log
fi
SPARSE="--ops 1,1 --fracs 0.04,0.01 --docs 5000000"
RXGPU_FT_TRACE=1 RXGPU_FT_STAMPS=${STAMP_BLOCK:-300} timeout 600 python tools/bench_bm25.py $SPARSE --queries 30 > gpurun_out/${TAG}_sparse_stamps.log 2>&1
grep "rxgpu ft" gpurun_out/${TAG}_sparse_stamps.log | tail -12
timeout 600 python tools/bench_bm25.py $SPARSE --queries 256 --batch 1,8,64 --out gpurun_out/${TAG}_sparse_batch.json > gpurun_out/${TAG}_sparse_batch.log 2>&1
python - <<PY
import json
d = json.load(open('gpurun_out/${TAG}_sparse_batch.json'))
print('single', d['gpu']['term_pass_ms_per_merge'], 'parity', d.get('parity'), d.get('cpu_reference'))
for t in d.get('batched_trains', []):
    print(t['queries_per_train'], 'ms/train', round(t['kernel_ms_per_train'], 4), 'same', t['identical_to_single_merge'], 'frac20', round(t['roofline']['frac_20B_per_posting'], 4))
PY
# dense merges must not get slower: 3 terms x (20 %, 5 %, 1 %) of 5M documents, trains of 16
timeout 600 python tools/bench_bm25.py --ops 1,1,1 --docs 5000000 --queries 64 --batch 16 --out gpurun_out/${TAG}_dense_batch.json > gpurun_out/${TAG}_dense_batch.log 2>&1
python - <<PY
import json
d = json.load(open('gpurun_out/${TAG}_dense_batch.json'))
print('dense single', d['gpu']['term_pass_ms_per_merge'], 'parity', d.get('parity'))
for t in d.get('batched_trains', []):
    print('dense', t['queries_per_train'], 'ms/merge', round(t['kernel_ms_per_merge'], 4), 'same', t['identical_to_single_merge'])
PY
timeout 900 python tools/bench_hybrid.py --dim ${HYB_DIM:-64} --queries 64 --cpu-queries ${HYB_CPU:-16} --out gpurun_out/${TAG}_hybrid.json > gpurun_out/${TAG}_hybrid.log 2>&1
python - <<PY
import json
d = json.load(open('gpurun_out/${TAG}_hybrid.json'))
f = d['ft_half']
print('hybrid ft_half ms/train', f['ms_kernels_per_train'], 'postings/merge', f['postings_per_merge'], 'frac', f['roofline']['frac'], 'same', f['identical_to_single_merges_frac'])
print('hybrid parity', d.get('parity'), 'gpu q/s', d['gpu']['queries_per_sec'])
PY
cd /tmp && rm -rf /tmp/prof_ft && mkdir -p /tmp/prof_ft
CMD="python $R/tools/bench_bm25.py $SPARSE --queries 256 --batch 64 --batch-only"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ft/trace -o t -- $CMD > /tmp/prof_ft/trace.log 2>&1
if [ "${PMC:-0}" = "1" ]; then
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_ft/pmc_fetch -o f -- $CMD > /tmp/prof_ft/fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_ft/pmc_write -o w -- $CMD > /tmp/prof_ft/write.log 2>&1
fi
cd "$R"
python tools/summarize_prof2.py /tmp/prof_ft ${TAG}_bm25_train64_sparse ft_ "rocprofv3 (--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE) -- $CMD" > gpurun_out/${TAG}_bm25_rocprof.log 2>&1
tail -c 1500 gpurun_out/${TAG}_bm25_rocprof.log
