#!/bin/bash
# round 4, session c: every FT test on the batched kernels; per-kernel times of a 64-query train (sparse: ~500 k postings per query; dense: 3.9 M)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ft_batch.py tests/test_gpu_bm25.py tests/test_gpu_ft_terms.py tests/test_gpu_ft_phrases.py tests/test_gpu_ft_synonyms.py tests/test_gpu_ft_seam.py tests/test_gpu_fuzz.py tests/test_gpu_hybrid.py tests/test_gpu_hybrid_fuse.py tests/test_gpu_concurrency.py -q 2>&1 | tail -15 > gpurun_out/r4c_ft_tests.txt
cat gpurun_out/r4c_ft_tests.txt
cd /tmp && export TMPDIR=/tmp
for shape in "sparse 1,1 0.04,0.01 64" "dense 1,1,1 0.2,0.05,0.01 16"; do
  set -- $shape
  rm -rf /tmp/prof_$1
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o p -- python $GRAFT_REPO_ROOT/tools/bench_bm25.py --ops $2 --fracs $3 --queries 256 --batch $4 --batch-only > $GRAFT_REPO_ROOT/gpurun_out/r4c_prof_$1.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/r4c_prof_$1.log
  f=$(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1)
  cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r4c_kernel_stats_$1.csv
  head -12 "$f" | cut -c1-150
done
