#!/bin/bash
# round 4, session d: seam re-commit fingerprint, resident sessions under concurrency, wide-k hybrid; per-kernel times of batched trains
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ft_seam.py tests/test_gpu_concurrency.py tests/test_gpu_hybrid.py tests/test_gpu_hybrid_fuse.py tests/test_gpu_ft_batch.py -q 2>&1 | tail -15 > gpurun_out/r4d_tests.txt
cat gpurun_out/r4d_tests.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for shape in "sparse 1,1 0.04,0.01 64" "dense 1,1,1 0.2,0.05,0.01 16"; do
  set -- $shape
  rm -rf /tmp/prof_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o p -- python $R/tools/bench_bm25.py --ops $2 --fracs $3 --queries 256 --batch $4 --batch-only > $R/gpurun_out/r4d_prof_$1.log 2>&1
  grep batch_only $R/gpurun_out/r4d_prof_$1.log
  f=$(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/r4d_kernel_stats_$1.csv
  head -12 "$f" | cut -c1-160
done
