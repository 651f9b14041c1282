#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for f in test_gpu_sq8 test_ann_cache test_gpu_ft_packed test_gpu_ft_seam test_gpu_hnsw test_gpu_cpp_engine; do
  timeout 900 python -X faulthandler -m pytest tests/$f.py -q -x 2>&1 > gpurun_out/r4g2_$f.txt
  echo "== $f: $(tail -1 gpurun_out/r4g2_$f.txt)"
  grep -n "Fatal Python\|Segmentation\|File \"/root/repo/tests\|in test_" gpurun_out/r4g2_$f.txt | head -8
done
