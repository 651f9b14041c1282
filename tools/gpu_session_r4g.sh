#!/bin/bash
# round 4, session g: SQ8 end to end (sampler, switch, code patches, quantised cache), packed decode host side, cache loader checks
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sq8.py tests/test_ann_cache.py tests/test_gpu_ft_packed.py tests/test_gpu_ft_seam.py tests/test_gpu_hnsw.py tests/test_gpu_cpp_engine.py -q 2>&1 | tail -12 > gpurun_out/r4g_tests.txt
cat gpurun_out/r4g_tests.txt
timeout 300 python tools/bench_ft_packed.py --out gpurun_out/r4g_ft_packed.json > /tmp/p.log 2>&1; tail -2 /tmp/p.log | cut -c1-1200
