#!/bin/bash
# round 4, session o: the resident hybrid form for queries with multi-word synonyms (tests), the FT / hybrid suites around it
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ft_synonyms.py tests/test_gpu_hybrid.py tests/test_gpu_hybrid_fuse.py tests/test_gpu_concurrency.py tests/test_gpu_ft_seam.py -q -m gpu 2>&1 | tail -15
