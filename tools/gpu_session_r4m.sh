#!/bin/bash
# round 4, session m: the bf16 shadow in tile-blocked order — parity (both layouts, both nomination kernels, the pruned scan, mutations), then
# the four-way A/B on the headline corpus, then the counters of the kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_batched.py tests/test_gpu_pruned.py -q -m gpu -x 2>&1 | tail -4
timeout 900 python tools/bench_gemm_ab.py --pruned --out gpurun_out/r4m_gemm_ab.json 2>&1 | grep -v amdgpu.ids | tail -6
bash tools/gpu_session_r4_gemm_pmc.sh 2>&1 | tail -12
