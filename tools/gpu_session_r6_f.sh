#!/bin/bash
# Round 6: the whole -m gpu suite (no -x: every failure listed), then the call-size sweep on a 10M x 768 graph (incl. one query repeated nq
# times: what concurrency costs on the device, apart from the mailbox and the host).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/rd6f_pytest_gpu.log 2>&1; tail -6 gpurun_out/rd6f_pytest_gpu.log | cut -c1-300
timeout 1800 python tools/bench_hnsw_nq_sweep.py --rows ${1:-10000000} --lanes 4 --threads 16,64 --out gpurun_out/rd6f_hnsw_nq_sweep_10m.json > gpurun_out/rd6f_hnsw_nq_sweep_10m.log 2>&1
grep -E "^nq|same query|lanes" gpurun_out/rd6f_hnsw_nq_sweep_10m.log | cut -c1-200
