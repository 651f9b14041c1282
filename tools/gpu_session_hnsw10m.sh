#!/bin/bash
# BASELINE configs[2] at its true size: 10M x 768 cosine HNSW, built on the box by the product's host builder, searched on the MI355X
# (float + SQ8) with the reference's engines on the same graph beside it.  Usage: gpurun --timeout 2700 -- bash tools/gpu_session_hnsw10m.sh <tag>
set -u
TAG=${1:-r2m}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
echo "memory.max=$(cat /sys/fs/cgroup/memory.max 2>/dev/null) cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" | tee gpurun_out/${TAG}_host.txt
free -g | tee -a gpurun_out/${TAG}_host.txt
timeout 2500 python tools/bench_hnsw.py --rows 10000000 --queries 16384 --build-threads 16 --no-map-legs --out gpurun_out/${TAG}_hnsw_10m.json \
	> gpurun_out/${TAG}_hnsw_10m.log 2>&1
echo "rc=$?"
tail -c 3000 gpurun_out/${TAG}_hnsw_10m.log
