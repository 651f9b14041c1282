#!/bin/bash
# round 5, closing measurements at the last commit (one gpurun call, ~7 min): the driver-shaped bench command with the HNSW leg at 1M rows
# (the 10M-row leg of the same command is profiles/rd5a_*, 690 s of host build) and without the hybrid / ft_packed legs (unchanged since
# rd5a), then the rocprofv3 kernel trace of the headline loop.  Usage: gpurun -- bash tools/gpu_session_r5z.sh
set -u
TAG=rd5z
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out/prof && export TMPDIR=/tmp
timeout 420 python bench.py --gpus 1 --steps 20 --warmup 5 --hnsw-rows 1000000 --hybrid-docs 0 --ft-packed-words 0 --full-json gpurun_out/${TAG}_bench_full.json > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_full.err
tail -c 400 gpurun_out/${TAG}_bench_full.err
wc -c gpurun_out/${TAG}_bench_line.json; head -c 1200 gpurun_out/${TAG}_bench_line.json; echo
PROF="python $R/bench.py --steps 30 --warmup 5 --no-cpu --hnsw-rows 0 --hybrid-docs 0 --ft-packed-words 0 --full-json /tmp/prof_bench_full.json"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/trace -o $TAG -- $PROF > /tmp/p1.log 2>&1
cd "$R"
tail -2 /tmp/p1.log
for f in $(find gpurun_out/prof/trace -name "${TAG}_*.csv"); do cp "$f" gpurun_out/prof/trace/ 2>/dev/null; done
ls gpurun_out/prof/trace | head
find gpurun_out/prof -name "*.csv" -size +4M -delete
