#!/bin/bash
# Round 6: where a single HNSW search spends its cycles.  Rebuilds hnsw_search.hip with -DRXGPU_HNSW_PHASES into a scratch copy of the
# library on the box (the shipped librxgpu.so is untouched) and runs tools/hnsw_phases.py.
#   gpurun -- bash tools/gpu_session_r6_phases.sh <tag> [rows]
set -u
TAG=${1:-rd6p}
ROWS=${2:-1000000}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
cp reindexer_amd/librxgpu.so /tmp/librxgpu_ship.so
touch reindexer_amd/csrc/hnsw_search.hip reindexer_amd/csrc/rxgpu_capi.hip
RXGPU_HIP_DEFINES=-DRXGPU_HNSW_PHASES python -m reindexer_amd.build > gpurun_out/${TAG}_build.log 2>&1
RXGPU_HNSW_PHASES=1 timeout 900 python tools/hnsw_phases.py --rows $ROWS --queries 64 > gpurun_out/${TAG}_phases.log 2>&1
tail -12 gpurun_out/${TAG}_phases.log
