#!/bin/bash
# Round 6, closing validation after the sharded-index work of the last session (ANN cache of the HNSW Map over a device list; phrases, synonyms,
# areas and batches over BM25 document-range shards): smoke(), the whole -m gpu suite, the headline loop.  The driver-shaped bench with every
# leg (13 minutes, 10 of them the host build of the 10M-row graph) ran at 54f2f35 (profiles/rd6final_*): no kernel it times changed since.
# Usage: gpurun -- bash tools/gpu_session_r6_l.sh
set -u
TAG=rd6l
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; tail -3 gpurun_out/${TAG}_smoke.log
timeout 900 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/${TAG}_pytest_gpu.log 2>&1
tail -14 gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --batch 0 --hnsw-rows 0 --hybrid-docs 0 --ft-packed-words 0 \
  --full-json gpurun_out/${TAG}_bench_full.json > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
head -c 900 gpurun_out/${TAG}_bench_line.json
