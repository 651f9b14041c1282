#!/bin/bash
# One gpurun call for the ft_fast merge with COUNTERS: the 3 x 3 OR merge at 5M vdocs under rocprofv3 three times — kernel trace + stats,
# --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, counters never combined with trace domains) — condensed by tools/summarize_prof2.py
# into gpurun_out/<tag>_bm25_rocprof.json.  Usage: gpurun -- bash tools/gpu_session_ft_pmc.sh <tag>
set -u
TAG=${1:-rd3b}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
CMD="python $R/tools/bench_bm25.py --ops 1,1,1 --docs 5000000 --queries 30"
timeout 600 $CMD --out gpurun_out/${TAG}_bm25_terms_1_1_1.json > gpurun_out/${TAG}_bm25_terms.log 2>&1
tail -c 1500 gpurun_out/${TAG}_bm25_terms.log
cd /tmp && rm -rf /tmp/prof_ft && mkdir -p /tmp/prof_ft
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ft/trace -o t -- $CMD > /tmp/prof_ft/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_ft/pmc_fetch -o f -- $CMD > /tmp/prof_ft/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_ft/pmc_write -o w -- $CMD > /tmp/prof_ft/write.log 2>&1
cd "$R"
python tools/summarize_prof2.py /tmp/prof_ft ${TAG}_bm25 ft_ "rocprofv3 (--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE) -- $CMD" > gpurun_out/${TAG}_bm25_rocprof.log 2>&1
tail -c 600 gpurun_out/${TAG}_bm25_rocprof.log
