#!/bin/bash
# the ft_fast merge TRAIN of 16 queries (3 x 3 OR merge, 5M vdocs) with counters: kernel trace + stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE in
# separate passes, condensed into gpurun_out/<tag>_bm25_train16_rocprof.json.  Usage: gpurun -- bash tools/gpu_session_r4_ft_pmc.sh <tag>
set -u
TAG=${1:-rd4l}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
CMD="python $R/tools/bench_bm25.py --ops 1,1,1 --docs 5000000 --queries 128 --batch 16 --batch-only"
cd /tmp && rm -rf /tmp/prof_ft && mkdir -p /tmp/prof_ft
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ft/trace -o t -- $CMD > /tmp/prof_ft/trace.log 2>&1; tail -1 /tmp/prof_ft/trace.log
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_ft/pmc_fetch -o f -- $CMD > /tmp/prof_ft/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_ft/pmc_write -o w -- $CMD > /tmp/prof_ft/write.log 2>&1
cd "$R"
python tools/summarize_prof2.py /tmp/prof_ft ${TAG}_bm25_train16 ft_ "rocprofv3 (--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE) -- $CMD" > gpurun_out/${TAG}_bm25_train16_rocprof.log 2>&1
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bm25_train16_rocprof.json"))
rd = wr = ms = 0.0
calls = 0
for k, e in d["kernels"].items():
    rd += e.get("hbm_read_bytes_total_corrected_x2", 0); wr += e.get("hbm_write_bytes_total", 0); ms += e.get("total_ms", 0)
    if "ft_ranges" in k: calls = e.get("calls", 0)
print("trains", calls, "per merge: read MB", rd / max(calls, 1) / 16 / 1e6, "written MB", wr / max(calls, 1) / 16 / 1e6, "kernel us", ms / max(calls, 1) / 16 * 1e3)
PY
