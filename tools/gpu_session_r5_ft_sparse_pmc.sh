#!/bin/bash
# round 5: the SPARSE ft_fast train (64 two-term merges of ~500 k postings, 5M documents — the shape of the hybrid query's FT half) with counters:
# kernel trace + stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE in separate passes -> gpurun_out/rd5_bm25_train64_sparse_rocprof.json
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
CMD="python $R/tools/bench_bm25.py --ops 1,1 --fracs 0.04,0.01 --queries 256 --batch 64 --batch-only"
cd /tmp && rm -rf /tmp/prof_ft && mkdir -p /tmp/prof_ft
timeout 75 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ft/trace -o t -- $CMD > /tmp/prof_ft/trace.log 2>&1; grep batch_only /tmp/prof_ft/trace.log | cut -c1-400
timeout 75 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_ft/pmc_fetch -o f -- $CMD > /tmp/prof_ft/fetch.log 2>&1
timeout 75 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_ft/pmc_write -o w -- $CMD > /tmp/prof_ft/write.log 2>&1
cd "$R"
python tools/summarize_prof2.py /tmp/prof_ft rd5_bm25_train64_sparse ft_ "rocprofv3 (--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE) -- $CMD" > gpurun_out/rd5_bm25_train64_sparse_rocprof.log 2>&1
tail -3 gpurun_out/rd5_bm25_train64_sparse_rocprof.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/rd5_bm25_train64_sparse_rocprof.json"))
rd = wr = ms = 0.0
calls = 0
for k, e in d["kernels"].items():
    rd += e.get("hbm_read_bytes_total_corrected_x2", 0); wr += e.get("hbm_write_bytes_total", 0); ms += e.get("total_ms", 0)
    if "ft_ranges" in k: calls = e.get("calls", 0)
print("trains", calls, "per merge: read MB", rd / max(calls, 1) / 64 / 1e6, "written MB", wr / max(calls, 1) / 64 / 1e6, "kernel us", ms / max(calls, 1) / 64 * 1e3)
PY
