#!/bin/bash
# Round-end validation in one gpurun call: the whole -m gpu suite, the driver-shaped bench line, and the rocprofv3 evidence for the scan
# kernel of the same command (kernel trace; FETCH_SIZE and WRITE_SIZE in their own passes).  Usage: gpurun -- bash tools/gpu_session_final.sh <tag>
set -u
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out/prof && export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; tail -4 gpurun_out/${TAG}_smoke.log
timeout 1500 python -m pytest tests -q -m gpu --durations=10 > gpurun_out/${TAG}_pytest_gpu.log 2>&1
tail -4 gpurun_out/${TAG}_pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-json gpurun_out/${TAG}_bench_full.json > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_full.err
tail -c 600 gpurun_out/${TAG}_bench_full.err
wc -c gpurun_out/${TAG}_bench_line.json; head -c 1500 gpurun_out/${TAG}_bench_line.json
PROF="python $R/bench.py --steps 30 --warmup 5 --no-cpu --batch 0 --hnsw-rows 0 --hybrid-docs 0 --ft-packed-words 0 --full-json /tmp/prof_bench_full.json"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/trace -o $TAG -- $PROF > /tmp/p1.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_fetch -o $TAG -- $PROF > /tmp/p2.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_write -o $TAG -- $PROF > /tmp/p3.log 2>&1
cd "$R"
find gpurun_out/prof -name "*.csv" | head; tail -2 /tmp/p1.log
# flatten rocprofv3's per-host subdirectories for tools/summarize_prof.py
for d in trace pmc_fetch pmc_write; do for f in $(find gpurun_out/prof/$d -name "${TAG}_*.csv"); do cp "$f" gpurun_out/prof/$d/ 2>/dev/null; done; done
python tools/summarize_prof.py gpurun_out/prof $TAG > gpurun_out/${TAG}_summarize.log 2>&1; tail -5 gpurun_out/${TAG}_summarize.log
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_rocprof_summary.json gpurun_out/ 2>/dev/null   # profiles/ does not travel back, gpurun_out/ does
# keep only what is small: of the counter files only this library's kernels
for f in $(find gpurun_out/prof -name "*counter_collection.csv"); do (head -1 "$f"; grep rxgpu "$f") > "$f.rx" && mv "$f.rx" "$f"; done
find gpurun_out/prof -name "*.csv" -size +4M -delete
