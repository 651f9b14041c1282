#!/bin/bash
# Round-3 side measurements in one gpurun call: the batched (MFMA) leg for L2 and cosine, the wide-k merge probe, IVF (device lists, wide probes,
# batch), BM25 with concurrent callers, then the HNSW session (tests, 1M x 768 leg, rocprofv3 trace + FETCH_SIZE / WRITE_SIZE of the search kernel).
set -u
TAG=${1:-rd3g}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
# the headline scan's PMC passes again (the first session's counter files did not survive its clean-up): FETCH_SIZE / WRITE_SIZE in their own runs
mkdir -p gpurun_out/prof
PROF="python $R/bench.py --steps 30 --warmup 5 --no-cpu --batch 0 --hnsw-rows 0 --hybrid-docs 0 --ft-packed-words 0"
(cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/trace -o rd3f -- $PROF > /tmp/p1.log 2>&1
 timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_fetch -o rd3f -- $PROF > /tmp/p2.log 2>&1
 timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_write -o rd3f -- $PROF > /tmp/p3.log 2>&1)
for d in trace pmc_fetch pmc_write; do for f in $(find gpurun_out/prof/$d -name "rd3f_*.csv"); do cp "$f" gpurun_out/prof/$d/ 2>/dev/null; done; done
python tools/summarize_prof.py gpurun_out/prof rd3f > gpurun_out/rd3f_summarize.log 2>&1; tail -3 gpurun_out/rd3f_summarize.log
cp profiles/rd3f_kernel_stats.csv profiles/rd3f_rocprof_summary.json gpurun_out/ 2>/dev/null
for f in $(find gpurun_out/prof -name "*counter_collection.csv"); do (head -1 "$f"; grep rxgpu "$f") > "$f.rx" && mv "$f.rx" "$f"; done
find gpurun_out/prof -name "*.csv" -size +4M -delete
for m in l2 cosine; do
  timeout 300 python bench.py --metric $m --no-cpu --hnsw-rows 0 --hybrid-docs 0 --ft-packed-words 0 --steps 5 --warmup 2 --batch-iters 5 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_$m.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench_$m.json').read())
b=d.get('batched',{})
print('BATCHED $m', d.get('value'), d['roofline']['frac'], json.dumps({k:b.get(k) for k in ('ms_per_batch','queries_per_sec','rescore_ms','equals_batch1_rows','equals_batch1_dist_bits')}), json.dumps(b.get('roofline'))[:400])
PY
done
timeout 300 python tools/probe_wide_k.py 2>/dev/null | grep PROBE > gpurun_out/${TAG}_wide_k.jsonl; cut -c1-200 gpurun_out/${TAG}_wide_k.jsonl
timeout 600 python tools/bench_ivf.py --out gpurun_out/${TAG}_ivf.json 2>&1 | tail -c 1500
RXGPU_FT_LANES=8 timeout 400 python tools/bench_bm25.py --ops 1,1,1 --docs 5000000 --queries 40 --threads 1,2,4,8,16 --out gpurun_out/${TAG}_bm25_concurrent.json 2>/dev/null | tail -c 600
bash tools/gpu_session_hnsw.sh ${TAG} 1000000 2>&1 | tail -40
cp /tmp/prof/*/${TAG}*stats*.csv gpurun_out/ 2>/dev/null
ls profiles | grep ${TAG} ; cp profiles/${TAG}* gpurun_out/ 2>/dev/null
