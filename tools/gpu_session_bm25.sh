#!/bin/bash
# One gpurun call for the ft_fast merge: parity tests, the 3 x 3 OR merge at 5M vdocs (and the single-term merge), and a rocprofv3 kernel
# trace of the same merge -> per-kernel average durations.  Usage: gpurun -- bash tools/gpu_session_bm25.sh <tag> [extra pytest args]
set -u
TAG=${1:-r2n}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bm25.py tests/test_gpu_ft_terms.py tests/test_gpu_hybrid.py -x -q > gpurun_out/${TAG}_tests.log 2>&1
tail -8 gpurun_out/${TAG}_tests.log
RXGPU_FT_TRACE=1 RXGPU_FT_STAMPS=${STAMP_BLOCK:-300} timeout 600 python tools/bench_bm25.py --ops 1,1,1 --docs 5000000 --queries 30 --out gpurun_out/${TAG}_bm25_terms_1_1_1.json > gpurun_out/${TAG}_bm25_terms.log 2>&1
tail -c 1800 gpurun_out/${TAG}_bm25_terms.log
timeout 600 python tools/bench_bm25.py --docs 5000000 --queries 20 --out gpurun_out/${TAG}_bm25_single.json > gpurun_out/${TAG}_bm25_single.log 2>&1
tail -c 600 gpurun_out/${TAG}_bm25_single.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bm25 -o $TAG -- python $R/tools/bench_bm25.py --ops 1,1,1 --docs 5000000 --queries 30 > /tmp/prof_bm25.log 2>&1
cd "$R"
python - <<PY
import csv, glob, json
f = glob.glob('/tmp/prof_bm25/**/*kernel_stats.csv', recursive=True)
out = {}
if f:
    for r in csv.DictReader(open(f[0])):
        if 'ft_' in r['Name']:
            out[r['Name'][:60]] = {'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3, 'min_us': float(r['MinNs']) / 1e3, 'max_us': float(r['MaxNs']) / 1e3}
json.dump({'command': 'rocprofv3 --kernel-trace --stats -- python tools/bench_bm25.py --ops 1,1,1 --docs 5000000 --queries 30', 'kernels': out,
           'train_avg_us': sum(v['avg_us'] for v in out.values())}, open('gpurun_out/${TAG}_bm25_kernels.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
print('train avg us', sum(v['avg_us'] for v in out.values()))
PY
