#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_gpu_sq8.py tests/test_gpu_ft_packed.py tests/test_gpu_ft_seam.py -q > gpurun_out/r4h_tests.txt 2>&1; tail -5 gpurun_out/r4h_tests.txt
timeout 300 python tools/bench_ft_packed.py --out gpurun_out/r4h_ft_packed.json > /tmp/p.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4h_ft_packed.json'))['device']
print({k:d[k] for k in ('seconds','seconds_through_python_harness','words_per_sec')}, d['kernels']['count_ms'], d['kernels']['write_ms'], d['parity'])
PY
