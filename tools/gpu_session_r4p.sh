#!/bin/bash
# round 4, session p: s_setprio around the MFMA bursts of the nomination kernel (A/B in one process, ip / l2 / cosine)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/bench_gemm_ab.py --metrics ip,l2,cosine --rounds 3 --iters 4 --modes split_ring_blocked_shadow,split_ring_blocked_shadow_setprio --out gpurun_out/r4p_gemm_setprio.json 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for line in sys.stdin:
    try:
        m, rest = line.split(' ', 1); d = json.loads(rest)
        print(m, 'plain', [round(x, 3) for x in d['split_ring_blocked_shadow']['gemm_ms_per_launch']], 'setprio', [round(x, 3) for x in d['split_ring_blocked_shadow_setprio']['gemm_ms_per_launch']], d['identical_results'])
    except Exception:
        print(line[:200])
"
