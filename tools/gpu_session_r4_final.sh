#!/bin/bash
# round 4, closing session: smoke, the whole -m gpu suite, the driver-shaped bench line + rocprofv3 evidence of the same command (tools/gpu_session_final.sh),
# then the counters of the BM25 launch train of 16 merges
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/gpu_session_final.sh rd4z
bash tools/gpu_session_r4_ft_pmc.sh rd4z
