#!/bin/bash
# PMC view of the bf16 nomination GEMM in three modes (RXGPU_GEMM_EXPERIMENT: 0 real, 2 no DMA, 3 no MFMA): matrix-core busy cycles,
# wave cycles, waits, LDS activity, effective clock.  Counters only (no trace domains beside the kernel trace).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" && mkdir -p gpurun_out && export TMPDIR=/tmp
CMD="python bench.py --metric ip --no-cpu --hnsw-rows 0 --hybrid-docs 0 --ft-packed-words 0 --steps 2 --warmup 1 --batch-iters 2"
for ex in 2; do
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-40)
    rm -rf /tmp/pmc_$ex
    RXGPU_GEMM_PIPE=${PIPE:-0} RXGPU_GEMM_EXPERIMENT=$ex timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$ex -o out --output-format csv -- $CMD > /dev/null 2>&1
    f=$(find /tmp/pmc_$ex -name "*counter_collection.csv" | head -1)
    python - "$f" "$ex" <<'PY'
import csv, sys, collections
f, ex = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
try:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "knn_gemm_bf16_glds" not in k or "Li1ELi1E" not in k.replace(" ", ""):
            if "glds<1, 1" not in k:
                continue
        c = row["Counter_Name"]
        acc[c][0] += 1
        acc[c][1] += float(row["Counter_Value"])
    for c, (n, v) in acc.items():
        print("PMC EXP", ex, c, "launches", n, "avg", v / max(n, 1))
except Exception as e:
    print("PMC EXP", ex, "error", repr(e), f)
PY
  done
done
