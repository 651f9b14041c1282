#!/bin/bash
# round 5: counters of the default nomination kernel (knn_gemm_bf16_qreg: rows by LDS-DMA, queries through registers) on the headline
# corpus, one counter set per pass with the kernel trace beside it — the same sets as profiles/rd4_gemm_pmc.json holds for the DMA-only kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; rm -rf gpurun_out/gemm_pmc; mkdir -p gpurun_out/gemm_pmc; export TMPDIR=/tmp
CMD="python $R/tools/bench_gemm_ab.py --metrics ip --rounds 1 --iters 2 --modes split_ring_blocked_shadow"
cd /tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/gemm_pmc/pass$i -o g -- $CMD > /tmp/gp$i.log 2>&1; echo "pass $i rc=$?"; tail -1 /tmp/gp$i.log | cut -c1-200
done
cd $R
for f in $(find gpurun_out/gemm_pmc -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do (head -1 "$f"; grep knn_gemm_bf16 "$f") > "$f.rx" && mv "$f.rx" "$f"; done
find gpurun_out/gemm_pmc -name "*.csv" -size +2M -delete
python tools/summarize_gemm_pmc.py gpurun_out/gemm_pmc gpurun_out/rd5_gemm_pmc.json "rocprofv3 --kernel-trace --pmc <one set per pass>, tools/bench_gemm_ab.py --metrics ip --rounds 1 --iters 2 --modes split_ring_blocked_shadow with the round-5 default RXGPU_GEMM_QREG=1 (10M x 768, 256 queries; tile-blocked bf16 shadow)"
