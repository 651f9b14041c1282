set -x
timeout 600 python -m pytest tests/test_gpu_ft_terms.py tests/test_gpu_ft_seam.py tests/test_gpu_bm25.py tests/test_gpu_hybrid_fuse.py -x -q 2>&1 | tail -8
mkdir -p gpurun_out
timeout 600 python tools/bench_bm25.py --ops 1,1,1 --docs 5000000 --queries 40 --threads 1,2,4,8 --out gpurun_out/bm25_conc.json 2>&1 | tail -3
RXGPU_FT_LANES=8 timeout 600 python tools/bench_bm25.py --ops 1,1,1 --docs 5000000 --queries 40 --threads 4,8,16 --out gpurun_out/bm25_conc8.json 2>&1 | tail -3
