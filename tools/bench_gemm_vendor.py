#!/usr/bin/env python3
"""Calibration of the bf16 nomination GEMM (knn_batched_bf16.hip) against the vendor library on the same box, in one process: the same
shape — 256 queries x 768 against 10M rows x 768, bf16 inputs, f32 accumulation — through torch.matmul (hipBLASLt / rocBLAS under
PyTorch-ROCm), in the two operand orders a library may prefer, against this library's kernel timed by its own HIP events.

The vendor GEMM must WRITE its product (256 x 10M bf16 = 5.1 GB; the nomination kernel writes candidates only), so it runs over row chunks
(1M rows: a 512 MB product) and its time is given both as measured and minus the HBM time of that write at 8 TB/s — the second is the figure
to hold the nomination kernel against.  kernel names + durations of the same run: rocprofv3 --kernel-trace (tools/gpu_session_r6_gemm_vendor.sh).
    python tools/bench_gemm_vendor.py [--rows 10000000] [--batch 256] [--chunk 1000000] [--out f.json]"""
import argparse
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import bench  # noqa: E402
from reindexer_amd import capi  # noqa: E402


def timed(fn, iters, dev):
    fn()
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(iters):
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize(dev)
        best = min(best, a.elapsed_time(b))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--chunk", type=int, default=1_000_000)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    corpus = bench.make_corpus(a.rows, a.dim, 20260924, dev)
    queries = bench.make_corpus(a.batch, a.dim, 777, dev).contiguous()
    out = {"rows": a.rows, "dim": a.dim, "batch": a.batch, "chunk": a.chunk, "torch": torch.__version__,
           "flops": 2.0 * a.batch * a.rows * a.dim, "bf16_rows_bytes": 2.0 * a.rows * a.dim, "product_bytes_bf16": 2.0 * a.batch * a.rows}

    # ---- this library: the nomination GEMM alone (the "gemm" profile scope = HIP events around the filter pass over the whole corpus)
    kk = 11
    ix = capi.VectorIndex(1, a.dim, device=0)
    ix.adopt_device_rows(corpus.data_ptr(), a.rows, a.dim, None, keepalive=(corpus,))
    od = torch.empty((a.batch, kk), dtype=torch.float32, device=dev)
    orow = torch.empty((a.batch, kk), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ix.search_knn_device(queries.data_ptr(), a.batch, kk, od.data_ptr(), orow.data_ptr(), None, stream)
    torch.cuda.synchronize(dev)
    ix.profile_enable(True)
    ix.profile_read("gemm")
    ours = []
    for _ in range(a.iters):
        ix.search_knn_device(queries.data_ptr(), a.batch, kk, od.data_ptr(), orow.data_ptr(), None, stream)
        torch.cuda.synchronize(dev)
        n, ms = ix.profile_read("gemm")
        ours.append(ms / max(n, 1))
    ix.profile_enable(False)
    out["nomination_gemm_ms"] = {"per_launch": ours, "best": min(ours)}
    ix.close()

    # ---- the vendor GEMM on a bf16 copy of the rows (row-major [N][D], what the library's shadow holds before blocking)
    rows16 = corpus.to(torch.bfloat16)
    del corpus
    torch.cuda.empty_cache()
    q16 = queries.to(torch.bfloat16)
    qT = q16.t().contiguous()            # [D][B]
    chunks = [(s, min(a.rows, s + a.chunk)) for s in range(0, a.rows, a.chunk)]
    prod_a = torch.empty((a.chunk, a.batch), dtype=torch.bfloat16, device=dev)   # rows x queries
    prod_b = torch.empty((a.batch, a.chunk), dtype=torch.bfloat16, device=dev)   # queries x rows

    def rows_times_qT():   # [Nc x D] . [D x B]: the tall GEMM, A row-major
        for s, e in chunks:
            torch.matmul(rows16[s:e], qT, out=prod_a[:e - s])

    def q_times_rowsT():   # [B x D] . [D x Nc]: B given as the transpose of the row-major rows
        for s, e in chunks:
            torch.matmul(q16, rows16[s:e].t(), out=prod_b[:, :e - s])

    write_ms = out["product_bytes_bf16"] / 8e12 * 1e3
    for name, fn in (("rows_x_queriesT", rows_times_qT), ("queries_x_rowsT", q_times_rowsT)):
        ms = timed(fn, a.iters, dev)
        out["vendor_" + name] = {"ms": ms, "ms_minus_product_write_at_8TBs": ms - write_ms, "tflops": out["flops"] / ms / 1e9,
                                 "frac_of_2500TF": out["flops"] / ms / 1e9 / 2500.0}
    best_vendor = min(out["vendor_rows_x_queriesT"]["ms"], out["vendor_queries_x_rowsT"]["ms"])
    out["product_write_ms_at_8TBs"] = write_ms
    out["floors_ms"] = {"hbm_rows_bf16_at_8TBs": out["bf16_rows_bytes"] / 8e12 * 1e3, "mfma_bf16_dense_2500TF": out["flops"] / 2.5e15 * 1e3,
                        "mfma_fp32_157TF": out["flops"] / 157e12 * 1e3}
    out["nomination_vs_vendor"] = {"vendor_best_ms": best_vendor, "vendor_best_minus_write_ms": best_vendor - write_ms, "ours_ms": min(ours),
                                   "ours_over_vendor": min(ours) / best_vendor, "ours_over_vendor_minus_write": min(ours) / (best_vendor - write_ms)}
    # north_star: "MFMA only for the batched-query x corpus GEMM case, evidenced ... vs the fp32 roofline"
    out["vs_fp32_roofline"] = {"nomination_tflops": out["flops"] / min(ours) / 1e9, "fp32_mfma_peak_tflops": 157.0,
                               "times_the_fp32_peak": out["flops"] / min(ours) / 1e9 / 157.0}
    print(json.dumps(out, indent=1))
    if a.out:
        Path(a.out).parent.mkdir(parents=True, exist_ok=True)
        Path(a.out).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
