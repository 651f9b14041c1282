#!/usr/bin/env python3
"""A/B of the two bf16 nomination kernels (knn_batched_bf16.hip) inside ONE process, on one corpus and one box: the single-ring kernel
(RXGPU_GEMM_SPLIT=0) against the split-ring one (default), alternating, for every metric.  Kernel time = the library's own HIP events around
the nomination GEMM's filter pass over the whole corpus (the "gemm" profile scope); results are checked against each other.
    python tools/bench_gemm_ab.py [--rows 10000000] [--dim 768] [--batch 256] [--metrics ip,l2,cosine] [--rounds 3] [--iters 4] [--out f.json]"""
import argparse
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import bench  # noqa: E402
from reindexer_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metrics", default="ip,l2,cosine")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    corpus = bench.make_corpus(a.rows, a.dim, 20260924, dev)
    queries = bench.make_corpus(a.batch, a.dim, 777, dev)
    kk = a.k + 1
    stream = torch.cuda.current_stream(dev).cuda_stream
    result = {"rows": a.rows, "dim": a.dim, "batch": a.batch, "kk": kk, "metrics": {}}
    for metric in a.metrics.split(","):
        mid = capi.METRICS[metric]
        d_inv = (1.0 / torch.linalg.vector_norm(corpus, dim=1)) if mid == 2 else None
        q = queries if mid != 2 else queries / torch.linalg.vector_norm(queries, dim=1, keepdim=True)
        q = q.contiguous()
        ix = capi.VectorIndex(mid, a.dim, device=0)
        ix.adopt_device_rows(corpus.data_ptr(), a.rows, a.dim, d_inv.data_ptr() if d_inv is not None else None, keepalive=(corpus, d_inv))
        modes = [("split_ring", {"RXGPU_GEMM_SPLIT": "1"}), ("single_ring", {"RXGPU_GEMM_SPLIT": "0"})]
        if a.batch > 128:
            modes.insert(1, ("split_ring_7_2", {"RXGPU_GEMM_SPLIT": "1", "RXGPU_GEMM_RINGS": "72"}))

        def set_mode(env):
            os.environ.pop("RXGPU_GEMM_RINGS", None)
            os.environ.update(env)

        od = {m: torch.empty((a.batch, kk), dtype=torch.float32, device=dev) for m, _ in modes}
        orow = {m: torch.empty((a.batch, kk), dtype=torch.int32, device=dev) for m, _ in modes}
        times = {m: [] for m, _ in modes}
        for mode, env in modes:   # warm-up of all (shadow, row statistics, LDS attribute)
            set_mode(env)
            ix.search_knn_device(q.data_ptr(), a.batch, kk, od[mode].data_ptr(), orow[mode].data_ptr(), None, stream)
        torch.cuda.synchronize(dev)
        ix.profile_enable(True)
        for _ in range(a.rounds):
            for mode, env in modes:
                set_mode(env)
                ix.profile_read("gemm")
                for _ in range(a.iters):
                    ix.search_knn_device(q.data_ptr(), a.batch, kk, od[mode].data_ptr(), orow[mode].data_ptr(), None, stream)
                torch.cuda.synchronize(dev)
                n, ms = ix.profile_read("gemm")
                times[mode].append(ms / max(n, 1))
        ix.profile_enable(False)
        first = modes[0][0]
        same = all(bool(torch.equal(orow[m], orow[first]) and torch.equal(od[m].view(torch.int32), od[first].view(torch.int32))) for m, _ in modes)
        flops = 2.0 * (128 if a.batch <= 128 else 256) * a.rows * ((a.dim + 63) // 64 * 64)
        shadow = float(a.rows) * ((a.dim + 63) // 64 * 64) * 2
        entry = {"identical_results": same}
        for name, _ in modes:
            best = min(times[name])
            entry[name] = {"gemm_ms_per_launch": times[name], "best_ms": best, "mfma_frac_of_2500TF": flops / (best / 1e3) / 1e12 / 2500.0,
                           "hbm_frac_on_shadow": shadow / (best / 1e3) / 1e9 / 8000.0}
        result["metrics"][metric] = entry
        print(metric, json.dumps(entry), flush=True)
        del ix
    os.environ.pop("RXGPU_GEMM_SPLIT", None)
    os.environ.pop("RXGPU_GEMM_RINGS", None)
    if a.out:
        Path(a.out).parent.mkdir(parents=True, exist_ok=True)
        Path(a.out).write_text(json.dumps(result) + "\n")


if __name__ == "__main__":
    main()
