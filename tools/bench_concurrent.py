#!/usr/bin/env python3
"""The reference's concurrency model — T query threads, each running its own SearchKnn on the shared index (cf. runMultithreadQueries,
gtests/tests/unit/float_vector_index.cc:258-294) — through GpuBruteforceMap with and without query coalescing.

    python tools/bench_concurrent.py --rows 2000000 --dim 768 --threads 64 --per-thread 40"""
import argparse
import json
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from reindexer_amd import hostapi  # noqa: E402


def run(m, queries, threads, per_thread, k):
    def work(t):
        for j in range(per_thread):
            m.search_knn(queries[(t * per_thread + j) % queries.shape[0]], k)
    ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    [t.start() for t in ths]
    [t.join() for t in ths]
    return threads * per_thread / (time.perf_counter() - t0)


def main_hnsw(args):
    rng = np.random.default_rng(4)
    centres = rng.normal(0, 0.25, (500, args.dim)).astype(np.float32)
    rows = (centres[rng.integers(0, 500, args.rows)] + rng.normal(0, 0.08, (args.rows, args.dim))).astype(np.float32)
    queries = (centres[rng.integers(0, 500, 1024)] + rng.normal(0, 0.08, (1024, args.dim))).astype(np.float32)
    queries = np.stack([hostapi.normalize_copy(q)[0] for q in queries])
    m = hostapi.GpuHnswMap(2, args.dim, args.rows, M=16, ef_construction=200)
    m.add(rows, np.arange(args.rows, dtype=np.uint64) << np.uint64(32))

    class Wrap:   # run() calls search_knn(q, k)
        def search_knn(self, q, k):
            return m.search_knn(q, k, 128)
    w = Wrap()
    w.search_knn(queries[0], args.k)
    out = {"workload": f"{args.threads} threads x {args.per_thread} single-query SearchKnn(k={args.k}, ef=128) calls on one GpuHnswMap, {args.rows} x {args.dim} cosine"}
    t0 = time.perf_counter()
    for i in range(50):
        w.search_knn(queries[i], args.k)
    out["single_thread_qps"] = 50 / (time.perf_counter() - t0)
    L = hostapi.lib()
    import ctypes as C
    L.rxhost_hnsw_enable_coalescing.argtypes = [C.c_void_p, C.c_int]
    L.rxhost_hnsw_enable_coalescing(m.h, 0)
    out["threads_no_coalescing_qps"] = run(w, queries, args.threads, args.per_thread, args.k)
    L.rxhost_hnsw_enable_coalescing(m.h, 1)
    out["threads_coalescing_qps"] = run(w, queries, args.threads, args.per_thread, args.k)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--threads", type=int, default=64)
    ap.add_argument("--per-thread", type=int, default=40)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--hnsw", action="store_true", help="the same experiment on GpuHnswMap (cosine, M=16, efC=200, ef=128); use --rows 50000")
    args = ap.parse_args()
    if args.hnsw:
        return main_hnsw(args)
    rng = np.random.default_rng(3)
    rows = rng.normal(0, 0.25, (args.rows, args.dim)).astype(np.float32)
    queries = rng.normal(0, 0.25, (1024, args.dim)).astype(np.float32)
    m = hostapi.GpuBruteforceMap(1, args.dim, args.rows)
    step = 200_000
    for a in range(0, args.rows, step):
        m.add(rows[a:a + step], np.arange(a, min(a + step, args.rows), dtype=np.uint64) << np.uint64(32))
    m.search_knn(queries[0], args.k)
    out = {"workload": f"{args.threads} threads x {args.per_thread} single-query SearchKnn calls on one GpuBruteforceMap, {args.rows} x {args.dim} ip k={args.k}"}
    t0 = time.perf_counter()
    for i in range(20):
        m.search_knn(queries[i], args.k)
    out["single_thread_qps"] = 20 / (time.perf_counter() - t0)
    m.enable_coalescing(False)
    out["threads_no_coalescing_qps"] = run(m, queries, args.threads, max(2, args.per_thread // 8), args.k)
    m.enable_coalescing(True)
    run(m, queries, args.threads, 2, args.k)   # warm-up: builds the bf16 shadow on the first batch
    b0, q0 = m.coalescing_stats()
    out["threads_coalescing_qps"] = run(m, queries, args.threads, args.per_thread, args.k)
    b1, q1 = m.coalescing_stats()
    out["avg_batch"] = (q1 - q0) / max(1, b1 - b0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
