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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--threads", type=int, default=64)
    ap.add_argument("--per-thread", type=int, default=40)
    ap.add_argument("--k", type=int, default=10)
    args = ap.parse_args()
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
