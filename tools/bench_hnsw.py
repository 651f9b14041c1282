#!/usr/bin/env python3
"""HNSW leg (BASELINE configs[2], scaled): cosine, M=16, ef_construction=200, ef=128, k=10 on N x 768.

    python tools/bench_hnsw.py --rows 100000 --queries 4096 [--out profiles/r1_hnsw.json]

The full 10M x 768 graph takes hours to build on any CPU (the reference's build is CPU-only too), so this leg runs a scaled
corpus: graph built on the host by the product's builder, searched (a) on the MI355X through GpuHnswMap / rxgpu_hnsw_search_knn
and (b) by the reference engine itself (oracle/_ref, AVX-512) on the SAME graph where that library loads — reporting
queries/s, achieved HBM GB/s from the kernel's own counters (distance evaluations x D x 4 + hops x (1+2M) x 4), the fraction
of queries whose result equals the reference's, and recall@10 vs exact brute force (computed on the GPU by the exact scan).
"""
import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("RX_TARGET_INSTRUCTIONS", "avx512")

from reindexer_amd import capi, hostapi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=4096)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--efc", type=int, default=200)
    ap.add_argument("--metric", default="cosine")
    ap.add_argument("--cpu-queries", type=int, default=256)
    ap.add_argument("--clusters", type=int, default=2000,
                    help="synthetic corpus = cluster centre + noise (embedding-like: low intrinsic dimension); 0 = i.i.d. gaussian, "
                         "the reference tests' distribution, on which ANY graph index has poor recall at 768 dims")
    ap.add_argument("--graph", default=None, help="graph saved by tools/build_hnsw_graph.py (same corpus seed): skip the host build")
    ap.add_argument("--build-threads", type=int, default=0,
                    help="build the graph here with this many inserting threads (GpuHnswMap<OnInsertions>, the reference's multithreaded build); "
                         "the CPU baseline is then the restated engine on the same graph")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    metric = capi.METRICS[args.metric]
    saved = None
    if args.graph:
        saved = np.load(args.graph)
        meta = saved["meta"]
        args.rows, args.dim, args.M, args.clusters, args.efc = int(meta[1]), int(meta[2]), int(meta[3]), int(meta[8]), int(meta[9])
        assert int(meta[0]) == metric
    rng = np.random.default_rng(20260924)
    if args.clusters:
        centres = rng.normal(0, 0.25, (args.clusters, args.dim)).astype(np.float32)
        rows = (centres[rng.integers(0, args.clusters, args.rows)] + rng.normal(0, 0.08, (args.rows, args.dim))).astype(np.float32)
        queries = (centres[rng.integers(0, args.clusters, args.queries)] + rng.normal(0, 0.08, (args.queries, args.dim))).astype(np.float32)
    else:
        rows = rng.normal(0, 0.25, (args.rows, args.dim)).astype(np.float32)
        queries = rng.normal(0, 0.25, (args.queries, args.dim)).astype(np.float32)
    labels = np.arange(args.rows, dtype=np.uint64) << np.uint64(32)
    if metric == 2:
        queries = np.stack([hostapi.normalize_copy(q)[0] for q in queries])

    m = None
    if saved is None:
        t0 = time.perf_counter()
        m = hostapi.GpuHnswMap(metric, args.dim, args.rows, M=args.M, ef_construction=args.efc, multithread=args.build_threads > 0)
        m.add(rows, labels, threads=args.build_threads)
        build_s = time.perf_counter() - t0
        g = m.export_graph()
        if args.build_threads:   # internal ids follow arrival order: bring the corpus into the graph's order (label i << 32 = original row i)
            order = (g["labels"][:args.rows] >> np.uint64(32)).astype(np.int64)
            rows, labels = rows[order], labels[order]
    else:
        meta = saved["meta"]
        build_s = float(saved["build_seconds"])
        if "labels" in saved:   # a concurrently built graph: internal ids follow arrival order
            order = (saved["labels"][:args.rows] >> np.uint64(32)).astype(np.int64)
            rows, labels = rows[order], labels[order]
        g = dict(metric=metric, n=args.rows, dim=args.dim, M=args.M, maxM0=int(meta[4]), maxlevel=int(meta[5]), entry=int(meta[6]),
                 num_deleted=int(meta[7]), links0=saved["links0"], upper_off=saved["upper_off"], upper=saved["upper"], levels=saved["levels"],
                 labels=labels, deleted=saved["deleted"])

    # GPU search through the C-ABI in one batched call (the Map's SearchKnn is the nq = 1 case of the same entry point)
    inv = np.array([hostapi.l2_module(r) for r in rows], np.float32) if metric == 2 else None
    ix = capi.VectorIndex(metric, args.dim, args.rows)
    ix.upload_rows(0, rows, inv)
    ix.hnsw_attach_graph(g)
    ix.hnsw_search_knn(queries[:64], args.k, args.ef)   # warmup
    ix.hnsw_read_stats()
    ix.profile_enable(True)
    t0 = time.perf_counter()
    dist, row, cnt = ix.hnsw_search_knn(queries, args.k, args.ef)
    gpu_s = time.perf_counter() - t0
    launches, kernel_ms = ix.profile_read("hnsw")
    redo_launches, redo_ms = ix.profile_read("hnsw_redo")
    ix.profile_enable(False)
    evals, hops = ix.hnsw_read_stats()
    bytes_algo = evals * args.dim * 4 + hops * (1 + 2 * args.M) * 4
    # single-query latency (through the Map when it was built here, else the nq = 1 case of the same C-ABI entry)
    t0 = time.perf_counter()
    for q in queries[:32]:
        if m is not None:
            m.search_knn(q, args.k, args.ef)
        else:
            ix.hnsw_search_knn(q[None, :], args.k, args.ef)
    lat_ms = (time.perf_counter() - t0) / 32 * 1e3
    ix.hnsw_read_stats()

    stream = None
    if m is not None:   # a15: one streaming session, 10 batches of 10 (the planner's post-filter pattern), per-call latency
        sess = m.stream(queries[0], args.ef)
        t0 = time.perf_counter()
        got = 0
        for _ in range(10):
            d_, l_, ex_ = sess.next(10)
            got += len(d_)
        stream = {"batches": 10, "batch": 10, "ef": args.ef, "returned": got, "ms_per_continue": (time.perf_counter() - t0) / 10 * 1e3}
        sess.close()

    # exact ground truth on the GPU (fused scan / batched path)
    bf = capi.VectorIndex(metric, args.dim, args.rows)
    bf.upload_rows(0, rows, inv)
    tq = min(args.queries, 512)
    _, trow, _ = bf.search_knn(queries[:tq], args.k)
    recall = float(np.mean([len(set(trow[i].tolist()) & set(row[i, :int(cnt[i])].tolist())) / args.k for i in range(tq)]))

    out = {
        "workload": f"HNSW {args.metric} M={args.M} efC={args.efc} ef={args.ef} k={args.k}, {args.rows} x {args.dim} (scaled from BASELINE configs[2]), "
                    + (f"{args.clusters} gaussian clusters" if args.clusters else "i.i.d. gaussian"),
        "build_seconds_host": build_s, "build_threads": max(args.build_threads, 1),
        "gpu": {"queries": args.queries, "queries_per_sec": args.queries / gpu_s, "kernel_ms_total": kernel_ms, "launches": launches,
                "queries_per_sec_kernel_only": args.queries / ((kernel_ms + redo_ms) / 1e3) if kernel_ms else None,
                "redo_launches": redo_launches, "redo_ms": redo_ms,
                "map_single_query_latency_ms": lat_ms, "distance_evals_per_query": evals / args.queries, "hops_per_query": hops / args.queries,
                "roofline": {"bound": "hbm", "achieved": bytes_algo / (kernel_ms / 1e3) / 1e9 if kernel_ms else None, "peak": 8000.0, "unit": "GB/s",
                             "frac": bytes_algo / (kernel_ms / 1e3) / 1e9 / 8000.0 if kernel_ms else None,
                             "algorithmic_bytes": bytes_algo, "note": "random 3 KB row gathers; bytes = evals*D*4 + hops*(1+2M)*4"}},
        "recall_at_k_vs_exact": recall,
        "streaming_session": stream,
    }
    if saved is not None or args.build_threads:   # CPU baseline on the SAME graph: the restated engine (pinned equal to the reference's), 1 thread
        try:
            from oracle.pyoracle import Oracle, oracle_hnsw_search_knn
            orc = Oracle()
            g2 = dict(g)
            g2["vectors"] = rows
            nq = min(args.cpu_queries, args.queries)
            t0 = time.perf_counter()
            res = [oracle_hnsw_search_knn(orc, g2, queries[i], args.k, args.ef, inv) for i in range(nq)]
            cpu_s = time.perf_counter() - t0
            same = sum(int(np.array_equal(np.sort(labels[row[i, :int(cnt[i])]]), np.sort(res[i][1]))) for i in range(nq))
            out["cpu_baseline"] = {"kind": "port", "value": nq / cpu_s, "unit": "queries/s", "cores": 1, "sample": f"{nq} queries, same graph"}
            out["equal_to_reference_frac"] = same / nq
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
        text = json.dumps(out)
        print(text)
        if args.out:
            Path(args.out).write_text(text + "\n")
        return
    try:
        from oracle import pyoracle
        ref = pyoracle.ref_or_none()
        if ref is not None and ref.simd_level == 3:
            r = pyoracle.RefHnsw(ref, metric, args.dim, args.rows, M=args.M, ef_construction=args.efc)
            t0 = time.perf_counter()
            r.add(rows, labels)
            ref_build = time.perf_counter() - t0
            nq = min(args.cpu_queries, args.queries)
            t0 = time.perf_counter()
            res = [r.search_knn(queries[i], args.k, args.ef) for i in range(nq)]
            cpu_s = time.perf_counter() - t0
            same = 0
            for i in range(nq):
                c = int(cnt[i])
                gl = np.sort(labels[row[i, :c]])
                same += int(np.array_equal(gl, np.sort(res[i][1])))
            threads = min(os.cpu_count() or 1, 64)
            def worker(t):
                for j in range(8):
                    r.search_knn(queries[(t * 8 + j) % args.queries], args.k, args.ef)
            ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
            t0 = time.perf_counter()
            [t.start() for t in ths]
            [t.join() for t in ths]
            cpu_all_s = time.perf_counter() - t0
            out["cpu_baseline"] = {"kind": "reference", "value": nq / cpu_s, "unit": "queries/s", "cores": 1, "sample": f"{nq} queries, same graph",
                                   "all_cores": {"value": threads * 8 / cpu_all_s, "cores": threads}, "build_seconds": ref_build}
            out["equal_to_reference_frac"] = same / nq
            r.close()
    except Exception as e:  # the GPU numbers are still reported
        out["cpu_baseline"] = {"error": repr(e)}
    text = json.dumps(out)
    print(text)
    if args.out:
        Path(args.out).write_text(text + "\n")


if __name__ == "__main__":
    main()
