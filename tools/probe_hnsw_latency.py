#!/usr/bin/env python3
"""Single-query HNSW latency, split into kernel time and everything around it (C-ABI level, nq = 1):
    python tools/probe_hnsw_latency.py [--rows 200000] [--dim 768] [--ef 128]"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
from reindexer_amd import capi, hostapi  # noqa: E402
from bench_hnsw import make_clustered  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=200_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--ab", action="store_true", help="after the default run: the same queries with the visited set in global memory / without the link prefetch")
    ap.add_argument("--threads", type=int, default=32)
    a = ap.parse_args()
    rows = make_clustered(a.rows + a.queries, a.dim, 2000, 7, 0)
    data, queries = rows[:a.rows], rows[a.rows:]
    m = hostapi.GpuHnswMap(2, a.dim, a.rows, M=16, ef_construction=200, multithread=True)
    t0 = time.perf_counter()
    m.add(data, np.arange(a.rows, dtype=np.uint64) << np.uint64(32), threads=a.threads)
    build_s = time.perf_counter() - t0
    m.search_knn(queries[0], a.k, a.ef)
    t0 = time.perf_counter()
    for q in queries:
        m.search_knn(q, a.k, a.ef)
    map_ms = (time.perf_counter() - t0) / a.queries * 1e3
    g = m.export_graph(with_views=True)
    inv = np.array(g["inv_norms"]) if g["inv_norms"] is not None else None
    ix = capi.VectorIndex(2, a.dim, a.rows)
    ix.upload_rows(0, np.array(g["vectors"]), inv)
    ix.hnsw_attach_graph(g)
    qn = np.stack([hostapi.normalize_copy(q)[0] for q in queries])
    ix.hnsw_search_knn(qn[:1], a.k, a.ef)
    ix.profile_enable(True)
    t0 = time.perf_counter()
    for q in qn:
        ix.hnsw_search_knn(q[None, :], a.k, a.ef)
    abi_ms = (time.perf_counter() - t0) / a.queries * 1e3
    n, ms = ix.profile_read("hnsw")
    evals, hops = ix.hnsw_read_stats()
    if a.ab:
        import os
        ab = {}
        ref = [ix.hnsw_search_knn(q[None, :], a.k, a.ef) for q in qn]
        for tag, env in (("lds_set", {}), ("global_set", {"RXGPU_HNSW_VISITED_LDS": "0"}), ("global_hash", {"RXGPU_HNSW_VISITED": "hash"}),
                         ("lds_no_prefetch", {"RXGPU_HNSW_PREFETCH": "0"}), ("lds_set_again", {})):
            for k_, v_ in env.items():
                os.environ[k_] = v_
            ix.hnsw_search_knn(qn[:1], a.k, a.ef)
            ix.profile_read("hnsw")
            best = None
            for rep in range(3):
                t0 = time.perf_counter()
                got = [ix.hnsw_search_knn(q[None, :], a.k, a.ef) for q in qn]
                dt = (time.perf_counter() - t0) / a.queries * 1e3
                best = dt if best is None else min(best, dt)
            n2, ms2 = ix.profile_read("hnsw")
            same = all(np.array_equal(x[1], y[1]) and np.array_equal(x[0].view(np.uint32), y[0].view(np.uint32)) for x, y in zip(got, ref))
            ab[tag] = {"c_abi_ms_per_query": best, "kernel_ms_per_query": ms2 / max(n2, 1), "same_results": bool(same)}
            for k_ in env:
                del os.environ[k_]
        print(json.dumps({"ab": ab}))
    print(json.dumps({"rows": a.rows, "build_s": build_s, "map_ms_per_query": map_ms, "c_abi_ms_per_query": abi_ms, "kernel_ms_per_query": ms / max(n, 1),
                      "launches": n, "evals_per_query": evals / (a.queries + 1), "hops_per_query": hops / (a.queries + 1),
                      "us_per_hop_kernel": ms / max(n, 1) * 1e3 / max(hops / (a.queries + 1), 1)}))


if __name__ == "__main__":
    main()
