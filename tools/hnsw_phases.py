#!/usr/bin/env python3
"""Where a single HNSW search spends its cycles (library built with RXGPU_HIP_DEFINES=-DRXGPU_HNSW_PHASES, run with RXGPU_HNSW_PHASES=1):
builds a graph over --rows clustered rows and runs --queries single-query calls through the C-ABI, team form and one-wavefront form."""
import argparse, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
from reindexer_amd import capi, hostapi
from bench_hnsw import make_clustered

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=300_000)
ap.add_argument("--queries", type=int, default=64)
a = ap.parse_args()
d, k, ef = 768, 10, 128
corpus = make_clustered(a.rows + a.queries, d, 2000, 20260924, 0)
rows, queries = corpus[:a.rows], corpus[a.rows:]
queries = np.stack([hostapi.normalize_copy(q)[0] for q in queries])
m = hostapi.GpuHnswMap(2, d, a.rows, M=16, ef_construction=200, multithread=True)
t0 = time.perf_counter()
m.add(rows, np.arange(a.rows, dtype=np.uint64) << np.uint64(32), threads=32)
print("build s", round(time.perf_counter() - t0, 1))
g = m.export_graph(with_views=True)
ix = capi.VectorIndex(2, d, a.rows)
ix.upload_rows(0, g["vectors"], g["inv_norms"])
ix.hnsw_attach_graph(g)
for team in ("4", "1"):
    os.environ["RXGPU_HNSW_TEAM"] = team
    ix.hnsw_search_knn(queries[:1], k, ef)
    ix.hnsw_read_stats()
    t0 = time.perf_counter()
    for q in queries:
        ix.hnsw_search_knn(q[None, :], k, ef)
    dt = (time.perf_counter() - t0) / len(queries)
    print("team", team, "ms per single-query call", round(dt * 1e3, 3), flush=True)
    ix.hnsw_read_stats()
