#!/usr/bin/env python3
"""Single-query HNSW latency through the C-ABI, variant against variant on ONE graph: the resident kernel (mailbox) and the team launch, each
with and without the link blocks that come along with a hop's rows (RXGPU_HNSW_NBL) and the look-ahead distance batches (RXGPU_HNSW_SPEC).  Every variant must return the same result sets (labels and distance
bits); hops / evaluations / distance trips come from the kernel's counters (RXGPU_HNSW_TRIPS=1 prints them).
    python tools/bench_hnsw_single.py [--rows 1000000] [--queries 256] [--threads 16] [--out f.json]"""
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
sys.path.insert(0, str(ROOT / "tools"))
from bench_hnsw import make_clustered  # noqa: E402
from reindexer_amd import capi, hostapi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--queries", type=int, default=256)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--ef", type=int, default=128)
ap.add_argument("--only", default=None, help="comma list of variant names")
ap.add_argument("--out", default=None)
a = ap.parse_args()
d, k, ef = a.dim, 10, a.ef
corpus = make_clustered(a.rows + a.queries, d, 2000, 20260924, 0)
rows, queries = corpus[:a.rows], corpus[a.rows:]
queries = np.stack([hostapi.normalize_copy(q)[0] for q in queries])
m = hostapi.GpuHnswMap(2, d, a.rows, M=16, ef_construction=200, multithread=True)
t0 = time.perf_counter()
m.add(rows, np.arange(a.rows, dtype=np.uint64) << np.uint64(32), threads=32)
print("build s", round(time.perf_counter() - t0, 1), flush=True)
g = m.export_graph(with_views=True)
out = {"rows": a.rows, "dim": d, "ef": ef, "queries": a.queries, "variants": {}}
base = None
os.environ["RXGPU_HNSW_TRIPS"] = "1"
VARIANTS = (("mailbox_plain", {}), ("mailbox_rows_page_aligned", {"RXGPU_ROW_ALIGN": "4096"}), ("mailbox_linkblocks", {"RXGPU_HNSW_NBL": "1"}), ("mailbox_lookahead", {"RXGPU_HNSW_SPEC": "1"}),
            ("launch_plain", {"RXGPU_HNSW_SERVER": "0"}), ("launch_linkblocks", {"RXGPU_HNSW_SERVER": "0", "RXGPU_HNSW_NBL": "1"}))
for name, env in VARIANTS:
    if a.only and name not in a.only.split(","):
        continue
    for kk in ("RXGPU_HNSW_SPEC", "RXGPU_HNSW_SERVER", "RXGPU_HNSW_NBL", "RXGPU_ROW_ALIGN"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    ix = capi.VectorIndex(2, d, a.rows)
    ix.upload_rows(0, g["vectors"], g["inv_norms"])
    ix.hnsw_attach_graph(g)
    for q in queries[:8]:
        ix.hnsw_search_knn(q[None, :], k, ef)
    ix.hnsw_read_stats()
    res = []
    t0 = time.perf_counter()
    for q in queries:
        res.append(ix.hnsw_search_knn(q[None, :], k, ef))
    ms = (time.perf_counter() - t0) / len(queries) * 1e3
    print(name, "single-query ms", round(ms, 4), flush=True)
    evals, hops = ix.hnsw_read_stats()
    # T threads, one query each at a time (python threads: the C call releases the GIL; ~20 us of interpreter per call)
    done = [0] * a.threads

    def worker(t):
        for j in range(64):
            ix.hnsw_search_knn(queries[(t * 64 + j) % len(queries)][None, :], k, ef)
            done[t] += 1

    th = [threading.Thread(target=worker, args=(t,)) for t in range(a.threads)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    qps = sum(done) / (time.perf_counter() - t0)
    served = ix.hnsw_server_stats()
    key = [tuple(sorted(zip(r[0][0, :int(r[2][0])].view(np.uint32).tolist(), r[1][0, :int(r[2][0])].tolist()))) for r in res]
    if base is None:
        base = key
    same = sum(int(x == y) for x, y in zip(key, base))
    out["variants"][name] = {"single_query_ms": ms, "threads": a.threads, "queries_per_sec_python_threads": qps, "evals_per_query": evals / len(queries),
                             "hops_per_query": hops / len(queries), "served_by_mailbox": served[0], "generations": served[1], "equal_to_first_variant": same / len(queries)}
    print(name, json.dumps(out["variants"][name]), flush=True)
    ix.close()
print(json.dumps(out))
if a.out:
    Path(a.out).write_text(json.dumps(out, indent=1))
