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
ap.add_argument("--reorder", action="store_true", help="also: the same graph with its nodes stored in BFS order (variant mailbox_bfs_order)")
ap.add_argument("--out", default=None)
a = ap.parse_args()
d, k, ef = a.dim, 10, a.ef
corpus, cluster_of = make_clustered(a.rows + a.queries, d, 2000, 20260924, 0, return_pick=True)
rows, queries = corpus[:a.rows], corpus[a.rows:]
queries = np.stack([hostapi.normalize_copy(q)[0] for q in queries])
m = hostapi.GpuHnswMap(2, d, a.rows, M=16, ef_construction=200, multithread=True)
t0 = time.perf_counter()
m.add(rows, np.arange(a.rows, dtype=np.uint64) << np.uint64(32), threads=32)
print("build s", round(time.perf_counter() - t0, 1), flush=True)
g = m.export_graph(with_views=True)
del corpus, rows   # (the builder holds the vectors; a reordered copy comes on top)
out = {"rows": a.rows, "dim": d, "ef": ef, "queries": a.queries, "variants": {}}


def bfs_order(g):
    """Breadth-first order of the level-0 graph from the entry point (unreached nodes appended): nodes that are neighbours in the graph become
    neighbours in memory.  -> new_of_old, old_of_new"""
    n, links0 = g["n"], g["links0"]
    seen = np.zeros(n, bool)
    order = np.empty(n, np.int64)
    at = 0
    frontier = np.array([int(g["entry"])], np.int64)
    seen[frontier] = True
    while frontier.size:
        order[at:at + frontier.size] = frontier
        at += frontier.size
        cnt = links0[frontier, 0].astype(np.int64)
        nb = links0[frontier, 1:].astype(np.int64)
        mask = np.arange(nb.shape[1])[None, :] < cnt[:, None]
        cand = nb[mask]
        cand = cand[~seen[cand]]
        _, first = np.unique(cand, return_index=True)     # keep discovery order
        cand = cand[np.sort(first)]
        seen[cand] = True
        frontier = cand
    rest = np.flatnonzero(~seen)
    order[at:at + rest.size] = rest
    new_of_old = np.empty(n, np.int64)
    new_of_old[order] = np.arange(n)
    return new_of_old, order


def reordered(g, order=None):
    """The same graph with its nodes renamed (BFS order, or the given old_of_new order) and every per-node array permuted accordingly: the
    traversal is the same walk over the same link lists (orders inside the lists kept), only the addresses change."""
    if order is None:
        new_of_old, old_of_new = bfs_order(g)
    else:
        old_of_new = np.asarray(order, np.int64)
        new_of_old = np.empty(g["n"], np.int64)
        new_of_old[old_of_new] = np.arange(g["n"])
    n = g["n"]
    l0 = np.ascontiguousarray(g["links0"][old_of_new]).copy()
    cnt = l0[:, 0].astype(np.int64)
    mask = np.arange(l0.shape[1] - 1)[None, :] < cnt[:, None]
    ids = l0[:, 1:]
    ids[mask] = new_of_old[ids[mask].astype(np.int64)].astype(np.uint32)
    # upper levels: blocks of (1 + M) words per (node, level), CSR by node
    uo_old = np.asarray(g["upper_off"], np.int64)
    per = (uo_old[1:] - uo_old[:-1])[old_of_new]
    uo_new = np.zeros(n + 1, np.uint64)
    uo_new[1:] = np.cumsum(per)
    upper_old = np.asarray(g["upper"]).reshape(-1, 1 + g["M"])
    upper_new = np.zeros_like(upper_old)
    has = np.flatnonzero(per > 0)
    for nn in has:   # few nodes have upper levels (1 / M of them): a loop is fine
        o = old_of_new[nn]
        blk = upper_old[uo_old[o]:uo_old[o + 1]].copy()
        for b in blk:
            c = int(b[0])
            b[1:1 + c] = new_of_old[b[1:1 + c].astype(np.int64)].astype(np.uint32)
        upper_new[int(uo_new[nn]):int(uo_new[nn + 1])] = blk
    r = dict(g)
    r.update(links0=l0, upper_off=uo_new, upper=upper_new, levels=np.asarray(g["levels"])[old_of_new], labels=np.asarray(g["labels"])[old_of_new],
             deleted=np.asarray(g["deleted"])[old_of_new], entry=int(new_of_old[int(g["entry"])]),
             vectors=np.ascontiguousarray(np.asarray(g["vectors"])[old_of_new]), inv_norms=np.ascontiguousarray(np.asarray(g["inv_norms"])[old_of_new]))
    return r, old_of_new


g_alt, old_of_new_alt = {}, {}


def make_alt(name):   # one reordered copy at a time (30 GB at 10M x 768)
    t0 = time.perf_counter()
    if name == "mailbox_bfs_order":
        g_alt[name], old_of_new_alt[name] = reordered(g)
    else:   # stored cluster by cluster (the synthetic corpus knows its clusters; a product would take the grouping from the graph's upper levels)
        by_cluster = np.argsort(cluster_of[(np.asarray(g["labels"]) >> np.uint64(32)).astype(np.int64)], kind="stable")
        g_alt[name], old_of_new_alt[name] = reordered(g, by_cluster)
    print(name, "reorder s", round(time.perf_counter() - t0, 1), flush=True)


base = None
os.environ["RXGPU_HNSW_TRIPS"] = "1"
VARIANTS = (("mailbox_plain", {}), ("mailbox_bfs_order", {}), ("mailbox_cluster_order", {}), ("mailbox_rows_page_aligned", {"RXGPU_ROW_ALIGN": "4096"}), ("mailbox_linkblocks", {"RXGPU_HNSW_NBL": "1"}), ("mailbox_lookahead", {"RXGPU_HNSW_SPEC": "1"}),
            ("launch_plain", {"RXGPU_HNSW_SERVER": "0"}), ("launch_linkblocks", {"RXGPU_HNSW_SERVER": "0", "RXGPU_HNSW_NBL": "1"}))
for name, env in VARIANTS:
    if a.only and name not in a.only.split(","):
        continue
    for kk in ("RXGPU_HNSW_SPEC", "RXGPU_HNSW_SERVER", "RXGPU_HNSW_NBL", "RXGPU_ROW_ALIGN"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    if name in ("mailbox_bfs_order", "mailbox_cluster_order"):
        if not a.reorder:
            continue
        make_alt(name)
    gg = g_alt.get(name, g)
    ix = capi.VectorIndex(2, d, a.rows)
    ix.upload_rows(0, gg["vectors"], gg["inv_norms"])
    ix.hnsw_attach_graph(gg)
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
    back = (lambda rows_, t=old_of_new_alt.get(name): t[rows_]) if name in old_of_new_alt else (lambda rows_: rows_)   # the renamed graph answers in its own row numbers
    key = [tuple(sorted(zip(r[0][0, :int(r[2][0])].view(np.uint32).tolist(), back(r[1][0, :int(r[2][0])].astype(np.int64)).tolist()))) for r in res]
    if base is None:
        base = key
    same = sum(int(x == y) for x, y in zip(key, base))
    out["variants"][name] = {"single_query_ms": ms, "threads": a.threads, "queries_per_sec_python_threads": qps, "evals_per_query": evals / len(queries),
                             "hops_per_query": hops / len(queries), "served_by_mailbox": served[0], "generations": served[1], "equal_to_first_variant": same / len(queries)}
    print(name, json.dumps(out["variants"][name]), flush=True)
    ix.close()
    g_alt.pop(name, None)
print(json.dumps(out))
if a.out:
    Path(a.out).write_text(json.dumps(out, indent=1))
