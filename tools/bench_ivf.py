#!/usr/bin/env python3
"""IVF-Flat on the GPU engines (first cut): build (k-means on the matrix-core batch path) and search at several nprobe.

    python tools/bench_ivf.py --rows 1000000 --dim 768 --nlist 1024 --queries 200 [--out profiles/r1_ivf_1m.json]

Per query: coarse quantiser (KNN over the centroids) -> host merge of the probed lists' rows -> row-list scan (knn_scan_subset).  Recall@k is
measured against the exact result (the same index with every list probed, which the tests pin to the brute-force oracle).  There is no CPU
line: the reference's IVF backend is FAISS, which cannot be built in this image (BLAS)."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from reindexer_amd import capi, hostapi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nlist", type=int, default=1024)
    ap.add_argument("--clusters", type=int, default=2000)
    ap.add_argument("--queries", type=int, default=200)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="cosine")
    ap.add_argument("--nprobe", default="1,4,16,64,128,256")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    metric = capi.METRICS[args.metric]
    rng = np.random.default_rng(20260924)
    centres = rng.normal(0, 0.25, (args.clusters, args.dim)).astype(np.float32)
    rows = (centres[rng.integers(0, args.clusters, args.rows)] + rng.normal(0, 0.08, (args.rows, args.dim))).astype(np.float32)
    queries = (centres[rng.integers(0, args.clusters, args.queries)] + rng.normal(0, 0.08, (args.queries, args.dim))).astype(np.float32)
    ids = np.arange(args.rows, dtype=np.int64)
    ivf = hostapi.GpuIvfFlat(metric, args.dim, args.nlist)
    t0 = time.perf_counter()
    step = 100_000
    for a in range(0, args.rows, step):
        ivf.add_with_ids(rows[a:a + step], ids[a:a + step])
    add_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    ivf.train()
    train_s = time.perf_counter() - t0
    sizes = ivf.list_sizes()
    ivf.search(queries[0], args.k, nprobe=1)
    exact = [ivf.search(q, args.k, nprobe=args.nlist)[1] for q in queries]
    out = {"workload": f"IVF-Flat {args.metric}, {args.rows} x {args.dim}, nlist={args.nlist}, k={args.k}, {args.clusters} gaussian clusters",
           "arch": capi.device_arch(0), "add_seconds": add_s,
           "train_seconds": train_s, "train": "k-means 10 iterations over min(rows, 256 * nlist) points + assignment of every row, 256 points per device call",
           "list_sizes": {"min": int(sizes.min()), "median": float(np.median(sizes)), "max": int(sizes.max()), "empty": int((sizes == 0).sum())},
           "nprobe": []}
    for nprobe in [int(x) for x in args.nprobe.split(",")]:
        t0 = time.perf_counter()
        got = [ivf.search(q, args.k, nprobe=nprobe)[1] for q in queries]
        dt = (time.perf_counter() - t0) / args.queries
        recall = float(np.mean([len(set(g.tolist()) & set(e.tolist())) / args.k for g, e in zip(got, exact)]))
        scanned = float(np.mean([ivf.probed_rows(q, nprobe).size for q in queries[:32]]))
        out["nprobe"].append({"nprobe": nprobe, "ms_per_query": dt * 1e3, "queries_per_sec": 1.0 / dt, "recall_at_k_vs_exact": recall,
                              "rows_scanned_avg": scanned, "fraction_of_corpus": scanned / args.rows})
    t0 = time.perf_counter()
    for q in queries[:32]:
        ivf.search(q, args.k, nprobe=args.nlist)
    out["all_lists_ms_per_query"] = (time.perf_counter() - t0) / 32 * 1e3
    # the queries of one call side by side (GpuIvfFlat::SearchBatch: a few host threads, every search on its own stream) and the range form
    ivf.search_batch(queries[:8], args.k, nprobe=16)
    t0 = time.perf_counter()
    bd, bl = ivf.search_batch(queries, args.k, nprobe=16)
    dt = (time.perf_counter() - t0) / args.queries
    one = [ivf.search(q, args.k, nprobe=16) for q in queries]
    out["batch_nprobe16"] = {"ms_per_query": dt * 1e3, "queries_per_sec": 1.0 / dt,
                             "identical_to_single_searches": bool(all(np.array_equal(bl[i], one[i][1]) and
                                                                      np.array_equal(bd[i].view(np.uint32), one[i][0].view(np.uint32)) for i in range(args.queries)))}
    kd, _ = ivf.search(queries[0], 50, nprobe=16)
    t0 = time.perf_counter()
    hits = 0
    for q in queries[:64]:
        kd, _ = ivf.search(q, 50, nprobe=16)
        rd, _ = ivf.range_search(q, float(kd[40]), nprobe=16)
        hits += rd.size
    out["range_nprobe16"] = {"ms_per_knn50_plus_range_query": (time.perf_counter() - t0) / 64 * 1e3, "hits_avg": hits / 64,
                             "note": "radius = the 41st best distance of the same probe: the device lists are scanned by rxgpu_search_range_lists"}
    ivf.close()
    text = json.dumps(out)
    print(text)
    if args.out:
        Path(args.out).write_text(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
