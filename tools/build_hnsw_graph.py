#!/usr/bin/env python3
"""Build an HNSW graph on the host (the product's builder, link-for-link the reference's graph) and save it for tools/bench_hnsw.py --graph.

    python tools/build_hnsw_graph.py --rows 1000000 --out gpurun_in/hnsw_1m.npz

Needs no GPU.  The corpus is regenerated from the same seed by the bench, so only the links travel."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import os  # noqa: E402

os.environ.setdefault("RXGPU_NO_TORCH", "1")
from reindexer_amd import capi, hostapi  # noqa: E402


def corpus(rows, dim, clusters, queries, seed=20260924):
    rng = np.random.default_rng(seed)
    if clusters:
        centres = rng.normal(0, 0.25, (clusters, dim)).astype(np.float32)
        r = (centres[rng.integers(0, clusters, rows)] + rng.normal(0, 0.08, (rows, dim))).astype(np.float32)
        q = (centres[rng.integers(0, clusters, queries)] + rng.normal(0, 0.08, (queries, dim))).astype(np.float32)
    else:
        r = rng.normal(0, 0.25, (rows, dim)).astype(np.float32)
        q = rng.normal(0, 0.25, (queries, dim)).astype(np.float32)
    return r, q


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=4096)
    ap.add_argument("--clusters", type=int, default=2000)
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--efc", type=int, default=200)
    ap.add_argument("--metric", default="cosine")
    ap.add_argument("--threads", type=int, default=0, help="concurrent construction with this many inserting threads (0: the sequential graph)")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    metric = capi.METRICS[args.metric]
    rows, _ = corpus(args.rows, args.dim, args.clusters, args.queries)
    labels = np.arange(args.rows, dtype=np.uint64) << np.uint64(32)
    g = hostapi.HnswGraph(metric, args.dim, args.rows, M=args.M, ef_construction=args.efc)
    t0 = time.perf_counter()
    step = 20000
    for a in range(0, args.rows, step):
        g.add(rows[a:a + step], labels[a:a + step], threads=args.threads)
        el = time.perf_counter() - t0
        print(f"{a + step}/{args.rows} rows, {el:.0f} s, {el / (a + step) * 1e3:.2f} ms/insert", flush=True)
    build_s = time.perf_counter() - t0
    e = g.export()
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    if args.threads:
        print("note: a concurrent build assigns internal ids in arrival order; `labels` (row << 32) is saved with the graph")
    np.savez(args.out, labels=e["labels"], links0=e["links0"], upper_off=e["upper_off"], upper=e["upper"], levels=e["levels"], deleted=e["deleted"],
             meta=np.array([e["metric"], e["n"], e["dim"], e["M"], e["maxM0"], e["maxlevel"], e["entry"], e["num_deleted"], args.clusters, args.efc], np.int64),
             build_seconds=np.float64(build_s))
    print("saved", args.out, f"{build_s:.0f} s")


if __name__ == "__main__":
    main()
