#!/usr/bin/env python3
"""What random 3 KB row gathers can reach on this GPU: the practical roof for the HNSW search.

    python tools/bench_gather.py --rows 1000000 --dim 768 --evals 3400000

Uses rxgpu_distances (knn_distances: one 16-lane group per listed row, exact distance, 4 bytes out per row) on a uniformly random row list —
the access pattern of the HNSW kernel without its dependent hops — and on a sorted list for comparison."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from reindexer_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--evals", type=int, default=3_400_000)
    args = ap.parse_args()
    rng = np.random.default_rng(1)
    rows = rng.normal(0, 0.25, (args.rows, args.dim)).astype(np.float32)
    q = rng.normal(0, 0.25, args.dim).astype(np.float32)
    out = {}
    with capi.VectorIndex("ip", args.dim, args.rows) as ix:
        ix.upload_rows(0, rows)
        for name, ids in (("random", rng.integers(0, args.rows, args.evals).astype(np.uint32)),
                          ("sorted", np.sort(rng.integers(0, args.rows, args.evals)).astype(np.uint32))):
            ix.distances(q, ids[:1000])
            ix.profile_enable(True)
            t0 = time.perf_counter()
            for _ in range(3):
                ix.distances(q, ids)
            wall = (time.perf_counter() - t0) / 3
            n, ms = ix.profile_read("distances")
            ix.profile_enable(False)
            kms = ms / max(n, 1) if n else None
            out[name] = {"rows_gathered": args.evals, "bytes": args.evals * args.dim * 4, "kernel_ms": kms, "wall_ms": wall * 1e3,
                         "GBps_kernel": args.evals * args.dim * 4 / (kms / 1e3) / 1e9 if kms else None}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
