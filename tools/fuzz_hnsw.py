#!/usr/bin/env python3
"""Randomised differential test of the HNSW engine (GpuHnswMap: host builder + GPU search / streaming) against the CPU restatement of the
reference's search (oracle/oracle_hnsw.c, pinned against the real engine).

    python tools/fuzz_hnsw.py --seconds 60 [--seed 1]

Every round draws a metric, dimension, corpus size, M / efConstruction, a set of deleted labels, then checks one-shot searches (random k, ef)
and whole streaming sessions (random batch plans) for exact equality of the returned (dist, label) sets."""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("RX_TARGET_INSTRUCTIONS", "avx512")

from oracle.pyoracle import Oracle, OracleHnswStream, oracle_hnsw_search_knn  # noqa: E402
from reindexer_amd import hostapi  # noqa: E402


def pairs(d, l):
    o = np.lexsort((l, d))
    return d[o].view(np.uint32), l[o]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    orc = Oracle()
    t_end = time.time() + args.seconds
    rounds = checks = 0
    while time.time() < t_end:
        metric = int(rng.integers(0, 3))
        d = int(rng.choice([3, 8, 17, 32, 64, 100, 128, 200, 512, 768]))
        n = int(rng.choice([1, 2, 5, 40, 300, 1500, 4000]))
        if d >= 512:
            n = min(n, 1500)
        M = int(rng.choice([4, 8, 16, 24]))
        efc = int(rng.choice([20, 50, 100, 200]))
        style = rng.choice(["gauss", "clustered", "dups"])
        if style == "gauss":
            rows = rng.normal(0, 0.25, (n, d)).astype(np.float32)
        elif style == "clustered":
            c = rng.normal(0, 1, (max(1, n // 30), d)).astype(np.float32)
            rows = (c[rng.integers(0, c.shape[0], n)] + rng.normal(0, 0.05, (n, d))).astype(np.float32)
        else:
            base = rng.normal(0, 1, (max(1, n // 10), d)).astype(np.float32)
            rows = base[rng.integers(0, base.shape[0], n)].copy()       # exact duplicates: equal distances everywhere
        if metric == 2:
            rows[np.all(rows == 0, axis=1)] = 1.0
        labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(int(rng.integers(0, 4)))
        m = hostapi.GpuHnswMap(metric, d, n + 4, M=M, ef_construction=efc)
        m.add(rows, labels)
        ndel = int(rng.integers(0, max(1, n // 4))) if n > 4 and rng.random() < 0.5 else 0
        for lab in labels[rng.choice(n, ndel, replace=False)] if ndel else []:
            m.mark_delete(lab)
        g = m.export_graph()
        g["vectors"] = rows
        inv = orc.l2_modules(rows) if metric == 2 else None
        for _ in range(6):
            q = (rows[rng.integers(0, n)] + rng.normal(0, 0.1, d)).astype(np.float32) if rng.random() < 0.6 else rng.normal(0, 0.3, d).astype(np.float32)
            if metric == 2:
                if not np.any(q):
                    q[0] = 1.0
                q, _ = orc.normalize_copy(q)
            k = int(rng.choice([1, 3, 10, 50, 200]))
            ef = int(rng.choice([0, k, k + 7, 64, 128, 500]))
            if ef and ef < k:
                ef = k
            gd, gl = m.search_knn(q, k, ef)
            wd, wl = oracle_hnsw_search_knn(orc, g, q, k, ef, inv)
            a, b = pairs(gd, gl), pairs(wd, wl)
            if not (np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])):
                print("MISMATCH search", dict(metric=metric, d=d, n=n, M=M, efc=efc, style=str(style), ndel=ndel, k=k, ef=ef, seed=args.seed, round=rounds))
                sys.exit(1)
            checks += 1
            sef = int(rng.choice([0, 4, 16, 100]))
            plan = [int(x) for x in rng.choice([1, 2, 5, 10, 64, 300, 2000], int(rng.integers(1, 7)))]
            gs, ws = m.stream(q, sef), OracleHnswStream(orc, g, q, sef, inv)
            for bsz in plan:
                gd, gl, gex = gs.next(bsz)
                wd, wl, wex = ws.next(bsz)
                a, b = pairs(gd, gl), pairs(wd, wl)
                if not (gex == wex and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])):
                    print("MISMATCH stream", dict(metric=metric, d=d, n=n, M=M, efc=efc, style=str(style), ndel=ndel, sef=sef, plan=plan, bsz=bsz,
                                                  seed=args.seed, round=rounds))
                    sys.exit(1)
                checks += 1
            gs.close()
            ws.close()
        m.close()
        rounds += 1
    print(f"hnsw fuzz ok: {rounds} rounds, {checks} checks, seed {args.seed}")


if __name__ == "__main__":
    main()
