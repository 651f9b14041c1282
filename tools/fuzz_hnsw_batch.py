#!/usr/bin/env python3
"""Randomised differential test of the BATCH paths of the HNSW search (rxgpu_hnsw_search_knn) against each other: whatever the visited
set (bitset, hash set in HBM, hash set in LDS), the size of an in-kernel restart's heap area, helper workgroups on or off, the link
prefetch — the returned (dist, row) sets of a batch must be the same, bit for bit.  The reference for a round is the most conservative
configuration (bitset, no helpers, default areas), itself pinned to the restated engine by tests/test_gpu_hnsw*.py and tools/fuzz_hnsw.py.

    python tools/fuzz_hnsw_batch.py --seconds 120 [--seed 1]

Every round draws a metric, a dimension (32 / 128 / 768: the latency form and the LDS set exist for 768), a corpus (continuous, or on a
small integer grid with every row four times: equal distances, restarts), M / efConstruction, a batch size on either side of the helper
threshold (2048), k and ef; then a handful of environment combinations."""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from reindexer_amd import capi, hostapi  # noqa: E402

VARIANTS = [
    {},                                                                        # the defaults of the launch (size rules decide)
    {"RXGPU_HNSW_VISITED": "hash"},
    {"RXGPU_HNSW_VISITED": "hash", "RXGPU_HNSW_VISITED_LOG2": "8"},            # tiny sets: overflows -> helpers / LDS re-run / global tiers
    {"RXGPU_HNSW_RESTART_CAND": "6"},                                          # every restart overflows
    {"RXGPU_HNSW_RESTART_CAND": "6", "RXGPU_HNSW_HELPER": "0"},
    {"RXGPU_HNSW_RESTART_CAND": "0"},                                          # flagged searches go back to the host as ties
    {"RXGPU_HNSW_PREFETCH": "0"},
    {"RXGPU_HNSW_VISITED_LDS": "0"},
    {"RXGPU_HNSW_SPLIT_UPLOAD": "0"},                                          # one upload, one launch for batches >= 8192
    {"RXGPU_HNSW_SORTED": "0"},                                                # heaps only
    {"RXGPU_HNSW_SORTED": "0", "RXGPU_HNSW_LDS_CAND_CAP": "8", "RXGPU_HNSW_GCAND_CAP": "64"},   # heap overflow -> both global tiers
]
KEYS = sorted({k for v in VARIANTS for k in v} | {"RXGPU_HNSW_VISITED", "RXGPU_HNSW_HELPER"})


def run(ix, queries, k, ef, env):
    for key in KEYS:
        os.environ.pop(key, None)
    os.environ.update(env)
    dist, row, cnt = ix.hnsw_search_knn(queries, k, ef)
    return dist.copy(), row.copy(), cnt.copy()


def same(a, b, nq):
    if not np.array_equal(a[2], b[2]):
        return False
    for q in range(nq):
        c = int(a[2][q])
        oa, ob = np.lexsort((a[1][q, :c], a[0][q, :c])), np.lexsort((b[1][q, :c], b[0][q, :c]))
        if not (np.array_equal(a[1][q, :c][oa], b[1][q, :c][ob]) and np.array_equal(a[0][q, :c][oa].view(np.uint32), b[0][q, :c][ob].view(np.uint32))):
            return False
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    rounds = checks = 0
    while time.time() < t_end:
        metric = int(rng.integers(0, 3))
        d = int(rng.choice([32, 128, 768], p=[0.4, 0.3, 0.3]))
        n = int(rng.integers(1500, 5000 if d == 768 else 9000))
        grid = rng.random() < 0.5
        if grid:
            base = rng.integers(-2, 3, size=((n + 3) // 4, d)).astype(np.float32)
            base[np.all(base == 0, axis=1)] = 1.0
            rows = np.ascontiguousarray(np.repeat(base, 4, axis=0)[rng.permutation(4 * base.shape[0])[:n]])
        else:
            rows = rng.normal(0, 0.25, size=(n, d)).astype(np.float32)
        M = int(rng.choice([4, 8, 16]))
        m = hostapi.GpuHnswMap(metric, d, n, M=M, ef_construction=int(rng.choice([20, 60])))
        m.add(rows, np.arange(n, dtype=np.uint64) << np.uint64(32))
        if rng.random() < 0.3:
            for v in rng.choice(n, n // 5, replace=False):
                m.mark_delete(int(v) << 32)
        g = m.export_graph(with_views=True)
        inv = np.array(g["inv_norms"]) if g["inv_norms"] is not None else None
        ix = capi.VectorIndex(metric, d, n)
        ix.upload_rows(0, np.array(g["vectors"]), inv)
        ix.hnsw_attach_graph(g)
        for _ in range(2):
            nq = int(rng.choice([1, 17, 300, 2048, 2500, 8200], p=[0.15, 0.15, 0.2, 0.2, 0.2, 0.1]))
            if grid:
                queries = rng.integers(-2, 3, size=(nq, d)).astype(np.float32)
            else:
                queries = rng.normal(0, 0.25, size=(nq, d)).astype(np.float32)
            if metric == 2:
                queries = np.stack([hostapi.normalize_copy(q)[0] for q in queries])
            k = int(rng.choice([1, 10, 50]))
            ef = int(rng.choice([k, 16, 64, 128, 200, 300]))
            ref = run(ix, queries, k, ef, {"RXGPU_HNSW_VISITED": "bitset", "RXGPU_HNSW_HELPER": "0"})
            for env in VARIANTS:
                got = run(ix, queries, k, ef, env)
                checks += 1
                if not same(got, ref, nq):
                    print("MISMATCH", dict(seed=a.seed, round=rounds, metric=metric, d=d, n=n, grid=grid, M=M, nq=nq, k=k, ef=ef, env=env), flush=True)
                    sys.exit(1)
        ix.close()
        m.close()
        rounds += 1
    for key in KEYS:
        os.environ.pop(key, None)
    print(f"fuzz_hnsw_batch: {rounds} rounds, {checks} batch comparisons, all identical")


if __name__ == "__main__":
    main()
