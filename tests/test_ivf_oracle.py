"""CPU: the DEFINITION the IVF-Flat index is held to (tests/test_gpu_ivf.py: "the exact search, in the engine's distance arithmetic, over the rows of
the nprobe nearest lists") against the REAL IVF backend of the reference — the patched FAISS vendored under cpp_src/vendor_subdirs/faiss,
compiled in place (oracle/_ref/libref_ivf.so) and driven the way IvfIndex drives it.  In this copy FAISS's fvec_L2sqr / fvec_inner_product call
the reference's own vector_dists functions (faiss/utils/distances.h:34-42), so with FAISS's trained state (centroids + lists) the restated
search must return FAISS's labels and DISTANCE BITS.  Training is not pinned (FAISS's random stream is not reproduced)."""
import os

import numpy as np
import pytest

from .conftest import lex_topk

os.environ.setdefault("RX_TARGET_INSTRUCTIONS", "avx512")


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def clustered(seed, n, d, clusters=48):
    rng = np.random.default_rng(seed)
    centres = rng.normal(0, 0.25, (clusters, d)).astype(np.float32)
    return (centres[rng.integers(0, clusters, n)] + rng.normal(0, 0.05, (n, d))).astype(np.float32)


@pytest.fixture(scope="module")
def faiss_ready(ref):   # `ref` skips when the reference tree / AVX-512 is missing
    from oracle import pyoracle
    if not pyoracle.ref_ivf_available():
        pytest.skip("oracle/_ref/libref_ivf.so not available")
    return pyoracle


def restated_ivf(oracle, metric, q, cent, lists_rows, rows, inv, nprobe):
    """-> (internal distances, row numbers) of every vector in the nprobe nearest lists; smaller internal distance = closer."""
    if metric == 0:
        coarse = oracle.dist_many(0, q, cent)
    else:
        coarse = oracle.dist_many(1, q, cent)                      # -ip
        if metric == 2:                                            # IndexFlatCosine quantiser: ip * 1/|centroid|
            coarse = coarse * oracle.l2_modules(cent)
    probe = lex_topk(coarse, nprobe)[1]
    cand = np.concatenate([lists_rows[int(l)] for l in probe]) if probe.size else np.empty(0, np.int64)
    if cand.size == 0:
        return np.empty(0, np.float32), cand
    return oracle.dist_many(metric, q, rows[cand], inv[cand] if inv is not None else None), cand


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_ivf_search_definition_equals_real_faiss(oracle, faiss_ready, metric):
    n, d, nlist = 9000, 48, 32
    rows = clustered(21, n, d)
    ids = np.arange(n, dtype=np.int64) * 5 + 11
    f = faiss_ready.RefIvf(metric, d, nlist, rows, ids)
    cent, lists = f.export()
    assert sum(len(l) for l in lists) == n
    row_of = {int(l): i for i, l in enumerate(ids)}
    lists_rows = [np.array([row_of[int(x)] for x in l], np.int64) for l in lists]
    inv = oracle.l2_modules(rows) if metric == 2 else None
    sign = 1.0 if metric == 0 else -1.0
    queries = clustered(22, 25, d)
    for q in queries:
        if metric == 2:
            q, _ = oracle.normalize_copy(q)            # IvfIndex::select normalises the key for cosine (ivf_index.cc:333-337)
        for nprobe in (1, 3, 8, nlist):
            dist, cand = restated_ivf(oracle, metric, q, cent, lists_rows, rows, inv, nprobe)
            for k in (1, 10, 60):
                fd, fl = f.search(q, k, nprobe)
                m = min(k, cand.size)
                wd, wpos = lex_topk(dist, m)
                got = np.lexsort((fl[:m], fd[:m] * sign))          # FAISS pops a heap: order both sides by (distance, label)
                want = np.lexsort((ids[cand[wpos]], wd))
                assert np.array_equal(fl[:m][got], ids[cand[wpos]][want]), (metric, nprobe, k)
                assert np.array_equal(bits((fd[:m] * sign)[got]), bits(wd[want]))
                assert np.all(fl[m:] == -1)
            # range search: L2 dist < radius, similarity > radius
            if cand.size > 30:
                radius_internal = float(np.sort(dist)[25])
                rd, rl = f.range_search(q, radius_internal * sign, nprobe)
                keep = dist < radius_internal
                assert sorted(rl.tolist()) == sorted(ids[cand[keep]].tolist())
                o1, o2 = np.argsort(rl, kind="stable"), np.argsort(ids[cand[keep]], kind="stable")
                assert np.array_equal(bits((rd * sign)[o1]), bits(dist[keep][o2]))
    # remove_ids: the vectors leave their lists; every list probed == exact search over the survivors
    victims = ids[np.random.default_rng(3).choice(n, 300, replace=False)]
    assert f.remove_ids(victims) == 300
    alive = ~np.isin(ids, victims)
    q = queries[0] if metric != 2 else oracle.normalize_copy(queries[0])[0]
    fd, fl = f.search(q, 20, nlist)
    wd, wpos = lex_topk(oracle.dist_many(metric, q, rows[alive], inv[alive] if inv is not None else None), 20)
    assert sorted(fl.tolist()) == sorted(ids[alive][wpos].tolist())
    f.close()
