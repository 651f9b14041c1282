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


def check_topk_whatever_the_tie_rule(got_dist, got_labels, cand_dist, cand_ids, k):
    """A top-k answer checked without assuming how equal distances are ordered or which of several candidates AT the k-th distance were
    kept (FAISS decides both by scan order and id; a (distance, row) order decides them differently): the distance bits, place by place;
    every label's own distance at its place; no label twice; everything strictly better than the k-th distance present; -1 behind the
    last hit.  got_dist / cand_dist are internal distances (smaller = closer)."""
    cand_dist = np.asarray(cand_dist, np.float32)
    cand_ids = np.asarray(cand_ids)
    m = min(k, cand_dist.size)
    want = np.sort(cand_dist, kind="stable")[:m]
    assert np.array_equal(np.asarray(got_dist[:m], np.float32).view(np.uint32), want.view(np.uint32))
    assert np.all(np.asarray(got_labels[m:]) == -1)
    got = [int(x) for x in got_labels[:m]]
    assert len(set(got)) == m
    where = {int(i): j for j, i in enumerate(cand_ids.tolist())}
    for j, lab in enumerate(got):
        assert lab in where and cand_dist[where[lab]].view(np.uint32) == want[j].view(np.uint32), (j, lab)
    if m:
        assert set(cand_ids[cand_dist < want[m - 1]].tolist()) <= set(got)


def test_tie_agnostic_checker_accepts_both_rules_and_rejects_wrong_answers():
    rng = np.random.default_rng(0)
    dist = rng.integers(0, 12, 200).astype(np.float32) / 4         # many ties
    ids = rng.permutation(1000)[:200].astype(np.int64)
    for k in (1, 7, 50, 200, 300):
        for l2 in (True, False):
            d, l = faiss_topk(dist, ids, k, l2)
            pad = max(0, k - len(l))
            check_topk_whatever_the_tie_rule(np.concatenate([d, np.full(pad, np.inf, np.float32)]), np.concatenate([l, np.full(pad, -1)]), dist, ids, k)
        o = np.lexsort((ids, dist))[:k]                              # the (distance, id) order
        pad = max(0, k - len(o))
        check_topk_whatever_the_tie_rule(np.concatenate([dist[o], np.full(pad, np.inf, np.float32)]), np.concatenate([ids[o], np.full(pad, -1)]), dist, ids, k)
    d, l = faiss_topk(dist, ids, 7, True)
    bad = l.copy()
    bad[0] = ids[np.argmax(dist)]                                    # a label whose distance is not the one reported
    with pytest.raises(AssertionError):
        check_topk_whatever_the_tie_rule(d, bad, dist, ids, 7)
    with pytest.raises(AssertionError):
        check_topk_whatever_the_tie_rule(d, np.concatenate([l[:6], l[:1]]), dist, ids, 7)   # a label twice
    with pytest.raises(AssertionError):
        check_topk_whatever_the_tie_rule(d + 1, l, dist, ids, 7)     # wrong distances


def faiss_topk(dist, ids, k, l2):
    """What IndexIVFFlat's scanner leaves in its heap when it meets the candidates in THIS order (the probed lists in coarse order, every
    list in storage order): a candidate enters only if it is strictly better than the heap top (IndexIVFFlat.cpp scan_codes,
    `if (C::cmp(simi[0], dis))`), the top is the worst entry by (distance, id) for L2 (CMax) resp. by (similarity, id) for inner product /
    cosine (CMin: among equal similarities the smallest id leaves first) — heap_replace_top orders with cmp2 (utils/Heap.h:112-150) — and
    heap_reorder lists ties by id ascending (L2) resp. descending.  `dist` are internal distances (smaller = closer)."""
    import heapq
    sg = 1 if l2 else -1
    heap = []          # (-dist, -sg * id): heapq's minimum is the FAISS heap top
    for d, i in zip(np.asarray(dist, np.float32).tolist(), np.asarray(ids).tolist()):
        if len(heap) < k:
            heapq.heappush(heap, (-d, -sg * i))
        elif -heap[0][0] > d:
            heapq.heapreplace(heap, (-d, -sg * i))
    out = sorted((-a, -b) for a, b in heap)
    return np.array([o[0] for o in out], np.float32), np.array([sg * o[1] for o in out], np.int64)


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_faiss_tie_rule_on_duplicate_heavy_data(oracle, faiss_ready, metric):
    """A third of the vectors are exact copies of others: equal distances everywhere, at the k-th place in particular.  The restated
    definition with the scanner's tie rule must give FAISS's labels in FAISS's order, also after remove_ids reshuffled the lists (the last
    entry of a list moves into the hole)."""
    rng = np.random.default_rng(40 + metric)
    n, d, nlist = 3000, 12, 16
    base = clustered(31 + metric, n, d)
    rows = np.where((rng.random(n) < 0.35)[:, None], base[rng.integers(0, n, n)], base).astype(np.float32)
    ids = rng.permutation(n * 3)[:n].astype(np.int64)
    f = faiss_ready.RefIvf(metric, d, nlist, rows, ids)
    sign = 1.0 if metric == 0 else -1.0
    inv = oracle.l2_modules(rows) if metric == 2 else None
    alive = np.ones(n, bool)
    boundary_ties = 0
    for round_ in range(2):
        cent, lists = f.export()
        row_of = {int(l): i for i, l in enumerate(ids)}
        lists_rows = [np.array([row_of[int(x)] for x in l], np.int64) for l in lists]
        for _ in range(40):
            q = (rows[rng.integers(0, n)] + rng.normal(0, 0.02, d)).astype(np.float32)
            if metric == 2:
                q, _ = oracle.normalize_copy(q)
            nprobe, k = int(rng.integers(1, nlist + 1)), int(rng.choice([1, 3, 10, 50]))
            dist, cand = restated_ivf(oracle, metric, q, cent, lists_rows, rows, inv, nprobe)
            wd, wl = faiss_topk(dist, ids[cand], k, metric == 0)
            fd, fl = f.search(q, k, nprobe)
            m = min(k, cand.size)
            assert np.array_equal(fl[:m], wl[:m]), (metric, round_, nprobe, k)
            assert np.array_equal(bits(fd[:m] * sign), bits(wd[:m])) and np.all(fl[m:] == -1)
            sd = np.sort(dist)
            boundary_ties += int(cand.size > k and sd[k - 1] == sd[k])
        victims = ids[rng.choice(np.flatnonzero(alive), 400, replace=False)]
        assert f.remove_ids(victims) == 400
        alive &= ~np.isin(ids, victims)
    assert boundary_ties > 10
    f.close()


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
