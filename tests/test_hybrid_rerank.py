"""CPU: hybrid FT + KNN rank fusion (reindexer_amd/host/hybrid_rerank.h) vs a direct restatement of the reference's
MergerRankedImpl (cpp_src/core/nsselecter/selectiteratorcontainer.cc:1343-1423), RanksHolder::InitRRFPositions
(ranks_holder.h:61-76) and the rerankers (core/sorting/reranker.h:11-39) — the reference's own hybrid tests
(gtests/tests/unit/hybrid.cc:119-143) check orderings by recomputing these formulas from separate FT and KNN queries."""
import numpy as np
import pytest


def rrf_positions(ranks_desc):
    pos, p, last = [], 1, ranks_desc[0] if len(ranks_desc) else 0
    for i, r in enumerate(ranks_desc):
        if r < last:
            last, p = r, i + 1
        pos.append(p)
    return pos


def restated(kind, params, knn_ids, knn_ranks, ft_ids, ft_ranks, union, desc, metric):
    f32 = np.float32
    merged = {}
    ft_index = {int(i): n for n, i in enumerate(ft_ids)}
    if kind == "rrf":
        c = params[0]
        order = sorted(range(len(ft_ids)), key=lambda i: -ft_ranks[i])
        ps = rrf_positions([ft_ranks[i] for i in order])
        ft_pos = {order[i]: ps[i] for i in range(len(order))}
        last, kpos = (knn_ranks[0] if len(knn_ranks) else 0), 1
        added = set()
        for i, (id_, r) in enumerate(zip(knn_ids, knn_ranks)):
            if (last < r) if metric == 0 else (last > r):
                last, kpos = r, i + 1
            id_ = int(id_)
            if id_ in ft_index:
                merged.setdefault(id_, f32(1.0 / (c + kpos) + 1.0 / (c + ft_pos[ft_index[id_]])))
                added.add(ft_index[id_])
            elif union:
                merged.setdefault(id_, f32(1.0 / (c + kpos)))
        if union:
            for n, id_ in enumerate(ft_ids):
                if n not in added:
                    merged.setdefault(int(id_), f32(1.0 / (c + ft_pos[n])))
    else:
        kk, kd, kf, fd, c = params
        added = set()
        for id_, r in zip(knn_ids, knn_ranks):
            id_ = int(id_)
            if id_ in ft_index:
                merged.setdefault(id_, f32(kk * float(r) + kf * float(ft_ranks[ft_index[id_]]) + c))
                added.add(ft_index[id_])
            elif union:
                merged.setdefault(id_, f32(kk * float(r) + kf * fd + c))
        if union:
            for n, id_ in enumerate(ft_ids):
                if n not in added:
                    merged.setdefault(int(id_), f32(kk * kd + kf * float(ft_ranks[n]) + c))
    items = sorted(merged.items(), key=lambda t: ((-t[1] if desc else t[1]), t[0]))
    return np.array([i for i, _ in items], np.int32), np.array([r for _, r in items], np.float32)


@pytest.mark.parametrize("kind,params", [("rrf", [60.0]), ("rrf", [1.0]), ("linear", [0.7, 0.1, 0.3, 5.0, 2.0])])
@pytest.mark.parametrize("union", [False, True])
@pytest.mark.parametrize("metric", [0, 1])
def test_merge_ranked_matches_restatement(kind, params, union, metric):
    from reindexer_amd import hostapi
    rng = np.random.default_rng(hash((kind, union, metric)) % 1000)
    for _ in range(20):
        nk, nf = int(rng.integers(0, 60)), int(rng.integers(0, 200))
        knn_ids = rng.choice(500, nk, replace=False).astype(np.int32)
        kr = np.sort(rng.integers(0, 12, nk).astype(np.float32))          # many equal ranks => shared RRF positions
        knn_ranks = kr if metric == 0 else kr[::-1].copy()
        ft_ids = np.sort(rng.choice(500, nf, replace=False)).astype(np.int32)
        ft_ranks = rng.integers(5, 256, nf).astype(np.float32)
        for desc in (True, False):
            wi, wr = restated(kind, params, knn_ids, knn_ranks, ft_ids, ft_ranks, union, desc, metric)
            gi, gr = hostapi.merge_ranked(kind, params, knn_ids, knn_ranks, ft_ids, ft_ranks, union=union, desc=desc, metric=metric)
            assert np.array_equal(gi, wi) and np.array_equal(gr.view(np.uint32), wr.view(np.uint32))


@pytest.mark.parametrize("kind,params", [("rrf", [60.0]), ("linear", [0.7, 0.1, -0.3, 5.0, 2.0]), ("linear", [1.0, 0.0, 1.0, 0.0, 0.0])])
@pytest.mark.parametrize("union", [False, True])
def test_merge_ranked_large_and_ft_order_entry(kind, params, union):
    """Sizes of BASELINE configs[4] (thousands of FT hits, ids up to 5M, k = 100): the radix-sorted paths against the restatement, and the
    entry that takes the FT result as the merger returns it (best rank first; also an arbitrary order) against the id-ordered entry."""
    from reindexer_amd import hostapi
    rng = np.random.default_rng(11)
    for nf in (0, 1, 3000, 20000):
        nk = 100
        ft_ids = np.sort(rng.choice(5_000_000, nf, replace=False)).astype(np.int32)
        ft_ranks = rng.integers(1, 256, nf).astype(np.float32)
        knn_ids = rng.choice(5_000_000, nk, replace=False).astype(np.int32)
        if nf:
            knn_ids[:40] = ft_ids[rng.choice(nf, 40, replace=nf < 40)]
            knn_ids = np.unique(knn_ids)[: nk]
            rng.shuffle(knn_ids)
        knn_ranks = np.sort(rng.random(knn_ids.size).astype(np.float32))[::-1].copy()
        for desc in (True, False):
            wi, wr = restated(kind, params, knn_ids, knn_ranks, ft_ids, ft_ranks, union, desc, 1)
            gi, gr = hostapi.merge_ranked(kind, params, knn_ids, knn_ranks, ft_ids, ft_ranks, union=union, desc=desc, metric=1)
            assert np.array_equal(gi, wi) and np.array_equal(gr.view(np.uint32), wr.view(np.uint32))
            by_rank = np.argsort(-ft_ranks, kind="stable")
            for order in (by_rank, rng.permutation(nf)):
                oi, orr = hostapi.merge_ranked(kind, params, knn_ids, knn_ranks, ft_ids[order], ft_ranks[order], union=union, desc=desc,
                                               metric=1, ft_order="rank")
                assert np.array_equal(oi, wi) and np.array_equal(orr.view(np.uint32), wr.view(np.uint32))


def test_merge_ranked_ft_order_rejects_duplicate_ids():
    from reindexer_amd import hostapi
    with pytest.raises(Exception):
        hostapi.merge_ranked("rrf", [60.0], np.array([1], np.int32), np.array([1.0], np.float32), np.array([5, 5], np.int32),
                             np.array([9.0, 3.0], np.float32), union=True, ft_order="rank")
