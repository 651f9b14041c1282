"""-m gpu: the launch train for SPARSELY hit document ranges (reindexer_amd/csrc/ft_sparse.hip: one wavefront per (query, range), one bitmap
per sub-term in LDS, ranking after the merge slots are known) against the restated Merger::Merge (itself pinned to the real ft::Merger,
tests/test_bm25_oracle.py) AND against the dense train (ft_merge.hip) on the same queries.

Bar: the merged documents IN MERGE ORDER, raw rank bits, fields, terms counters / uint8 ranks and the preselect flag — for OR / AND / NOT
terms, Simple() queries with the mergeLimit cut, the preselect phase with ties at the threshold (kept in document order across ranges),
removed and excluded documents, batches that mix both trains, and the resident (hybrid) form.  Every case asserts that the sparse train
really ran (rxgpu_ft_read_train_stats), and that queries it must not take (unequal field boosts, more than 16 sub-terms) go to the dense
train with the same result."""
import numpy as np
import pytest

from oracle.pyoracle import FtOracle, make_fpos
from .test_bm25_oracle import MULTI_CASES, _multi_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ft(oracle):
    return FtOracle(oracle)


@pytest.fixture()
def hostapi(rxgpu):
    from reindexer_amd import hostapi as h
    h.lib()
    yield h
    h.set_ft_train_mode(-1)


def same(a, b):
    return (np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2])
            and np.array_equal(a[3], b[3]) and a[4] == b[4])


class Pair:
    """Two mergers over the same index: one only ever runs the dense train, the other the sparse one whenever the query is eligible — a train
    never finds the other's result of the same query in its output buffers."""

    def __init__(self, hostapi, nf, words, avg, removed, store):
        self.ms = []
        for _ in range(2):
            m = hostapi.GpuFtMerger(nf)
            m.set_docs(words, avg, removed)
            for s in store:
                m.set_word_fpos(s["word"], s)
            self.ms.append(m)
        self.dense, self.sparse = self.ms

    def close(self):
        for m in self.ms:
            m.close()


def load(hostapi, nf, words, avg, removed, store):
    return Pair(hostapi, nf, words, avg, removed, store)


def both_trains(hostapi, m, cfg, gterms, exc, expect_sparse=True):
    hostapi.set_ft_train_mode(1)
    m.sparse.read_train_stats()
    s = m.sparse.merge_query(cfg, gterms, exc, sort_by_rank=False)
    assert m.sparse.read_train_stats() == ((0, 1) if expect_sparse else (1, 0))
    hostapi.set_ft_train_mode(0)
    m.dense.read_train_stats()
    d = m.dense.merge_query(cfg, gterms, exc, sort_by_rank=False)
    assert m.dense.read_train_stats() == (1, 0)
    return d, s


@pytest.mark.parametrize("seed,nf,total,limit,ops,arr,fbs", [c for c in MULTI_CASES if c[6] is None])
def test_sparse_train_equals_restated_merger_and_dense_train(hostapi, ft, seed, nf, total, limit, ops, arr, fbs):
    _, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, limit, ops, arr, fbs)
    m = load(hostapi, nf, words, avg, removed, store)
    gterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    for variant, (dboost, dweight) in enumerate(((1.0, 0.5), (1.7, 0.8))):
        cfg = ft.default_config(nf, merge_limit=limit, min_rank=5 if variant == 0 else 60)
        cfg["distance_boost"], cfg["distance_weight"] = dboost, dweight
        for exc in (None, excluded):
            d, s = both_trains(hostapi, m, cfg, gterms, exc)
            assert same(d, s), (variant, exc is not None, len(d[0]), len(s[0]))
            wd, wp, wf, wn, wpre = ft.merge_query(cfg, terms, total, words, avg, removed, exc, sort_by_rank=False, distance_boost=dboost, distance_weight=dweight)
            assert np.array_equal(s[0], wd.astype(np.int32)) and np.array_equal(s[1].view(np.uint32), wp.view(np.uint32)) and s[4] == wpre
            assert np.array_equal(s[2], wf) and np.array_equal(s[3], wn)
    m.close()


@pytest.mark.parametrize("limit,ops,total,sizes", [
    (20000, (1, 1), 200_000, (3000, 12_000)),          # no limit in reach: every document in (row, document) order
    (900, (1, 1, 1), 200_000, (3000, 12_000)),         # preselect: few distinct scores, hundreds of ties at the threshold over 25 ranges
    (2500, (2, 1), 200_000, (20_000, 60_000)),         # AND + OR
    (700, (2, 2), 150_000, (30_000, 90_000)),
    (20000, (1, 3, 2), 200_000, (3000, 12_000)),       # a NOT term
    (150, (1, 3, 1), 100_000, (2000, 9000)),
    (3000, (1, 1, 1, 1), 300_000, (20_000, 50_000)),
])
def test_sparse_train_many_ranges(hostapi, ft, limit, ops, total, sizes):
    nf = 2
    _, words, avg, removed, excluded, terms, store = _multi_case(8100 + limit + len(ops), nf, total, limit, ops, False, None, sizes=sizes)
    m = load(hostapi, nf, words, avg, removed, store)
    gterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    cfg = ft.default_config(nf, merge_limit=limit)
    for exc in (None, excluded):
        d, s = both_trains(hostapi, m, cfg, gterms, exc)
        assert same(d, s), (exc is not None, len(d[0]), len(s[0]))
        w = ft.merge_query(cfg, terms, total, words, avg, removed, exc, sort_by_rank=False)
        assert np.array_equal(s[0], w[0].astype(np.int32)) and np.array_equal(s[1].view(np.uint32), w[1].view(np.uint32)) and s[4] == w[4]
        assert np.array_equal(s[2], w[2]) and np.array_equal(s[3], w[3])
    m.close()


def test_sparse_train_keeps_threshold_ties_in_document_order(hostapi, ft):
    """One proc for every posting: documents of both terms score 2 p, documents of one term p.  mergeLimit between the two counts makes p the
    threshold with ~34 000 ties of which a few thousand are kept — the first ones in document order, i.e. those of the leading ranges: the
    ordered count over the units (ft_sp_select's look-back) decides which."""
    nf, total, limit = 1, 70_000, 26_900
    rng = np.random.default_rng(11)
    words = np.ones((total, nf), np.float32) * 5
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    store, terms, gterms = [], [], []
    for t in range(2):
        doc = np.sort(rng.choice(np.arange(1, total), 30_000, replace=False)).astype(np.uint32)
        po = np.arange(doc.shape[0] + 1, dtype=np.uint32)
        fp = make_fpos(rng.integers(0, 40, doc.shape[0]), np.zeros(doc.shape[0], np.int64)).astype(np.uint64)
        s = dict(word=t, doc=doc, pos_off=po, fpos=fp, proc=100.0)
        store.append(s)
        o = hostapi.default_ft_opts(nf)
        terms.append(dict(op=1, opts=o, subs=[s]))
        gterms.append(dict(op=1, opts=o, subs=[(t, 100.0)]))
    cfg = ft.default_config(nf, merge_limit=limit)
    w = ft.merge_query(cfg, terms, total, words, avg, None, None, sort_by_rank=False)
    m = load(hostapi, nf, words, avg, None, store)
    d, s = both_trains(hostapi, m, cfg, gterms, None)
    assert s[4] and len(s[0]) == len(w[0]) <= limit
    assert same(d, s)
    assert np.array_equal(s[0], w[0].astype(np.int32)) and np.array_equal(s[1].view(np.uint32), w[1].view(np.uint32)) and np.array_equal(s[3], w[3])
    only_one = np.setxor1d(store[0]["doc"], store[1]["doc"])
    kept = np.intersect1d(s[0].astype(np.uint32), only_one)
    assert 0 < len(kept) < len(only_one) and kept.max() < only_one[len(kept) + 5]   # a prefix of the ties, in document order
    m.close()


@pytest.mark.parametrize("nsub_range,sizes", [((3, 6), (2000, 9000)), ((2, 3), (30_000, 60_000))])
def test_sparse_train_simple_query_and_merge_limit_cut(hostapi, ft, nsub_range, sizes):
    """Merger::mergeSimple: max over the sub-terms per document, the first mergeLimit documents in (sub-term row, document) order.  The sparse
    train ranks only the documents whose slot lies below the limit."""
    nf, total = 2, 150_000
    _, words, avg, removed, excluded, terms, store = _multi_case(909, nf, total, 20000, (1,), False, None, sizes=sizes, nsub_range=nsub_range)
    m = load(hostapi, nf, words, avg, removed, store)
    gterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    for limit in (20000, 3000, 300):
        cfg = ft.default_config(nf, merge_limit=limit)
        for exc in (None, excluded):
            d, s = both_trains(hostapi, m, cfg, gterms, exc)
            assert same(d, s), (limit, len(d[0]), len(s[0]))
            assert len(s[0]) <= limit
            w = ft.merge_query(cfg, terms, total, words, avg, removed, exc, sort_by_rank=False)
            assert np.array_equal(s[0], w[0].astype(np.int32)) and np.array_equal(s[1].view(np.uint32), w[1].view(np.uint32))
    m.close()


def test_queries_the_sparse_train_must_not_take_run_dense(hostapi, ft):
    """Unequal field boosts (calcTermScores / calcTermBitmask then look at every occurrence's fields) and more than 16 sub-terms."""
    nf, total = 3, 60_000
    fbs = [[1.0, 0.0, 2.0], [1.0, 1.0, 1.0]]
    _, words, avg, removed, excluded, terms, store = _multi_case(77, nf, total, 500, (1, 1), True, fbs, sizes=(1000, 4000))
    m = load(hostapi, nf, words, avg, removed, store)
    gterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    cfg = ft.default_config(nf, merge_limit=500)
    d, s = both_trains(hostapi, m, cfg, gterms, None, expect_sparse=False)
    assert same(d, s)
    m.close()
    _, words, avg, removed, excluded, terms, store = _multi_case(78, 2, total, 500, (1, 1), False, None, sizes=(300, 900), nsub_range=(9, 12))
    m = load(hostapi, 2, words, avg, removed, store)
    gterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    cfg = ft.default_config(2, merge_limit=500)
    d, s = both_trains(hostapi, m, cfg, gterms, None, expect_sparse=False)
    assert same(d, s)
    m.close()


def test_batch_mixing_both_trains(hostapi, ft):
    """MergeQueryBatch with queries of both kinds in one call: the sparse ones run as one train, the dense ones as another."""
    nf, total = 2, 120_000
    _, words, avg, removed, excluded, terms_all, store = _multi_case(4242, nf, total, 700, (1, 1, 2, 1, 3, 1), False, None, sizes=(1500, 9000), nsub_range=(2, 4))
    wide = _multi_case(4243, nf, total, 700, (1, 1), False, None, sizes=(300, 900), nsub_range=(10, 12))
    for s in wide[6]:
        s["word"] += 1000
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg, removed)
    for s_ in store + wide[6]:
        m.set_word_fpos(s_["word"], s_)
    rng = np.random.default_rng(9)
    queries, oracle_terms = [], []
    for it in range(12):
        if it % 4 == 3:
            terms = wide[5]
        else:
            pick = sorted(rng.choice(len(terms_all), int(rng.integers(1, 4)), replace=False).tolist())
            terms = [terms_all[i] for i in pick]
            if all(t["op"] == 3 for t in terms):
                terms = [terms_all[0]]
        oracle_terms.append(terms)
        queries.append([dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms])
    cfg = ft.default_config(nf, merge_limit=700)
    hostapi.set_ft_train_mode(1)
    m.read_train_stats()
    got = m.merge_query_batch(cfg, queries, sort_by_rank=False)
    dense_n, sparse_n = m.read_train_stats()
    assert dense_n == 3 and sparse_n == 9
    for terms, g in zip(oracle_terms, got):
        w = ft.merge_query(cfg, terms, total, words, avg, removed, None, sort_by_rank=False)
        assert np.array_equal(g[0], w[0].astype(np.int32)) and np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32)) and g[4] == w[4]
        assert np.array_equal(g[3], w[3])
    m.close()


def test_one_merger_many_shapes_auto_mode(hostapi, ft):
    """ONE merger, many query shapes in turn, the host picking the train per query: the tables both trains share (pre-score histogram,
    synchronisation words, look-back words) are handed back clean by whichever ran."""
    nf, total = 2, 100_000
    _, words, avg, removed, excluded, terms_all, store = _multi_case(5151, nf, total, 20000, (1, 1, 2, 1, 3, 1), False, None, sizes=(800, 30_000), nsub_range=(2, 5))
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg, removed)
    for s_ in store:
        m.set_word_fpos(s_["word"], s_)
    rng = np.random.default_rng(5)
    hostapi.set_ft_train_mode(-1)
    m.read_train_stats()
    for it in range(24):
        pick = sorted(rng.choice(len(terms_all), int(rng.integers(1, len(terms_all) + 1)), replace=False).tolist())
        terms = [terms_all[i] for i in pick]
        if all(t["op"] == 3 for t in terms):
            continue
        gterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
        cfg = ft.default_config(nf, merge_limit=int(rng.choice([20000, 1500, 200, 40])))
        exc = excluded if it % 3 == 0 else None
        g = m.merge_query(cfg, gterms, exc, sort_by_rank=False)
        w = ft.merge_query(cfg, terms, total, words, avg, removed, exc, sort_by_rank=False)
        assert np.array_equal(g[0], w[0].astype(np.int32)) and np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32)) and g[4] == w[4], (it, pick)
        assert np.array_equal(g[2], w[2]) and np.array_equal(g[3], w[3])
    dense_n, sparse_n = m.read_train_stats()
    assert dense_n > 0 and sparse_n > 0, (dense_n, sparse_n)
    m.close()
