"""-m gpu: Q ft_fast merges in ONE launch train (rxgpu_ft_merge_batch_raw / GpuFtMerger::MergeQueryBatch: the merge kernels run with the
query as the second grid dimension, a single merge being a batch of one).  Bar: query i of a batch returns exactly what MergeQuery returns
for it alone — documents in merge order, raw rank bits, fields, uint8 ranks, the preselect flag — and a sample of them is checked against the
restated reference merger (pinned bit-exact to the real ft::Merger by tests/test_bm25_oracle.py) so that the pair cannot be wrong together.
Shapes in one batch: Simple() queries, AND / OR / NOT terms, preselected and mergeLimit-cut merges, empty queries, more than 16 merged
sub-terms (ft_slot_bases path) next to small ones (own bases), phrases (run one by one inside the call), batches longer than one train."""
import numpy as np
import pytest

from oracle.pyoracle import FtOracle
from .test_bm25_oracle import _multi_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ft(oracle):
    return FtOracle(oracle)


@pytest.fixture(scope="module")
def hostapi(rxgpu):
    from reindexer_amd import hostapi as h
    h.lib()
    return h


def _same(a, b):
    return (np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2])
            and np.array_equal(a[3], b[3]) and a[4] == b[4])


def _queries_from(terms_all, rng, n, max_terms=4):
    """random sub-queries over the case's terms: 1..max_terms of them, ops reshuffled"""
    out = []
    for _ in range(n):
        k = int(rng.integers(1, max_terms + 1))
        pick = rng.choice(len(terms_all), size=min(k, len(terms_all)), replace=False)
        q = []
        for j, ti in enumerate(sorted(pick.tolist())):
            t = terms_all[ti]
            op = int(rng.choice([1, 1, 2, 3])) if (k > 1 and j > 0) else int(rng.choice([1, 2]))
            q.append(dict(op=op, opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]))
        out.append(q)
    return out


@pytest.mark.parametrize("limit,total,sizes,nsub", [(20000, 40_000, (200, 3000), (2, 6)), (300, 40_000, (400, 4000), (2, 5)), (20000, 9_000, (50, 400), (6, 14))])
def test_batch_equals_single_merges_and_the_reference(hostapi, ft, limit, total, sizes, nsub):
    nf = 2
    rng = np.random.default_rng(limit + total)
    _, words, avg, removed, excluded, terms_all, store = _multi_case(777 + limit, nf, total, limit, (1, 1, 2, 1, 3, 1), False, None, sizes=sizes, nsub_range=nsub)
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg, removed)
    for s in store:
        m.set_word_fpos(s["word"], s)
    cfg = ft.default_config(nf, merge_limit=limit, min_rank=5)
    queries = _queries_from(terms_all, rng, 23)
    queries.insert(5, [])                                                         # Empty()
    queries.insert(9, [dict(op=3, opts=terms_all[0]["opts"], subs=queries[0][0]["subs"])])   # a single NOT term: Empty()
    queries.insert(11, [dict(op=1, opts=terms_all[0]["opts"], subs=[])])          # Simple() without sub-terms
    for sort_by_rank in (False, True):
        single = [m.merge_query(cfg, q, None, sort_by_rank=sort_by_rank) if q else (np.zeros(0, np.int32), np.zeros(0, np.float32), np.zeros(0, np.uint8),
                                                                                   np.zeros(0, np.uint8), False) for q in queries]
        batch = m.merge_query_batch(cfg, queries, sort_by_rank=sort_by_rank)
        assert len(batch) == len(queries)
        for i, (a, b) in enumerate(zip(batch, single)):
            assert _same(a, b), (i, len(a[0]), len(b[0]), [t["op"] for t in queries[i]])
    if limit < 1000:
        assert any(b[4] for b in batch), "the small mergeLimit must preselect somewhere"
    # ... and against the restated merger (terms as the oracle wants them: sub-term dicts)
    by_word = {s["word"]: s for s in store}
    for i in (0, 3, 7, 14, 20):
        q = queries[i]
        if not q or (len(q) == 1 and q[0]["op"] == 3) or not any(t["subs"] for t in q):
            continue
        oterms = [dict(op=t["op"], opts=t["opts"], subs=[dict(by_word[w], proc=p) for w, p in t["subs"]]) for t in q]
        wd, wp, wf, wn, wpre = ft.merge_query(cfg, oterms, total, words, avg, removed, None, sort_by_rank=False)
        gd, gp, gf, gn, gpre = m.merge_query_batch(cfg, [q], sort_by_rank=False)[0]
        assert gpre == wpre and np.array_equal(gd, wd.astype(np.int32)) and np.array_equal(gp.view(np.uint32), wp.view(np.uint32))
        assert np.array_equal(gf, wf) and np.array_equal(gn, wn)
    m.close()


def test_batch_longer_than_one_train_and_repeated(hostapi, ft):
    """150 queries = three trains (64 + 64 + 22); the same merger then runs single merges and another batch: the kept-clean tables of every
    batch lane (histograms, bucket counters, sync words, entry-row occupancy) must come back zeroed from the train."""
    nf, total, limit = 2, 30_000, 500
    rng = np.random.default_rng(5)
    _, words, avg, removed, excluded, terms_all, store = _multi_case(4243, nf, total, limit, (1, 1, 2, 1, 1), False, None, sizes=(300, 2500), nsub_range=(2, 5))
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg, removed)
    for s in store:
        m.set_word_fpos(s["word"], s)
    cfg = ft.default_config(nf, merge_limit=limit, min_rank=5)
    queries = _queries_from(terms_all, rng, 150, max_terms=3)
    single = [m.merge_query(cfg, q, None, sort_by_rank=False) for q in queries]
    for rep in range(2):
        batch = m.merge_query_batch(cfg, queries, sort_by_rank=False)
        for i, (a, b) in enumerate(zip(batch, single)):
            assert _same(a, b), (rep, i)
        again = m.merge_query(cfg, queries[rep], None, sort_by_rank=False)
        assert _same(again, single[rep])
    m.close()


def test_batch_with_phrases_inside(hostapi, ft):
    """A phrase query inside a batch runs through its own kernels (one by one inside the call); its neighbours share the train."""
    nf, total = 2, 3000
    _, words, avg, removed, excluded, terms, store = _multi_case(103, nf, total, 20000, (1, 1, 1), False, None, sizes=(400, 1500), nsub_range=(1, 4))
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg, removed)
    for s in store:
        m.set_word_fpos(s["word"], s)
    cfg = ft.default_config(nf, merge_limit=20000, min_rank=5)
    plain = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    phrase = [dict(plain[0]), dict(plain[1], phrase=0, distance=1), dict(plain[2], phrase=0, distance=10)]
    queries = [phrase, plain[:1], plain[:2], phrase, plain[1:]]
    single = [m.merge_query(cfg, q, None, sort_by_rank=False) for q in queries]
    assert len(single[0][0]) > 0
    batch = m.merge_query_batch(cfg, queries, sort_by_rank=False)
    for i, (a, b) in enumerate(zip(batch, single)):
        assert _same(a, b), i
    m.close()
