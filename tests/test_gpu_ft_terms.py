"""-m gpu: ft_fast MULTI-TERM merge on the GPU (ft_merge.hip through rxgpu_ft_merge_terms_raw and GpuFtMerger::MergeQuery) vs the
CPU restatement of Merger::Merge, which tests/test_bm25_oracle.py pins bit-exact against the real reference merger.
Bar: the same documents in the same merge order, the same raw-rank bits, fields and uint8 ranks — for AND / OR / NOT terms, zero field
boosts, array positions, the mergeLimit cut inside mergeTerm and the preselect phase."""
import numpy as np
import pytest

from oracle.pyoracle import FtOracle
from .test_bm25_oracle import MULTI_CASES, _multi_case, make_pos_postings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ft(oracle):
    return FtOracle(oracle)


@pytest.fixture(scope="module")
def hostapi(rxgpu):
    from reindexer_amd import hostapi as h
    h.lib()
    return h


def _check(hostapi, ft, nf, total, limit, words, avg, removed, excluded, terms, store, expect_pre=None, variants=((1.0, 0.5), (1.7, 0.8), (0.0, 1.0)),
           bm25_type="rx"):
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg, removed)
    for s in store:
        m.set_word_fpos(s["word"], s)
    gterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    saw_pre = False
    for variant, (dboost, dweight) in enumerate(variants):
        cfg = ft.default_config(nf, merge_limit=limit, min_rank=5 if variant != 1 else 60, bm25_type=bm25_type)
        cfg["distance_boost"], cfg["distance_weight"] = dboost, dweight
        for exc in (None, excluded):
            wd, wp, wf, wn, wpre = ft.merge_query(cfg, terms, total, words, avg, removed, exc, sort_by_rank=False, distance_boost=dboost,
                                                  distance_weight=dweight)
            gd, gp, gf, gn, gpre = m.merge_query(cfg, gterms, exc, sort_by_rank=False)
            saw_pre |= gpre
            assert gpre == wpre
            assert np.array_equal(gd, wd.astype(np.int32)), (variant, len(gd), len(wd))
            assert np.array_equal(gn, wn) and np.array_equal(gf, wf)
            assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32))
            # rank-sorted flavour: same (doc -> rank) map, non-increasing ranks
            sd, _, sf, sn, _ = m.merge_query(cfg, gterms, exc, sort_by_rank=True)
            assert np.all(np.diff(sn.astype(int)) <= 0)
            o1, o2 = np.argsort(sd, kind="stable"), np.argsort(gd, kind="stable")
            assert np.array_equal(sd[o1], gd[o2]) and np.array_equal(sn[o1], gn[o2]) and np.array_equal(sf[o1], gf[o2])
    if expect_pre is not None:
        assert saw_pre == expect_pre
    m.close()


@pytest.mark.parametrize("seed,nf,total,limit,ops,arr,fbs", MULTI_CASES)
def test_gpu_multi_term_merge_equals_restated_merger(hostapi, ft, seed, nf, total, limit, ops, arr, fbs):
    _, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, limit, ops, arr, fbs)
    _check(hostapi, ft, nf, total, limit, words, avg, removed, excluded, terms, store, expect_pre=(limit < 1000 and 1 in ops))


@pytest.mark.parametrize("bm25_type", ["classic", "word_count"])
@pytest.mark.parametrize("case", [0, 3, len(MULTI_CASES) - 1])
def test_gpu_multi_term_merge_classic_and_word_count(hostapi, ft, bm25_type, case):
    """The other two calculators of Bm25Calculator<BM> (bm25.h:38-68) through the multi-term merge."""
    seed, nf, total, limit, ops, arr, fbs = MULTI_CASES[case]
    _, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, limit, ops, arr, fbs)
    _check(hostapi, ft, nf, total, limit, words, avg, removed, excluded, terms, store, variants=((1.0, 0.5),), bm25_type=bm25_type)


@pytest.mark.parametrize("limit,ops", [(20000, (1, 1)), (3000, (1, 1, 1)), (2500, (2, 1)), (700, (2, 2)), (20000, (1, 3, 2))])
def test_gpu_multi_term_many_workgroups(hostapi, ft, limit, ops):
    """Posting lists of 20-90 K documents: dozens of ticket-ordered workgroups per launch, slots assigned across them."""
    nf, total = 2, 200_000
    _, words, avg, removed, excluded, terms, store = _multi_case(1000 + limit, nf, total, limit, ops, False, None, sizes=(20_000, 90_000))
    _check(hostapi, ft, nf, total, limit, words, avg, removed, excluded, terms, store, variants=((1.0, 0.5),))


@pytest.mark.parametrize("limit,ops,nsub", [(20000, (1, 1, 2, 1), (30, 50)), (400, (1, 1, 1), (40, 60)), (20000, (2, 1), (9, 12)), (150, (1, 3, 1), (20, 30))])
def test_gpu_multi_term_wide_queries(hostapi, ft, limit, ops, nsub):
    """Dozens of sub-terms per term (typo / stem variants): more merged sub-terms than ft_finish sums slot bases for on its own
    (ft_slot_bases path), more than its staged replay descriptors, several document ranges."""
    nf, total = 2, 30_000
    _, words, avg, removed, excluded, terms, store = _multi_case(2000 + limit + len(ops), nf, total, limit, ops, False, None, sizes=(150, 900),
                                                                 nsub_range=nsub)
    assert sum(len(t["subs"]) for t in terms if t["op"] != 3) > 16
    _check(hostapi, ft, nf, total, limit, words, avg, removed, excluded, terms, store, variants=((1.0, 0.5),))


@pytest.mark.parametrize("limit,ops,nsub,total,sizes", [
    (20000, (1, 1, 1), (1, 4), 20_000, (40, 120)),      # a few hundred records per document range, 1-3 postings per document
    (20000, (2, 1, 1), (2, 4), 20_000, (60, 200)),
    (60, (1, 1, 1), (2, 4), 20_000, (100, 250)),        # ... and the mergeLimit cut inside the sparse ranges
    (20000, (1, 1, 1), (6, 8), 600, (15, 40)),          # up to ~6 postings per document, still ordered in registers
    (20000, (1, 1, 2, 1), (9, 11), 200, (10, 20)),      # ~40 sub-terms over 200 documents: documents with more postings than the
    (20000, (1, 1, 1), (12, 14), 150, (8, 16)),         # sparse replay orders in registers => the range falls back to the entry rows
    (30, (1, 1, 1), (12, 14), 150, (8, 16)),
])
def test_gpu_multi_term_sparse_ranges(hostapi, ft, limit, ops, nsub, total, sizes):
    """Document ranges with no more than a few hundred surviving records take ft_finish's sparse replay (records chained per document in
    LDS, one thread per merged document); a document with more postings than its sorting network holds sends the range down the general
    path.  Both must give the merger's result."""
    nf = 2
    _, words, avg, removed, excluded, terms, store = _multi_case(3000 + limit + total + len(ops), nf, total, limit, ops, False, None, sizes=sizes,
                                                                 nsub_range=nsub)
    _check(hostapi, ft, nf, total, limit, words, avg, removed, excluded, terms, store, variants=((1.0, 0.5), (1.7, 0.8)))


def test_gpu_one_merger_many_query_shapes(hostapi, ft):
    """The tables a merge finds zeroed and hands back zeroed (histogram copies, bucket counters, entry-row occupancy, sync words) live
    across merges: ONE merger instance runs wide, narrow, simple, AND-only, preselected and cut queries in turn — different numbers of
    sub-term rows and different mergeLimits reshape the entry rows between merges — and every result must still be the oracle's."""
    nf, total = 2, 40_000
    rng = np.random.default_rng(77)
    _, words, avg, removed, excluded, terms_all, store = _multi_case(4242, nf, total, 20000, (1, 1, 2, 1, 3, 1), False, None, sizes=(200, 3000),
                                                                     nsub_range=(2, 14))
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg, removed)
    for s_ in store:
        m.set_word_fpos(s_["word"], s_)

    def gq(ts):
        return [dict(op=t["op"], opts=t["opts"], subs=[(x["word"], x["proc"]) for x in t["subs"]]) for t in ts]

    shapes = [([0, 1, 2, 3, 4, 5], 20000), ([0], 20000), ([2, 3], 300), ([0, 1], 150), ([5], 50), ([0, 1, 2, 3, 4, 5], 700), ([1, 3, 5], 20000),
              ([2], 20000), ([0, 2, 4], 97), ([0, 1, 2, 3, 4, 5], 20000)]
    for rnd in range(2):
        for pick, limit in shapes:
            ts = [terms_all[i] for i in pick]
            if all(t["op"] == 3 for t in ts):
                continue
            cfg = ft.default_config(nf, merge_limit=limit, min_rank=5)
            exc = excluded if (rnd + len(pick)) % 2 else None
            wd, wp, wf, wn, wpre = ft.merge_query(cfg, ts, total, words, avg, removed, exc, sort_by_rank=False)
            gd, gp, gf, gn, gpre = m.merge_query(cfg, gq(ts), exc, sort_by_rank=False)
            assert gpre == wpre, (pick, limit)
            assert np.array_equal(gd, wd.astype(np.int32)), (pick, limit, len(gd), len(wd))
            assert np.array_equal(gn, wn) and np.array_equal(gf, wf)
            assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32))
    m.close()


def test_gpu_multi_term_edge_cases(hostapi, ft):
    nf, total = 2, 500
    rng = np.random.default_rng(5)
    words = rng.integers(1, 4, (total, nf)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg)
    a = make_pos_postings(rng, total, nf, 60, 100.0)
    b = make_pos_postings(rng, total, nf, 80, 100.0)
    m.set_word_fpos(0, a)
    m.set_word_fpos(1, b)
    cfg = hostapi.default_ft_config(nf)
    o = hostapi.default_ft_opts(nf)
    # Empty(): no terms, or a lone NOT
    assert len(m.merge_query(cfg, [])[0]) == 0
    assert len(m.merge_query(cfg, [dict(op=3, opts=o, subs=[(0, 100.0)])])[0]) == 0
    # one OR term is Simple(): same result as merge()
    q1 = m.merge_query(cfg, [dict(op=1, opts=o, subs=[(0, 100.0)])], sort_by_rank=False)
    q2 = m.merge(cfg, o, [(0, 100.0)], sort_by_rank=False)
    assert all(np.array_equal(x, y) for x, y in zip(q1[:4], q2))
    # a term without sub-terms: OR contributes nothing, AND empties the result
    oa = ft.merge_query(cfg, [dict(op=1, opts=o, subs=[]), dict(op=1, opts=o, subs=[b])], total, words, avg, None, None, sort_by_rank=False)
    ga = m.merge_query(cfg, [dict(op=1, opts=o, subs=[]), dict(op=1, opts=o, subs=[(1, 100.0)])], sort_by_rank=False)
    assert np.array_equal(ga[0], oa[0].astype(np.int32)) and np.array_equal(ga[3], oa[3])
    assert len(m.merge_query(cfg, [dict(op=2, opts=o, subs=[]), dict(op=1, opts=o, subs=[(1, 100.0)])])[0]) == 0
    # a word uploaded without positions cannot take part in a multi-term merge: loud error, no fallback
    from .test_bm25_oracle import make_postings
    m.set_word_flat(2, make_postings(rng, total, nf, 30))
    with pytest.raises(Exception):
        m.merge_query(cfg, [dict(op=1, opts=o, subs=[(2, 100.0)]), dict(op=1, opts=o, subs=[(1, 100.0)])])
    m.close()


def test_gpu_concurrent_callers_share_one_index(hostapi, ft):
    """Several planner threads query ONE text index at a time: every caller gets its own lane of the handle (stream, scratch, staging,
    kept-clean tables) behind the shared dictionary.  Different query shapes in flight together, each result the oracle's; then the same
    query from 6 native threads x 8 (checked inside the driver against a merge made alone)."""
    import threading
    nf, total = 2, 60_000
    _, words, avg, removed, excluded, terms_all, store = _multi_case(5151, nf, total, 20000, (1, 1, 2, 1, 3, 1), False, None, sizes=(500, 6000),
                                                                     nsub_range=(2, 6))
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg, removed)
    for s_ in store:
        m.set_word_fpos(s_["word"], s_)

    def gq(ts):
        return [dict(op=t["op"], opts=t["opts"], subs=[(x["word"], x["proc"]) for x in t["subs"]]) for t in ts]

    shapes = [([0, 1, 2, 3, 4, 5], 20000), ([0], 20000), ([2, 3], 300), ([0, 1], 150), ([1, 3, 5], 20000), ([0, 2, 4], 97)]
    want = []
    for pick, limit in shapes:
        cfg = ft.default_config(nf, merge_limit=limit, min_rank=5)
        ts = [terms_all[i] for i in pick]
        want.append((cfg, ts, ft.merge_query(cfg, ts, total, words, avg, removed, excluded, sort_by_rank=False)))
    errors = []

    def worker(k):
        try:
            for rnd in range(6):
                cfg, ts, (wd, wp, wf, wn, wpre) = want[(k + rnd) % len(want)]
                gd, gp, gf, gn, gpre = m.merge_query(cfg, gq(ts), excluded, sort_by_rank=False)
                assert gpre == wpre and np.array_equal(gd, wd.astype(np.int32)) and np.array_equal(gn, wn) and np.array_equal(gf, wf)
                assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32))
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    cfg, ts, (wd, *_rest) = want[0]
    n, wall_ms = m.merge_query_concurrent(cfg, gq(ts), threads=6, repeats=8, excluded=excluded)
    assert n == len(wd) and wall_ms > 0
    # a dictionary update while nobody merges, then merges again (the lanes keep their scratch, the statistics are re-read)
    m.set_docs(words, avg, removed)
    gd, *_ = m.merge_query(cfg, gq(ts), excluded, sort_by_rank=False)
    assert np.array_equal(gd, wd.astype(np.int32))
    m.close()
