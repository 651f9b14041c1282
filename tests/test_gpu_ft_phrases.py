"""-m gpu: PHRASES in the ft_fast merge on the GPU (ft_phrase.hip + ft_merge.hip through rxgpu_ft_merge_query_raw and
GpuFtMerger::MergeQuery) against the REAL reference merger — ft::Merger<IdRelVec, MergeData, uint32_t>::Merge with PhraseMerger
(phrasemergerimpl.h:161-329, mergerimpl.h:39-90, 326-384, 504-510) compiled in place (oracle/_ref/libref_ft.so; the shim groups the terms
into PhraseResults the way Selector::Process does, selecterimpl.h:482-572).
Bar: the same documents in the same merge order, the same raw-rank bits, fields and uint8 ranks."""
import numpy as np
import pytest

from oracle.pyoracle import FtOracle, make_fpos, ref_ft_or_none
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


def _engines(hostapi, nf, words, avg, removed, store):
    real = ref_ft_or_none(nf)
    if real is None:
        pytest.skip("oracle/_ref/libref_ft.so not available")
    real.set_docs(words, avg, removed)
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg, removed)
    for s in store:
        real.set_word_fpos(s["word"], s)
        m.set_word_fpos(s["word"], s)
    return real, m


def _query(terms, phrases, distances):
    """terms of _multi_case -> engine terms; phrases[i] = phrase number of term i (-1: plain), distances[i] = its FtDslOpts::distance"""
    return [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]], phrase=int(ph), distance=int(d))
            for t, ph, d in zip(terms, phrases, distances)]


def _compare(real, m, ft, nf, limit, q, excluded, variants=((1.0, 0.5), (1.7, 0.8)), bm25_type="rx", min_results=None):
    most = 0
    for variant, (dboost, dweight) in enumerate(variants):
        cfg = ft.default_config(nf, merge_limit=limit, min_rank=5 if variant != 1 else 40, bm25_type=bm25_type)
        cfg["distance_boost"], cfg["distance_weight"] = dboost, dweight
        real.set_config(cfg, distance_boost=dboost, distance_weight=dweight)
        for exc in (None, excluded):
            wd, wp, wf, wn = real.merge(q, exc, rank_sort_type=1)
            gd, gp, gf, gn, _ = m.merge_query(cfg, q, exc, sort_by_rank=False)
            assert np.array_equal(gd, wd), (variant, len(gd), len(wd), gd[:8], wd[:8])
            assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32)), (variant, gp[:8], wp[:8])
            assert np.array_equal(gn, wn) and np.array_equal(gf, wf)
            most = max(most, len(wd))
            sd, _, sf, sn, _ = m.merge_query(cfg, q, exc, sort_by_rank=True)
            rd, _, rf, rn = real.merge(q, exc, rank_sort_type=0)
            assert np.all(np.diff(sn.astype(int)) <= 0)
            o1, o2 = np.argsort(sd, kind="stable"), np.argsort(rd, kind="stable")
            assert np.array_equal(sd[o1], rd[o2]) and np.array_equal(sn[o1], rn[o2]) and np.array_equal(sf[o1], rf[o2])
    if min_results is not None:
        assert most >= min_results, most   # the case must not pass on empty results
    return most


PHRASE_CASES = [
    # (seed, nf, total, limit, ops, phrases, distances, array_fields, nsub_range)
    (101, 1, 3000, 20000, (1, 1), (0, 0), (1, 8), False, (1, 4)),                    # the query IS one phrase
    (102, 2, 3000, 20000, (1, 1, 1), (0, 0, 0), (1, 12, 12), False, (2, 5)),         # three terms, several sub-terms each
    (103, 2, 3000, 20000, (1, 1, 1), (-1, 0, 0), (1, 1, 10), False, (1, 4)),         # term OR phrase
    (104, 3, 3000, 20000, (1, 1, 1), (0, 0, -1), (1, 10, 1), True, (1, 4)),          # phrase OR term, array positions
    (105, 2, 3000, 20000, (2, 2, 1), (0, 0, -1), (1, 15, 1), False, (2, 4)),         # AND phrase: restricts the term
    (106, 2, 3000, 20000, (1, 3, 3), (-1, 0, 0), (1, 1, 15), False, (2, 4)),         # NOT phrase
    (107, 2, 3000, 20000, (1, 1, 1, 1, 1), (-1, 0, 0, -1, -1), (1, 1, 9, 1, 1), False, (1, 4)),   # terms around a phrase (switchToNextWord bookkeeping)
    (108, 2, 3000, 20000, (1, 1, 1, 1), (0, 0, 1, 1), (1, 9, 1, 14), False, (1, 4)),  # two phrases in a row
    (109, 2, 3000, 20000, (1, 1, 2, 1, 1), (0, 0, -1, 1, 1), (1, 20, 1, 1, 20), False, (1, 3)),
    (110, 2, 3000, 60, (1, 1, 1), (0, 0, -1), (1, 25, 1), False, (2, 5)),            # mergeLimit in the main merge, preselect path with a phrase
    (111, 1, 3000, 40, (1, 1), (0, 0), (1, 30), False, (3, 6)),                      # mergeLimit inside the PhraseMerger (maxMergedDocs_ cut)
]


@pytest.mark.parametrize("seed,nf,total,limit,ops,phrases,distances,arr,nsub", PHRASE_CASES)
def test_gpu_phrase_merge_equals_real_merger(hostapi, ft, seed, nf, total, limit, ops, phrases, distances, arr, nsub):
    _, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, limit, ops, arr, None, sizes=(400, 1500), nsub_range=nsub)
    real, m = _engines(hostapi, nf, words, avg, removed, store)
    try:
        _compare(real, m, ft, nf, limit, _query(terms, phrases, distances), excluded, min_results=1)
    finally:
        real.close()
        m.close()


@pytest.mark.parametrize("bm25_type", ["classic", "word_count"])
def test_gpu_phrase_merge_other_calculators(hostapi, ft, bm25_type):
    nf, total = 2, 3000
    _, words, avg, removed, excluded, terms, store = _multi_case(120, nf, total, 20000, (1, 1, 1), False, None, sizes=(400, 1500), nsub_range=(2, 4))
    real, m = _engines(hostapi, nf, words, avg, removed, store)
    try:
        _compare(real, m, ft, nf, 20000, _query(terms, (0, 0, -1), (1, 12, 1)), excluded, variants=((1.0, 0.5),), bm25_type=bm25_type, min_results=1)
    finally:
        real.close()
        m.close()


def test_gpu_phrase_zero_field_boosts_and_rank_zero_subterms(hostapi, ft):
    """Fields with a zero boost make calcTermRank return 0 for some occurrences: such a posting neither adds the document to the
    PhraseMerger nor takes part in MergeWithDist (phrasemergerimpl.h:197-200) — the first-term admission has to skip it too."""
    nf, total = 3, 3000
    fbs = [[1.0, 0.0, 0.0], [0.0, 1.0, 2.0], [1.0, 1.0, 0.0]]
    _, words, avg, removed, excluded, terms, store = _multi_case(130, nf, total, 20000, (1, 1, 1), True, fbs, sizes=(500, 1500), nsub_range=(2, 5))
    real, m = _engines(hostapi, nf, words, avg, removed, store)
    try:
        for phrases, dist in (((0, 0, 0), (1, 30, 30)), ((0, 0, -1), (1, 30, 1)), ((-1, 0, 0), (1, 1, 30))):
            _compare(real, m, ft, nf, 20000, _query(terms, phrases, dist), excluded, variants=((1.0, 0.5),))
    finally:
        real.close()
        m.close()


def test_gpu_phrase_adjacent_words_corpus(hostapi, ft):
    """A corpus where the phrase really occurs: "w0 w1 w2" at consecutive positions in a third of the documents that hold all three,
    at distance 2-3 in another third.  distance 1 finds the first group only, a wider distance both (with a smaller normDist)."""
    nf, total = 2, 20_000
    rng = np.random.default_rng(7)
    words = rng.integers(3, 30, (total, nf)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    removed = np.zeros(total, np.uint8)
    removed[rng.choice(total, 300, replace=False)] = 1
    excluded = np.zeros(total, np.uint8)
    excluded[rng.choice(total, 300, replace=False)] = 1
    docs = np.sort(rng.choice(np.arange(1, total), 6000, replace=False))
    kind = rng.integers(0, 3, docs.shape[0])          # 0: adjacent, 1: gaps of 2-3, 2: scattered
    per_word = [dict(doc=[], pos=[]) for _ in range(3)]
    for d, k in zip(docs, kind):
        f = int(rng.integers(0, nf))
        base = int(rng.integers(0, 20))
        gaps = (1, 1) if k == 0 else ((int(rng.integers(2, 4)), int(rng.integers(2, 4))) if k == 1 else (int(rng.integers(6, 15)), int(rng.integers(6, 15))))
        p = [base, base + gaps[0], base + gaps[0] + gaps[1]]
        for w in range(3):
            extra = [int(x) for x in rng.integers(30, 60, int(rng.integers(0, 3)))]      # other occurrences far away
            per_word[w]["doc"].append(int(d))
            per_word[w]["pos"].append(sorted({(f, p[w])} | {(int(rng.integers(0, nf)), e) for e in extra}))
    store = []
    for w in range(3):
        pos_off, fp = [0], []
        for plist in per_word[w]["pos"]:
            fp.extend(int(make_fpos([p], [f])[0]) for f, p in plist)
            pos_off.append(len(fp))
        # (sorted by (field, pos) == ascending PosType words)
        s = dict(doc=np.array(per_word[w]["doc"], np.uint32), pos_off=np.array(pos_off, np.uint32), fpos=np.array(fp, np.uint64), proc=100.0 - 10 * w, word=w)
        store.append(s)
    noise = _multi_case(9, nf, total, 20000, (1,), False, None, sizes=(2000, 4000), nsub_range=(2, 3))[5][0]
    for i, s in enumerate(noise["subs"]):
        s["word"] = 10 + i
        store.append(s)
    real, m = _engines(hostapi, nf, words, avg, removed, store)
    opts = dict(boost=1.0, term_len_boost=1.0, field_boost=[1.0] * nf, need_sum_rank=[0] * nf)
    try:
        sizes = []
        for dist in (1, 3, 20):
            q = [dict(op=1, opts=opts, subs=[(w, store[w]["proc"])], phrase=0, distance=(1 if w == 0 else dist)) for w in range(3)]
            q.append(dict(op=1, opts=opts, subs=[(s["word"], s["proc"]) for s in noise["subs"]], phrase=-1, distance=1))
            only_phrase = q[:3]
            sizes.append(_compare(real, m, ft, nf, 20000, only_phrase, excluded, variants=((1.0, 0.5),), min_results=500))
            _compare(real, m, ft, nf, 20000, q, excluded, variants=((1.0, 0.5),), min_results=2000)
            q[3]["op"] = 2                                                     # phrase OR'ed into an AND term
            _compare(real, m, ft, nf, 20000, q, excluded, variants=((1.0, 0.5),), min_results=100)
        assert sizes[0] < sizes[1] < sizes[2], sizes
    finally:
        real.close()
        m.close()


def test_gpu_phrase_long_lists_many_workgroups(hostapi, ft):
    """Posting lists of 30-80 K documents: the admission runs over dozens of ticket-ordered workgroups, the PhraseMerger's own
    maxMergedDocs_ cut falls in the middle of them."""
    nf, total = 2, 200_000
    _, words, avg, removed, excluded, terms, store = _multi_case(140, nf, total, 20000, (1, 1, 1), False, None, sizes=(30_000, 80_000), nsub_range=(2, 4))
    real, m = _engines(hostapi, nf, words, avg, removed, store)
    try:
        _compare(real, m, ft, nf, 20000, _query(terms, (0, 0, -1), (1, 25, 1)), excluded, variants=((1.0, 0.5),), min_results=1000)
        _compare(real, m, ft, nf, 1500, _query(terms, (0, 0, -1), (1, 25, 1)), excluded, variants=((1.0, 0.5),), min_results=100)
        _compare(real, m, ft, nf, 900, _query(terms, (0, 0, 0), (1, 35, 35)), excluded, variants=((1.0, 0.5),), min_results=50)
    finally:
        real.close()
        m.close()


def test_gpu_phrase_empty_and_missing_terms(hostapi, ft):
    """A phrase whose first term matches nothing, a phrase term without sub-terms, a lone NOT phrase (Empty())."""
    nf, total = 2, 3000
    _, words, avg, removed, excluded, terms, store = _multi_case(150, nf, total, 20000, (1, 1, 1), False, None, sizes=(400, 1500), nsub_range=(2, 4))
    real, m = _engines(hostapi, nf, words, avg, removed, store)
    try:
        q = _query(terms, (0, 0, -1), (1, 12, 1))
        q0 = [dict(q[0], subs=[]), q[1], q[2]]                       # the first term of the phrase has no sub-terms
        _compare(real, m, ft, nf, 20000, q0, excluded, variants=((1.0, 0.5),), min_results=1)   # (the plain term still answers)
        q1 = [q[0], dict(q[1], subs=[]), q[2]]                       # the second one has none: no document holds the phrase
        _compare(real, m, ft, nf, 20000, q1, excluded, variants=((1.0, 0.5),), min_results=1)
        q2 = [dict(q[0], op=2), dict(q[1], op=2, subs=[]), q[2]]     # ... as an AND part: nothing at all
        assert _compare(real, m, ft, nf, 20000, q2, excluded, variants=((1.0, 0.5),)) == 0
        lone_not = [dict(q[0], op=3), dict(q[1], op=3)]
        assert _compare(real, m, ft, nf, 20000, lone_not, excluded, variants=((1.0, 0.5),)) == 0
    finally:
        real.close()
        m.close()
