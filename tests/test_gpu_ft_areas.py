"""-m gpu: MergeDataAreas<Area> on the device — the merge behind highlight() / snippet() (merger.h:36-57 with kWithRegularAreas; addAreas :196-204;
AreasInDocument / AreasInField::Insert / Area::Concat, core/ft/areaholder.h) — through the seam the patched Selector::mergeResults calls
(rx_ft_seam.h: TryMergeOnGpu<IdCont, MergeDataAreas<Area>> -> GpuFtMerger::MergeQueryAreas -> rxgpu_ft_merge_query_areas_raw -> the area
accumulation inside ft_finish's replay) against the REAL ft::Merger<IdCont, MergeDataAreas<Area>, uint32_t> compiled in place
(oracle/_ref/libref_ft_seam.so), over the reference's own containers (packed and plain).
Bar: the same documents, ranks and fields, and per document and field the same areas — as AreasInField::data_ holds them when the merge ends
(insertion order, circular overwrite once maxAreasInDoc is reached and only for a term rank above the document's best so far, joins with the
area inserted last) AND after Commit().  Phrases / multi-word synonyms are declined (the CPU merger builds their areas)."""
import numpy as np
import pytest

from oracle.pyoracle import FtOracle, ref_ft_seam_or_none
from .test_bm25_oracle import _multi_case, make_pos_postings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ft(oracle):
    return FtOracle(oracle)


def _seam(nf, words, avg, removed, store):
    s = ref_ft_seam_or_none(nf)
    if s is None:
        pytest.skip("oracle/_ref/libref_ft_seam.so not available")
    s.set_docs(words, avg, removed)
    for w in store:
        s.set_word_fpos(w["word"], w)
    return s


def _same(got, want, tag):
    assert got is not None, ("the GPU branch declined", tag)
    gd, gp, gf, gn, graw, gcom = got
    wd, wp, wf, wn, wraw, wcom = want
    assert np.array_equal(gd, wd), (tag, len(gd), len(wd))
    assert np.array_equal(gn, wn) and np.array_equal(gf, wf) and np.array_equal(gp.view(np.uint32), wp.view(np.uint32)), tag
    n_areas = 0
    for i in range(len(wd)):
        for f in range(len(wraw[i])):
            assert np.array_equal(graw[i][f], wraw[i][f]), (tag, "raw", i, f, graw[i][f], wraw[i][f])
            assert np.array_equal(gcom[i][f], wcom[i][f]), (tag, "committed", i, f)
            n_areas += len(wraw[i][f])
    return n_areas


@pytest.mark.parametrize("seed,nf,total,limit,ops,arr,max_areas,nsub", [
    (11, 2, 4000, 20000, (1,), False, 5, (2, 4)),            # Simple(): mergeSimple with areas
    (12, 2, 4000, 20000, (1, 1, 1), False, 5, (2, 4)),       # mergeTerm, OR terms: documents gather areas from several terms
    (13, 3, 4000, 20000, (2, 1, 1), True, 3, (1, 3)),        # AND restriction, array positions (areas of different array items never join)
    (14, 2, 4000, 150, (1, 1), False, 2, (2, 5)),            # mergeLimit cut + preselect, tiny maxAreasInDoc: the circular overwrite
    (15, 1, 3000, 20000, (1, 3, 1), False, 1, (2, 4)),       # a NOT term in between; ONE area per document
    (16, 2, 300, 20000, (1, 1, 1, 1), False, 4, (6, 9)),     # many sub-terms over few documents: documents with more postings than the sparse replay orders
])
def test_areas_equal_the_real_merger(rxgpu, ft, seed, nf, total, limit, ops, arr, max_areas, nsub):
    _, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, limit, ops, arr, None, sizes=(300, 1500) if total > 1000 else (40, 150), nsub_range=nsub)
    seam = _seam(nf, words, avg, removed, store)
    cfg = ft.default_config(nf, merge_limit=limit, min_rank=5)
    seam.set_config(cfg)
    assert seam.commit(0) == len(store)
    q = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    total_areas = 0
    for exc in (None, excluded):
        for packed in (True, False):
            for rst in (1, 0):   # RankAndID (merge order) / RankOnly (sorted by rank: areaIndex travels with the entry)
                want = seam.merge_areas(q, max_areas, exc, rank_sort_type=rst, packed=packed, gpu=False)
                got = seam.merge_areas(q, max_areas, exc, rank_sort_type=rst, packed=packed, gpu=True)
                if rst == 0 and got is not None:   # equal ranks may be ordered differently by an unstable sort: compare as (document -> everything) maps
                    o1, o2 = np.argsort(got[0], kind="stable"), np.argsort(want[0], kind="stable")
                    got = (got[0][o1], got[1][o1], got[2][o1], got[3][o1], [got[4][i] for i in o1], [got[5][i] for i in o1])
                    want = (want[0][o2], want[1][o2], want[2][o2], want[3][o2], [want[4][i] for i in o2], [want[5][i] for i in o2])
                total_areas += _same(got, want, (seed, packed, rst, exc is not None))
    assert total_areas > 0
    seam.close()


def test_adjacent_words_join_and_the_limit_overwrites(rxgpu, ft):
    """A corpus where words stand next to each other: w0 at position p, w1 at p + 1 (one area [p, p + 2) after the join), and documents with more
    word occurrences than maxAreasInDoc so that later words overwrite the oldest areas — only while their term rank beats the document's best."""
    nf, total = 1, 2000
    rng = np.random.default_rng(3)
    words = rng.integers(20, 60, (total, nf)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    docs = np.sort(rng.choice(np.arange(1, total), 600, replace=False)).astype(np.uint32)
    store = []
    for w in range(3):
        pos_off, fpos = [0], []
        for d in docs:
            base = int(rng.integers(0, 10))
            ps = sorted({base + w + 7 * k for k in range(int(rng.integers(1, 5)))})   # w1 right behind w0, w2 behind w1; up to 4 occurrences each
            fpos += [p for p in ps]
            pos_off.append(len(fpos))
        store.append(dict(word=w, doc=docs.copy(), pos_off=np.array(pos_off, np.uint32), fpos=np.array(fpos, np.uint64), proc=100.0 - 10 * w))
    seam = _seam(nf, words, avg, None, store)
    cfg = ft.default_config(nf, merge_limit=20000, min_rank=5)
    seam.set_config(cfg)
    assert seam.commit(0) == 3
    opts = dict(boost=1.0, term_len_boost=1.0, field_boost=[1.0], need_sum_rank=[0])
    q = [dict(op=1, opts=opts, subs=[(w, store[w]["proc"])]) for w in range(3)]
    for max_areas in (1, 2, 5, 40):
        want = seam.merge_areas(q, max_areas, None, packed=True, gpu=False)
        got = seam.merge_areas(q, max_areas, None, packed=True, gpu=True)
        n = _same(got, want, max_areas)
        assert n >= len(want[0])
    seam.close()


def test_phrases_and_unlimited_areas_are_declined(rxgpu, ft):
    nf, total = 2, 3000
    _, words, avg, removed, excluded, terms, store = _multi_case(21, nf, total, 20000, (1, 1, 1), False, None, sizes=(300, 900), nsub_range=(1, 3))
    seam = _seam(nf, words, avg, removed, store)
    seam.set_config(ft.default_config(nf, merge_limit=20000, min_rank=5))
    seam.commit(0)
    plain = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    phrase = [dict(plain[0], phrase=0, distance=1), dict(plain[1], phrase=0, distance=5), plain[2]]
    assert seam.merge_areas(phrase, 5, None, gpu=True) is None            # the CPU merger builds a phrase's areas
    assert seam.merge_areas(plain, -1, None, gpu=True) is None            # maxAreasInDoc <= 0 (unlimited): CPU merger
    assert seam.merge_areas(plain, 5, None, gpu=True) is not None
    seam.close()


@pytest.mark.parametrize("seed,nf,total,limit,ops,arr,max_areas,nsub", [
    (21, 2, 30_000, 20000, (1,), False, 5, (2, 4)),           # Simple() with areas over three shards
    (22, 2, 30_000, 20000, (1, 1, 1), False, 5, (2, 4)),      # OR terms: a document gathers areas from several terms, on the shard it lies in
    (23, 3, 30_000, 20000, (2, 1, 1), True, 3, (1, 3)),       # AND restriction, array positions
    (24, 2, 30_000, 900, (1, 1), False, 2, (2, 5)),           # mergeLimit cut + preselect across the shards, the circular overwrite
])
def test_areas_over_a_device_list_equal_the_real_merger(rxgpu, ft, seed, nf, total, limit, ops, arr, max_areas, nsub):
    """MergeDataAreas<Area> with the mirror over a DEVICE LIST (document-range shards): a document's areas are built by the shard
    that holds it, at the document's global merge slot — the slot-wise union of the shards' area arrays is the single merger's, i.e. the real
    ft::Merger<IdCont, MergeDataAreas<Area>>'s, raw and committed."""
    _, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, limit, ops, arr, None, sizes=(2500, 11_000), nsub_range=nsub)
    seam = _seam(nf, words, avg, removed, store)
    seam.set_config(ft.default_config(nf, merge_limit=limit, min_rank=5))
    assert seam.commit(devices=[0, 0, 0]) == len(store)
    q = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    total_areas = 0
    for exc in (None, excluded):
        for packed in (True, False):
            want = seam.merge_areas(q, max_areas, exc, rank_sort_type=1, packed=packed, gpu=False, cap=1 << 15, area_cap=1 << 22)
            got = seam.merge_areas(q, max_areas, exc, rank_sort_type=1, packed=packed, gpu=True, cap=1 << 15, area_cap=1 << 22)
            total_areas += _same(got, want, (seed, packed, exc is not None))
    assert total_areas > 0
    seam.close()
