"""-m gpu: ft_fast merge over DOCUMENT-RANGE shards (SURVEY §8(e) "BM25": "shard by doc-id range — each GPU holds the posting fragments of
its docs; idf uses global N and df ...; exchange = ... the uint16 pre-score histogram for the global threshold").

rxgpu_ft_create_sharded / GpuFtMerger(devices=[...]): the index cut into runs of 8192-document ranges, one per listed device (the 1-GPU box
lists device 0 several times: per-shard launch trains on their own streams, the pre-score histograms + mask popcounts and the table of
first-met documents exchanged in one all-gather each between the kernels — a device copy here, where one device holds every shard; an
ncclAllGather over the listed devices on a multi-GPU node — the merge slots global).

Bar: the merged documents IN MERGE ORDER, raw rank bits, fields, terms counters, uint8 ranks and the preselect flag of the single-device
merger — which tests/test_gpu_ft_terms.py holds to the restated Merger::Merge, itself pinned to the real ft::Merger — and of that restated
merger directly: multi-term queries with AND / OR / NOT, Simple() queries, the preselect phase with ties at the threshold that straddle shard
boundaries, the mergeLimit cut, excluded and removed documents, more shards than document ranges, one merger through many query shapes."""
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


def load(m, words, avg, removed, store):
    m.set_docs(words, avg, removed)   # first: the cut of a sharded index follows the documents
    for s in store:
        m.set_word_fpos(s["word"], s)


def same(a, b):
    return (np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2])
            and np.array_equal(a[3], b[3]) and a[4] == b[4])


def exchange_mode(rxgpu, m):
    return rxgpu.lib().rxgpu_ft_shard_exchange_mode(m.device_index), rxgpu.lib().rxgpu_ft_shard_collectives(m.device_index)


@pytest.mark.parametrize("mode", ["device", "host"])
@pytest.mark.parametrize("shards", [2, 3, 4])
@pytest.mark.parametrize("limit,ops,total,sizes", [
    (20000, (1, 1), 60_000, (3000, 12_000)),          # no limit in reach: the union of the shards' documents in (row, document) order
    (900, (1, 1, 1), 60_000, (3000, 12_000)),         # preselect: the threshold from the summed histograms, ties handed out across the shards
    (2500, (2, 1), 60_000, (4000, 20_000)),           # AND + OR
    (700, (2, 2), 50_000, (9000, 30_000)),
    (20000, (1, 3, 2), 60_000, (3000, 12_000)),       # a NOT term
    (150, (1, 3, 1), 30_000, (2000, 9000)),
])
def test_sharded_merge_equals_single_device_and_restated_merger(rxgpu, hostapi, ft, monkeypatch, mode, shards, limit, ops, total, sizes):
    if mode == "host":
        monkeypatch.setenv("RXGPU_SHARD_MERGE", "host")
    else:
        monkeypatch.delenv("RXGPU_SHARD_MERGE", raising=False)
    nf = 2
    _, words, avg, removed, excluded, terms, store = _multi_case(7000 + limit + shards + len(ops), nf, total, limit, ops, False, None, sizes=sizes)
    one = hostapi.GpuFtMerger(nf)
    many = hostapi.GpuFtMerger(nf, devices=[0] * shards)
    assert rxgpu.lib().rxgpu_ft_shard_count(many.device_index) == shards
    assert exchange_mode(rxgpu, many)[0] == (1 if mode == "device" else 0)
    load(one, words, avg, removed, store)
    load(many, words, avg, removed, store)
    gterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    c0 = exchange_mode(rxgpu, many)[1]
    merges = 0
    for variant, (dboost, dweight) in enumerate(((1.0, 0.5), (1.7, 0.8))):
        cfg = ft.default_config(nf, merge_limit=limit, min_rank=5 if variant == 0 else 60)
        cfg["distance_boost"], cfg["distance_weight"] = dboost, dweight
        for exc in (None, excluded):
            a = one.merge_query(cfg, gterms, exc, sort_by_rank=False)
            b = many.merge_query(cfg, gterms, exc, sort_by_rank=False)
            merges += 1
            assert same(a, b), (variant, exc is not None, len(a[0]), len(b[0]))
            wd, wp, wf, wn, wpre = ft.merge_query(cfg, terms, total, words, avg, removed, exc, sort_by_rank=False, distance_boost=dboost, distance_weight=dweight)
            assert np.array_equal(b[0], wd.astype(np.int32)) and np.array_equal(b[1].view(np.uint32), wp.view(np.uint32)) and b[4] == wpre
            assert np.array_equal(b[2], wf) and np.array_equal(b[3], wn)
    if mode == "device":   # one all-gather for the tables, one more for the histograms when the host half of the 2-phase gate held
        got = exchange_mode(rxgpu, many)[1] - c0
        assert merges <= got <= 2 * merges, (got, merges)
    one.close()
    many.close()


def test_preselect_ties_at_the_threshold_cross_shard_boundaries(hostapi, ft):
    """Every posting has the same pre-score (one sub-term per term, one proc): documents of both terms score 2 p, documents of one term p.
    With mergeLimit between the two counts the threshold is p and only mergeLimit - (documents at 2 p) of its ~34 000 ties are kept, in
    DOCUMENT order — shard 0's ties first, the quota that is left moves on to shard 1, 2, ...  A shard that did not take the ties of the
    shards in front of it into account would keep too many."""
    nf, total, limit = 1, 70_000, 26_900
    rng = np.random.default_rng(11)
    words = np.ones((total, nf), np.float32) * 5
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    from oracle.pyoracle import make_fpos
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
    for shards in (2, 5, 8):
        many = hostapi.GpuFtMerger(nf, devices=[0] * shards)
        load(many, words, avg, None, store)
        b = many.merge_query(cfg, gterms, None, sort_by_rank=False)
        assert b[4] and len(b[0]) == len(w[0]) <= limit
        assert np.array_equal(b[0], w[0].astype(np.int32)) and np.array_equal(b[1].view(np.uint32), w[1].view(np.uint32)) and np.array_equal(b[3], w[3])
        # the kept ties (documents of exactly one term) really straddle a shard boundary, and stop before the last shard
        only_one = np.setxor1d(store[0]["doc"], store[1]["doc"])
        kept = np.intersect1d(b[0].astype(np.uint32), only_one)
        per = -(-9 // shards)   # 70 000 documents = 9 ranges of 8192
        tie_shards = set((kept // 8192 // per).tolist())
        assert (shards - 1) not in tie_shards and (len(tie_shards) > 1 or shards == 2), (shards, sorted(tie_shards), len(kept))   # (two shards: the first holds more ties than the quota)
        many.close()


@pytest.mark.parametrize("shards", [2, 4])
def test_simple_query_and_merge_limit_cut_over_shards(hostapi, ft, shards):
    """Merger::mergeSimple over shards: max over the sub-terms per document, the first mergeLimit documents in (sub-term row, document) order
    — a cut that falls in the middle of the shards' documents."""
    nf, total = 2, 50_000
    _, words, avg, removed, excluded, terms, store = _multi_case(909, nf, total, 20000, (1,), False, None, sizes=(2000, 9000), nsub_range=(3, 6))
    one, many = hostapi.GpuFtMerger(nf), hostapi.GpuFtMerger(nf, devices=[0] * shards)
    load(one, words, avg, removed, store)
    load(many, words, avg, removed, store)
    gterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    for limit in (20000, 3000, 300):
        cfg = ft.default_config(nf, merge_limit=limit)
        for exc in (None, excluded):
            a = one.merge_query(cfg, gterms, exc, sort_by_rank=False)
            b = many.merge_query(cfg, gterms, exc, sort_by_rank=False)
            assert same(a, b), (limit, len(a[0]), len(b[0]))
            assert len(b[0]) <= limit
    one.close()
    many.close()


def test_more_shards_than_document_ranges_and_one_merger_many_shapes(hostapi, ft):
    """20 000 documents are three ranges: of five shards two hold nothing.  ONE sharded merger then runs wide, narrow, preselected and cut
    queries in turn (the kept-clean tables of every shard live across merges)."""
    nf, total = 2, 20_000
    _, words, avg, removed, excluded, terms_all, store = _multi_case(4343, nf, total, 20000, (1, 1, 2, 1, 3, 1), False, None, sizes=(200, 3000), nsub_range=(2, 9))
    one, many = hostapi.GpuFtMerger(nf), hostapi.GpuFtMerger(nf, devices=[0] * 5)
    load(one, words, avg, removed, store)
    load(many, words, avg, removed, store)
    rng = np.random.default_rng(5)
    for it in range(14):
        pick = sorted(rng.choice(len(terms_all), int(rng.integers(2, len(terms_all) + 1)), replace=False).tolist())
        terms = [terms_all[i] for i in pick]
        if all(t["op"] == 3 for t in terms):
            continue
        gterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
        cfg = ft.default_config(nf, merge_limit=int(rng.choice([20000, 1500, 200, 40])))
        exc = excluded if it % 3 == 0 else None
        a = one.merge_query(cfg, gterms, exc, sort_by_rank=False)
        b = many.merge_query(cfg, gterms, exc, sort_by_rank=False)
        assert same(a, b), (it, pick, len(a[0]), len(b[0]))
    one.close()
    many.close()


def test_sharded_index_grows_through_step_commits(rxgpu, hostapi, ft):
    """IndexText::commitFulltextImpl calls SetDocs with a growing totalDocs and uploads only the words that changed (rx_ft_seam.h
    SyncGpuFtMirror).  The cut of a sharded index must survive that: the fragments already on the shards stay where they are, the new
    document ranges go to the last shard, and the merge stays the single index's — across an 8192-document boundary, twice."""
    nf = 2
    full_total = 70_000
    _, words, avg, removed, excluded, terms, store = _multi_case(3131, nf, full_total, 900, (1, 1, 2), False, None, sizes=(4000, 15_000))
    gterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]

    def cut(s, total):   # the word as a commit that has seen `total` documents holds it
        n = int(np.searchsorted(s["doc"], total))
        po = s["pos_off"][:n + 1]
        return dict(word=s["word"], doc=s["doc"][:n], pos_off=po, fpos=s["fpos"][:int(po[-1])], proc=s["proc"])

    many = hostapi.GpuFtMerger(nf, devices=[0, 0, 0])
    lib = rxgpu.lib()
    lib.rxgpu_ft_shard_imbalance.restype = __import__("ctypes").c_double
    sizes_seen = {}
    for step, total in enumerate((7000, 30_000, full_total)):   # 1, 4, 9 document ranges
        w_t, r_t = words[:total], removed[:total]
        a_t = w_t[1:].mean(axis=0).astype(np.float32)
        many.set_docs(w_t, a_t, r_t)
        one = hostapi.GpuFtMerger(nf)
        one.set_docs(w_t, a_t, r_t)
        for s in store:
            c = cut(s, total)
            one.set_word_fpos(c["word"], c)
            if sizes_seen.get(s["word"]) != len(c["doc"]):   # only the words that changed travel again
                many.set_word_fpos(c["word"], c)
                sizes_seen[s["word"]] = len(c["doc"])
        cfg = ft.default_config(nf, merge_limit=900)
        for exc in (None, excluded[:total]):
            a = one.merge_query(cfg, gterms, exc, sort_by_rank=False)
            b = many.merge_query(cfg, gterms, exc, sort_by_rank=False)
            assert same(a, b), (step, total, len(a[0]), len(b[0]))
        oterms = [dict(op=t["op"], opts=t["opts"], subs=[cut(s, total) for s in t["subs"]]) for t in terms]
        w = ft.merge_query(cfg, oterms, total, w_t, a_t, r_t, None, sort_by_rank=False)
        b = many.merge_query(cfg, gterms, None, sort_by_rank=False)
        assert np.array_equal(b[0], w[0].astype(np.int32)) and np.array_equal(b[1].view(np.uint32), w[1].view(np.uint32)) and b[4] == w[4]
        one.close()
        imb = lib.rxgpu_ft_shard_imbalance(many.device_index)
        assert imb == pytest.approx((1.0, 2 / (4 / 3), 7 / 3)[step]), (step, imb)   # cut fixed at 1 range per shard: the last one takes the rest (1, 1, 2 of 4; 1, 1, 7 of 9)
    many.close()


def test_what_a_sharded_ft_index_does_not_offer_says_so(rxgpu, hostapi, ft):
    nf = 1
    m = hostapi.GpuFtMerger(nf, devices=[0, 0])
    with pytest.raises(Exception, match="rxgpu_ft_set_docs first"):
        m.set_word_fpos(0, dict(doc=np.array([1], np.uint32), pos_off=np.array([0, 1], np.uint32), fpos=np.array([3], np.uint64), proc=1.0))
    m.close()


SHARDED_PHRASE_CASES = [
    # (seed, nf, limit, ops, phrases, distances, nsub_range)
    (401, 1, 30000, (1, 1), (0, 0), (1, 8), (1, 4)),                          # the query IS one phrase
    (402, 2, 30000, (1, 1, 1), (0, 0, 0), (1, 12, 12), (2, 5)),               # three terms, several sub-terms each
    (403, 2, 30000, (1, 1, 1), (-1, 0, 0), (1, 1, 10), (1, 4)),               # term OR phrase
    (404, 2, 30000, (2, 2, 1), (0, 0, -1), (1, 15, 1), (2, 4)),               # AND phrase: restricts the term
    (405, 2, 30000, (1, 3, 3), (-1, 0, 0), (1, 1, 15), (2, 4)),               # NOT phrase
    (406, 2, 30000, (1, 1, 1, 1, 1), (-1, 0, 0, -1, -1), (1, 1, 9, 1, 1), (1, 4)),   # terms around a phrase
    (407, 2, 30000, (1, 1, 1, 1), (0, 0, 1, 1), (1, 9, 1, 14), (1, 4)),       # two phrases in a row
]


@pytest.mark.parametrize("shards", [2, 3])
@pytest.mark.parametrize("seed,nf,limit,ops,phrases,distances,nsub", SHARDED_PHRASE_CASES)
def test_phrases_over_document_range_shards(rxgpu, hostapi, ft, shards, seed, nf, limit, ops, phrases, distances, nsub):
    """PhraseResults over a device list (phrasemergerimpl.h:161-329): a phrase is decided inside a document, so every shard runs PhraseMerger
    over its fragments; the rows of the phrase are numbered alike on every shard, the admission cut of the whole index is settled between the
    shards' admission passes and NumDocsMerged() (the 2-phase estimate) is the sum.  The
    sharded merge = the single-device merger's (tests/test_gpu_ft_phrases.py holds that one to the real ft::Merger), merge order included.
    One sub-term of the first phrase term lives in ONE shard only (an empty fragment on the other, active, shards)."""
    total = 40_000
    _, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, limit, ops, False, None, sizes=(2000, 6000), nsub_range=nsub)
    first_phrase_term = next(i for i, ph in enumerate(phrases) if ph >= 0)
    lone = terms[first_phrase_term]["subs"][-1]
    keep = lone["doc"] < 8192   # the lowest-proc sub-term of the phrase's first term: postings in the first document range only
    pos_off = lone["pos_off"]
    sel = np.concatenate([np.arange(pos_off[i], pos_off[i + 1]) for i in np.flatnonzero(keep)]) if keep.any() else np.zeros(0, np.int64)
    lens = (pos_off[1:] - pos_off[:-1])[keep]
    lone["doc"], lone["fpos"] = lone["doc"][keep], lone["fpos"][sel.astype(np.int64)]
    lone["pos_off"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    q = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]], phrase=int(ph), distance=int(d))
         for t, ph, d in zip(terms, phrases, distances)]
    one = hostapi.GpuFtMerger(nf)
    many = hostapi.GpuFtMerger(nf, devices=[0] * shards)
    load(one, words, avg, removed, store)
    load(many, words, avg, removed, store)
    import ctypes
    for m in (one, many):   # the whole list's length on either handle (a shard holds a fragment)
        for sub in (lone, terms[0]["subs"][0]):
            df = ctypes.c_uint64(0)
            assert rxgpu.lib().rxgpu_ft_word_df(m.device_index, sub["word"], ctypes.byref(df)) == 0 and df.value == len(sub["doc"])
    most, differs = 0, False
    for variant, (dboost, dweight) in enumerate(((1.0, 0.5), (1.7, 0.8))):
        # 2500 / 600 / 40: PhraseMerger's own admission cut (at most mergeLimit documents of the first term in (row, document) order,
        # phrasemerger.h:341) falls inside the first term — in a later row, in the first one, inside the first shard's documents
        for lim in (limit, 2500, 600, 40):
            cfg = ft.default_config(nf, merge_limit=lim, min_rank=5 if variant == 0 else 40)
            cfg["distance_boost"], cfg["distance_weight"] = dboost, dweight
            for exc in (None, excluded):
                a = one.merge_query(cfg, q, exc, sort_by_rank=False)
                b = many.merge_query(cfg, q, exc, sort_by_rank=False)
                assert same(a, b), (variant, lim, exc is not None, len(a[0]), len(b[0]))
                most = max(most, len(b[0]))
                if variant == 0 and exc is None and lim == limit:   # the phrase is not the same query as its terms
                    c = many.merge_query(cfg, [dict(t, phrase=-1) for t in q], exc, sort_by_rank=False)
                    differs = differs or len(c[0]) != len(b[0]) or not np.array_equal(c[1].view(np.uint32), b[1].view(np.uint32))
    assert most > 0 and differs
    one.close()
    many.close()


SHARDED_SYN_CASES = [
    # (seed, limit, ops of the query parts, synonyms as lists of term counts, part -> synonym ids)
    (301, 20000, (1,), [2], [[0]]),                          # one OR term with a two-word synonym
    (302, 20000, (2, 1), [2], [[0], []]),                    # AND part: the synonym's mask is OR-ed into the restriction
    (303, 20000, (2, 2), [2, 2, 2], [[0, 1], [2]]),          # two synonyms on one AND part
    (304, 20000, (1, 3, 1), [2], [[0], [], []]),             # a NOT part among them
    (305, 700, (1, 1), [2, 2], [[0], [1]]),                  # mergeLimit + the preselect path with synonym terms in the scores, ties across shards
    (306, 450, (2, 1), [2], [[0], []]),                      # ... with an AND part
    (307, 20000, (1, 1, 1), [3], [[], [0], []]),             # a three-word synonym on the middle part
]


@pytest.mark.parametrize("shards", [2, 3])
@pytest.mark.parametrize("seed,limit,ops,syn_sizes,part_syn", SHARDED_SYN_CASES)
def test_multi_word_synonyms_over_document_range_shards(rxgpu, hostapi, ft, shards, seed, limit, ops, syn_sizes, part_syn):
    """QueryMergeData::synonyms over a device list (mergerimpl.h:347-361, 509-555): a synonym's mask, the term counting, containsFullMultiWordSynonym
    and the removal of the documents that hold only parts of a synonym are decided per DOCUMENT, and a document lies in one shard — the sharded
    merge = the single-device merger's result (which tests/test_gpu_ft_synonyms.py holds to the real ft::Merger), merge order included."""
    nf, total = 2, 50_000
    n_syn_terms = sum(syn_sizes)
    _, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, limit, tuple(ops) + (1,) * n_syn_terms, False, None, sizes=(3000, 12_000),
                                                                 nsub_range=(1, 4))

    def conv(t, op=None):
        return dict(op=t["op"] if op is None else op, opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]], phrase=-1, distance=1)

    parts = [conv(t) for t in terms[:len(ops)]]
    synonyms, at = [], len(ops)
    owner_op = {sid: parts[pi]["op"] for pi, ids in enumerate(part_syn) for sid in ids}
    for sid, k in enumerate(syn_sizes):
        synonyms.append([conv(t, op=owner_op.get(sid, 1)) for t in terms[at:at + k]])
        at += k
    # a duplicate: the first synonym term also finds a word of the first query part (SupressDuplicatesInSynonyms marks that sub-term)
    synonyms[0][0]["subs"] = sorted(synonyms[0][0]["subs"] + [(parts[0]["subs"][0][0], 21.0)], key=lambda x: -x[1])
    one = hostapi.GpuFtMerger(nf)
    many = hostapi.GpuFtMerger(nf, devices=[0] * shards)
    load(one, words, avg, removed, store)
    load(many, words, avg, removed, store)
    most = removed_partial = 0
    for variant, (dboost, dweight) in enumerate(((1.0, 0.5), (1.7, 0.8))):
        cfg = ft.default_config(nf, merge_limit=limit, min_rank=5 if variant == 0 else 40)
        cfg["distance_boost"], cfg["distance_weight"] = dboost, dweight
        for exc in (None, excluded):
            a = one.merge_query(cfg, parts, exc, sort_by_rank=False, synonyms=synonyms, part_synonyms=part_syn)
            b = many.merge_query(cfg, parts, exc, sort_by_rank=False, synonyms=synonyms, part_synonyms=part_syn)
            assert same(a, b), (variant, exc is not None, len(a[0]), len(b[0]))
            plain = many.merge_query(cfg, parts, exc, sort_by_rank=False)
            most = max(most, len(b[0]))
            removed_partial += int(len(plain[0]) != len(b[0]) or not np.array_equal(plain[0], b[0]))
    assert most > 0 and removed_partial > 0   # the synonyms changed the result
    one.close()
    many.close()


def test_query_batch_over_a_device_list_equals_the_single_merges(hostapi, ft):
    """MergeQueryBatch on a merger over a device list: the merges run one after the other (every shard's handle runs one train and its two
    exchanges at a time); each result is the single sharded merge's = the single index's."""
    rng = np.random.default_rng(77)
    total, nf = 40_000, 1
    words = rng.integers(20, 61, (total, nf)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    one, many = hostapi.GpuFtMerger(nf), hostapi.GpuFtMerger(nf, devices=[0, 0, 0])
    for m in (one, many):
        m.set_docs(words, avg)
    for t in range(6):
        doc = np.sort(rng.choice(np.arange(1, total), 3000 + 500 * t, replace=False)).astype(np.uint32)
        s = dict(doc=doc, pos_off=np.arange(doc.shape[0] + 1, dtype=np.uint32), fpos=rng.integers(0, 40, doc.shape[0]).astype(np.uint64), proc=100.0 - 5 * t)
        one.set_word_fpos(t, s)
        many.set_word_fpos(t, s)
    cfg = hostapi.default_ft_config(nf)
    cfg["merge_limit"] = 4000
    opts = hostapi.default_ft_opts(nf)
    queries = [[dict(op=1, opts=opts, subs=[(a, 100.0 - 5 * a)]), dict(op=1 + (a % 2), opts=opts, subs=[(b, 100.0 - 5 * b)])] for a, b in ((0, 1), (2, 3), (4, 5), (1, 4), (3, 0))]
    got = many.merge_query_batch(cfg, queries, sort_by_rank=False)
    for q, g in zip(queries, got):
        w = one.merge_query(cfg, q, None, sort_by_rank=False)
        assert np.array_equal(g[0], w[0]) and np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32)) and g[4] == w[4]
    one.close()
    many.close()
