"""The FT half of the drop-in boundary, EXECUTED over the reference's own types.

oracle/_ref/libref_ft_seam.so (built here from the reference tree by oracle/Makefile, travels to the GPU box) is a TU compiled INSIDE a
scratch copy of the reference's ft headers with integration/patches/0003-ft-fast-gpu-merger.patch applied: it holds a word table of
PackedWordEntry<PackedIdRelVec> / <IdRelVec> as DataHolder<IdCont>::words_ does, builds ft::QueryMergeData<IdCont> / FTConfig / FtDslOpts /
FtMergeStatuses::Statuses as Selector<IdCont> does, and merges it
  * with the reference's ft::Merger<IdCont, ft::MergeData, uint32_t>::Merge<Bm25T>   (what mergeResults runs today), and
  * with rxgpu::host::TryMergeOnGpu (reindexer_amd/host/rx_ft_seam.h)                 (what the patched mergeResults runs first):
    ToGpuCfg / ToGpuOpts / ToGpuTerms, the commit-time hand-over GpuFtMirror::SyncDocs / SyncWords (packed streams decoded on the device),
    GpuFtMerger::MergeQuery, ToRxMergeData.
Bar: the two ft::MergeData are identical — documents in merge order, rank bits, fields, uint8 ranks."""
import numpy as np
import pytest

from oracle.pyoracle import FtOracle, ref_ft_seam_or_none
from .test_bm25_oracle import MULTI_CASES, _multi_case, make_pos_postings


@pytest.fixture(scope="module")
def ft(oracle):
    return FtOracle(oracle)


def _seam(nf, words, avg, removed, store):
    seam = ref_ft_seam_or_none(nf)
    if seam is None:
        pytest.skip("oracle/_ref/libref_ft_seam.so not available (built where /root/reference exists)")
    seam.set_docs(words, avg, removed)
    for s in store:
        seam.set_word_fpos(s["word"], s)
    return seam


def _same(a, b, tag):
    assert a is not None and b is not None, tag
    assert np.array_equal(a[0], b[0]), (tag, len(a[0]), len(b[0]))
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), tag
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]), tag


@pytest.mark.parametrize("case", [0, 4, 5, 8])
def test_reference_merger_over_packed_lists_equals_plain_lists(ft, case):
    """CPU: the shim's two word tables hold the same postings — the reference's merger must not care which container it walks (and the
    library, which links the product's host layer, loads without a GPU)."""
    seed, nf, total, limit, ops, arr, fbs = MULTI_CASES[case]
    _, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, limit, ops, arr, fbs)
    seam = _seam(nf, words, avg, removed, store)
    seam.set_config(ft.default_config(nf, merge_limit=limit))
    rterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    for exc in (None, excluded):
        _same(seam.merge(rterms, exc, packed=True), seam.merge(rterms, exc, packed=False), case)
    seam.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,nf,total,limit,ops,arr,fbs", MULTI_CASES)
def test_patched_merge_results_branch_equals_reference_merger(rxgpu, ft, seed, nf, total, limit, ops, arr, fbs):
    _, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, limit, ops, arr, fbs)
    seam = _seam(nf, words, avg, removed, store)
    assert seam.commit(0) == len(store)
    rterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    for variant, (dboost, dweight) in enumerate([(1.0, 0.5), (1.7, 0.8)]):
        seam.set_config(ft.default_config(nf, merge_limit=limit, min_rank=5 if variant == 0 else 60), distance_boost=dboost, distance_weight=dweight)
        for exc in (None, excluded):
            for packed in (True, False):
                want = seam.merge(rterms, exc, rank_sort_type=1, packed=packed, gpu=False)
                got = seam.merge(rterms, exc, rank_sort_type=1, packed=packed, gpu=True)
                _same(got, want, (variant, packed))
                # RankOnly: sorted by uint8 rank, ties unspecified in the reference (pdqsort) — same (doc -> rank, field) map, non-increasing
                ws = seam.merge(rterms, exc, rank_sort_type=0, packed=packed, gpu=False)
                gs = seam.merge(rterms, exc, rank_sort_type=0, packed=packed, gpu=True)
                assert np.all(np.diff(gs[3].astype(int)) <= 0)
                o1, o2 = np.argsort(ws[0], kind="stable"), np.argsort(gs[0], kind="stable")
                assert np.array_equal(ws[0][o1], gs[0][o2]) and np.array_equal(ws[3][o1], gs[3][o2]) and np.array_equal(ws[2][o1], gs[2][o2])
    seam.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bm25_type", ["rx", "classic", "word_count"])
def test_seam_simple_queries_recommit_and_calculators(rxgpu, ft, bm25_type):
    """Simple() queries (one OR term, several sub-terms) for all three calculators; then a second commit that appends documents to some
    words and adds new words: only the changed lists travel again (fingerprints) and the merges follow."""
    nf, total = 2, 4000
    rng = np.random.default_rng(5)
    words = rng.integers(1, 6, (total, nf)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    store = []
    for w in range(6):
        s = make_pos_postings(rng, total // 2, nf, int(rng.integers(100, 600)), 100.0 - 7 * w, w % 2 == 1)   # documents of the first half
        s["word"] = w
        store.append(s)
    seam = _seam(nf, words, avg, None, store)
    cfg = ft.default_config(nf, merge_limit=300, bm25_type=bm25_type)
    seam.set_config(cfg)
    assert seam.commit(0) == 6
    opts = dict(boost=1.2, term_len_boost=0.9, field_boost=[1.0, 0.5], need_sum_rank=[0, 1])
    simple = [dict(op=1, opts=opts, subs=[(w, store[w]["proc"]) for w in (0, 2, 3)])]
    two = [dict(op=1, opts=opts, subs=[(0, 100.0), (1, 93.0)]), dict(op=2, opts=opts, subs=[(4, 72.0), (5, 65.0)])]
    for q in (simple, two):
        for packed in (True, False):
            _same(seam.merge(q, packed=packed, gpu=True), seam.merge(q, packed=packed, gpu=False), (bm25_type, packed))
    # second commit: words 0 and 4 get documents of the second half appended, word 6 is new
    for w in (0, 4, 6):
        extra = make_pos_postings(rng, total // 2, nf, 300, 100.0 - 7 * w, False)
        extra["doc"] = (np.asarray(extra["doc"], np.uint32) + np.uint32(total // 2 - 1)).astype(np.uint32)
        if w < 6:
            old = store[w]
            merged = dict(doc=np.concatenate([old["doc"], extra["doc"]]),
                          pos_off=np.concatenate([np.asarray(old["pos_off"], np.uint32), np.asarray(extra["pos_off"][1:], np.uint32) + np.uint32(old["pos_off"][-1])]),
                          fpos=np.concatenate([old["fpos"], extra["fpos"]]), proc=old["proc"], word=w)
            assert np.all(np.diff(merged["doc"].astype(np.int64)) > 0)
            store[w] = merged
        else:
            extra["word"] = w
            store.append(extra)
        seam.set_word_fpos(w, store[w])
    assert seam.commit(0) == 7
    wide = [dict(op=1, opts=opts, subs=[(0, 100.0), (6, 58.0)]), dict(op=1, opts=opts, subs=[(4, 72.0)]), dict(op=3, opts=opts, subs=[(2, 86.0)])]
    for q in (simple, two, wide):
        for packed in (True, False):
            _same(seam.merge(q, packed=packed, gpu=True), seam.merge(q, packed=packed, gpu=False), ("recommit", bm25_type, packed))
    # third commit: word 2 is replaced by a list of the SAME length with the SAME last entry but other earlier documents — what lands on an index
    # when the last step's words are erased and rebuilt (dataholder.cc:116).  The mirror hashes whole lists (packed and plain alike), so it travels.
    old = store[2]
    n = len(old["doc"])
    docs = np.asarray(old["doc"], np.uint32).copy()
    docs[: n - 1] = np.sort(rng.choice(np.arange(1, int(docs[-1])), n - 1, replace=False)).astype(np.uint32)
    assert not np.array_equal(docs, old["doc"]) and docs[-1] == old["doc"][-1] and np.all(np.diff(docs.astype(np.int64)) > 0)
    store[2] = dict(old, doc=docs)
    seam.set_word_fpos(2, store[2])
    assert seam.commit(0) == 7
    for q in (simple, wide):
        for packed in (True, False):
            _same(seam.merge(q, packed=packed, gpu=True), seam.merge(q, packed=packed, gpu=False), ("same-length replacement", bm25_type, packed))
    seam.close()


def _phrase_terms(terms, phrases, distances):
    return [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]], phrase=int(ph), distance=int(d))
            for t, ph, d in zip(terms, phrases, distances)]


PHRASE_SHAPES = [((1, 1, 1), (0, 0, -1), (1, 12, 1)), ((2, 2, 1, 1), (0, 0, -1, -1), (1, 20, 1, 1)), ((1, 1, 1, 1), (-1, 0, 0, 0), (1, 1, 15, 15))]


@pytest.mark.parametrize("ops,phrases,distances", PHRASE_SHAPES)
def test_reference_phrase_merger_over_packed_lists_equals_plain_lists(ft, ops, phrases, distances):
    """CPU: PhraseResults built by the shim like Selector::Process builds them; PhraseMerger walks PackedIdRelVec (occurrences moved out of
    the iterator) and IdRelVec (by reference) to the same result."""
    nf, total = 2, 3000
    _, words, avg, removed, excluded, terms, store = _multi_case(500 + len(ops), nf, total, 20000, ops, False, None, sizes=(400, 1500), nsub_range=(2, 4))
    seam = _seam(nf, words, avg, removed, store)
    seam.set_config(ft.default_config(nf, merge_limit=20000))
    q = _phrase_terms(terms, phrases, distances)
    plain = [dict(t, phrase=-1) for t in q]
    for exc in (None, excluded):
        a, b = seam.merge(q, exc, packed=True), seam.merge(q, exc, packed=False)
        _same(a, b, ops)
        assert len(a[0]) > 0
        c = seam.merge(plain, exc, packed=False)
        assert len(c[0]) != len(a[0]) or not np.array_equal(c[1], a[1])   # the phrase is not the same query as its terms
    seam.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ops,phrases,distances", PHRASE_SHAPES)
def test_patched_merge_results_branch_takes_phrases(rxgpu, ft, ops, phrases, distances):
    """Phrase queries through the patched Selector::mergeResults branch: ToGpuTerms hands the PhraseResults over term by term with a
    phrase number and FtDslOpts::distance, the PhraseMerger runs on the device — result identical to the reference's merger."""
    nf, total = 2, 3000
    _, words, avg, removed, excluded, terms, store = _multi_case(500 + len(ops), nf, total, 20000, ops, False, None, sizes=(400, 1500), nsub_range=(2, 4))
    seam = _seam(nf, words, avg, removed, store)
    assert seam.commit(0) == len(store)
    q = _phrase_terms(terms, phrases, distances)
    for limit in (20000, 80):
        seam.set_config(ft.default_config(nf, merge_limit=limit))
        for exc in (None, excluded):
            for packed in (True, False):
                want = seam.merge(q, exc, rank_sort_type=1, packed=packed, gpu=False)
                got = seam.merge(q, exc, rank_sort_type=1, packed=packed, gpu=True)
                _same(got, want, (limit, packed))
                assert len(want[0]) > 0
    seam.close()


def _syn_query(seed, nf, total, ops, syn_sizes, part_syn):
    n_syn_terms = sum(syn_sizes)
    _, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, 20000, tuple(ops) + (1,) * n_syn_terms, False, None, sizes=(300, 1200),
                                                                 nsub_range=(1, 4))
    def cv(t, op=None):
        return dict(op=t["op"] if op is None else op, opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]])
    parts = [cv(t) for t in terms[:len(ops)]]
    owner_op = {sid: parts[pi]["op"] for pi, ids in enumerate(part_syn) for sid in ids}
    synonyms, at = [], len(ops)
    for sid, k in enumerate(syn_sizes):
        synonyms.append([cv(t, owner_op.get(sid, 1)) for t in terms[at:at + k]])
        at += k
    # a duplicate: the first synonym term also finds a word of the first query term (SupressDuplicatesInSynonyms marks it)
    synonyms[0][0]["subs"] = sorted(synonyms[0][0]["subs"] + [(parts[0]["subs"][0][0], 21.0)], key=lambda x: -x[1])
    return words, avg, removed, excluded, store, parts, synonyms


SYN_SHAPES = [((1, 1), [2], [[0], []]), ((2, 1), [2, 2], [[0, 1], []]), ((1, 2, 1), [3], [[], [0], []])]


@pytest.mark.parametrize("ops,syn_sizes,part_syn", SYN_SHAPES)
def test_reference_synonym_merge_over_packed_lists_equals_plain_lists(ft, ops, syn_sizes, part_syn):
    nf, total = 2, 3000
    words, avg, removed, excluded, store, parts, synonyms = _syn_query(600 + len(ops), nf, total, ops, syn_sizes, part_syn)
    seam = _seam(nf, words, avg, removed, store)
    seam.set_config(ft.default_config(nf, merge_limit=20000))
    for exc in (None, excluded):
        a = seam.merge(parts, exc, packed=True, synonyms=synonyms, part_synonyms=part_syn)
        b = seam.merge(parts, exc, packed=False, synonyms=synonyms, part_synonyms=part_syn)
        _same(a, b, ops)
        c = seam.merge(parts, exc, packed=False)
        assert len(a[0]) > 0 and (len(c[0]) != len(a[0]) or not np.array_equal(c[1], a[1]))   # the synonyms change the result
    seam.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ops,syn_sizes,part_syn", SYN_SHAPES)
def test_patched_merge_results_branch_takes_multi_word_synonyms(rxgpu, ft, ops, syn_sizes, part_syn):
    """QueryMergeData::synonyms through the patched Selector::mergeResults branch: ToGpuTerms hands the synonyms, the parts' SynonymsIds and
    the Suppressed() marks over, the device merges — identical to the reference's merger for both containers."""
    nf, total = 2, 3000
    words, avg, removed, excluded, store, parts, synonyms = _syn_query(600 + len(ops), nf, total, ops, syn_sizes, part_syn)
    seam = _seam(nf, words, avg, removed, store)
    assert seam.commit(0) == len(store)
    for limit in (20000, 70):
        seam.set_config(ft.default_config(nf, merge_limit=limit))
        for exc in (None, excluded):
            for packed in (True, False):
                want = seam.merge(parts, exc, rank_sort_type=1, packed=packed, gpu=False, synonyms=synonyms, part_synonyms=part_syn)
                got = seam.merge(parts, exc, rank_sort_type=1, packed=packed, gpu=True, synonyms=synonyms, part_synonyms=part_syn)
                _same(got, want, (limit, packed))
                assert len(want[0]) > 0
    seam.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ops,syn_sizes,part_syn", SYN_SHAPES)
def test_patched_merge_results_branch_over_a_device_list(rxgpu, ft, ops, syn_sizes, part_syn):
    """The patched Selector::mergeResults branch with the mirror over a DEVICE LIST (RX_GPU_FT_INDEXES=0,0,0: document-range shards):
    term queries, multi-word synonyms and phrases are merged on the shards — identical to the reference's merger, both containers."""
    nf, total = 2, 30_000
    n_syn_terms = sum(syn_sizes)
    _, words, avg, removed, excluded, terms, store = _multi_case(640 + len(ops), nf, total, 20000, tuple(ops) + (1,) * n_syn_terms, False, None,
                                                                 sizes=(2000, 9000), nsub_range=(1, 4))

    def cv(t, op=None):
        return dict(op=t["op"] if op is None else op, opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]])

    parts = [cv(t) for t in terms[:len(ops)]]
    owner_op = {sid: parts[pi]["op"] for pi, ids in enumerate(part_syn) for sid in ids}
    synonyms, at = [], len(ops)
    for sid, k in enumerate(syn_sizes):
        synonyms.append([cv(t, owner_op.get(sid, 1)) for t in terms[at:at + k]])
        at += k
    seam = _seam(nf, words, avg, removed, store)
    assert seam.commit(devices=[0, 0, 0]) == len(store)
    for limit in (20000, 400):
        seam.set_config(ft.default_config(nf, merge_limit=limit))
        for exc in (None, excluded):
            for packed in (True, False):
                for syn in (None, synonyms):
                    kw = dict(synonyms=syn, part_synonyms=part_syn) if syn else {}
                    want = seam.merge(parts, exc, rank_sort_type=1, packed=packed, gpu=False, **kw)
                    got = seam.merge(parts, exc, rank_sort_type=1, packed=packed, gpu=True, **kw)
                    _same(got, want, (limit, packed, syn is not None))
                    assert len(want[0]) > 0
    if len(parts) >= 2:
        # a phrase: PhraseMerger runs on every shard, the admission cut of the whole index is settled between the shards
        phrase = [dict(parts[0], phrase=0, distance=12), dict(parts[1], op=parts[0]["op"], phrase=0, distance=12)] + [dict(p, phrase=-1) for p in parts[2:]]
        for limit in (60000, 400):   # 400: the PhraseMerger's admission cut falls inside the first term (settled between the shards)
            seam.set_config(ft.default_config(nf, merge_limit=limit))
            for packed in (True, False):
                want = seam.merge(phrase, excluded, rank_sort_type=1, packed=packed, gpu=False)
                got = seam.merge(phrase, excluded, rank_sort_type=1, packed=packed, gpu=True)
                _same(got, want, ("phrase", limit, packed))
                assert len(want[0]) > 0
    seam.close()
