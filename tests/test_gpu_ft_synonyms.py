"""-m gpu: MULTI-WORD SYNONYMS in the ft_fast merge on the GPU (rxgpu_ft_merge_query2_raw / GpuFtMerger::MergeQuery with QuerySynonyms) against
the REAL reference merger — ft::Merger::Merge with QueryMergeData::synonyms (mergerimpl.h:347-361 the synonyms' masks in the restricting
bitmask, :393-397 their terms in the pre-scores, merger.h:251-255 in the 2-phase estimate, :509-555 mergeTerm behind the query parts, the
term counting, containsFullMultiWordSynonym, the removal of documents that hold only parts of a synonym; SupressDuplicatesInSynonyms,
querymergedata.h:221-241) compiled in place (oracle/_ref/libref_ft.so).
Bar: the same documents in the same merge order, the same raw-rank bits, fields and uint8 ranks."""
import numpy as np
import pytest

from oracle.pyoracle import FtOracle, ref_ft_or_none
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


def _t(t, op=None, phrase=-1, distance=1):
    return dict(op=t["op"] if op is None else op, opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]], phrase=phrase, distance=distance)


def _compare(real, m, ft, nf, limit, parts, synonyms, part_syn, excluded, variants=((1.0, 0.5), (1.7, 0.8)), min_results=1):
    most = 0
    for variant, (dboost, dweight) in enumerate(variants):
        cfg = ft.default_config(nf, merge_limit=limit, min_rank=5 if variant != 1 else 40)
        cfg["distance_boost"], cfg["distance_weight"] = dboost, dweight
        real.set_config(cfg, distance_boost=dboost, distance_weight=dweight)
        for exc in (None, excluded):
            wd, wp, wf, wn = real.merge(parts, exc, rank_sort_type=1, synonyms=synonyms, part_synonyms=part_syn)
            gd, gp, gf, gn, _ = m.merge_query(cfg, parts, exc, sort_by_rank=False, synonyms=synonyms, part_synonyms=part_syn)
            assert np.array_equal(gd, wd), (variant, len(gd), len(wd), gd[:8], wd[:8])
            assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32)), (variant, gp[:8], wp[:8])
            assert np.array_equal(gn, wn) and np.array_equal(gf, wf)
            most = max(most, len(wd))
    assert most >= min_results, most
    return most


SYN_CASES = [
    # (seed, nf, total, limit, ops of the query parts, synonyms as lists of term counts, part -> synonym ids)
    (201, 2, 3000, 20000, (1,), [2], [[0]]),                         # one OR term with a two-word synonym
    (202, 2, 3000, 20000, (1, 1), [2, 3], [[0], [1]]),               # every part has its own synonym
    (203, 2, 3000, 20000, (2, 1), [2], [[0], []]),                   # AND part: the synonym's mask is OR-ed into the restriction
    (204, 2, 3000, 20000, (2, 2), [2, 2, 2], [[0, 1], [2]]),         # two synonyms on one AND part
    (205, 3, 3000, 20000, (1, 3, 1), [2], [[0], [], []]),            # a NOT part among them (queryParts.size() > merged parts: no full-match boost)
    (206, 2, 3000, 60, (1, 1), [2, 2], [[0], [1]]),                  # mergeLimit + the preselect path with synonym terms in the scores
    (207, 2, 3000, 45, (2, 1), [2], [[0], []]),                      # ... with an AND part
    (208, 1, 3000, 20000, (1, 1, 1), [2], [[], [0], []]),            # the synonym hangs on the middle part
]


@pytest.mark.parametrize("seed,nf,total,limit,ops,syn_sizes,part_syn", SYN_CASES)
def test_gpu_synonym_merge_equals_real_merger(hostapi, ft, seed, nf, total, limit, ops, syn_sizes, part_syn):
    n_syn_terms = sum(syn_sizes)
    _, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, limit, tuple(ops) + (1,) * n_syn_terms, False, None, sizes=(300, 1200),
                                                                 nsub_range=(1, 4))
    parts = [_t(t) for t in terms[:len(ops)]]
    synonyms, at = [], len(ops)
    owner_op = {sid: parts[pi]["op"] for pi, ids in enumerate(part_syn) for sid in ids}
    for sid, k in enumerate(syn_sizes):
        synonyms.append([_t(t, op=owner_op.get(sid, 1)) for t in terms[at:at + k]])   # a synonym's terms carry the options of the term they replace
        at += k
    real, m = _engines(hostapi, nf, words, avg, removed, store)
    try:
        plain = _compare(real, m, ft, nf, limit, parts, None, None, excluded, variants=((1.0, 0.5),))
        with_syn = _compare(real, m, ft, nf, limit, parts, synonyms, part_syn, excluded)
        assert with_syn >= 1 and (limit < 1000 or with_syn >= plain)
    finally:
        real.close()
        m.close()


def test_gpu_synonym_suppressed_subterms(hostapi, ft):
    """A synonym term whose sub-terms include words the query's own terms found (SupressDuplicatesInSynonyms): such a sub-term adds no
    document and no rank, it only counts the term for documents that are merged already (mergerimpl.h:144-151)."""
    nf, total = 2, 3000
    _, words, avg, removed, excluded, terms, store = _multi_case(210, nf, total, 20000, (1, 1, 1, 1), False, None, sizes=(400, 1500), nsub_range=(2, 4))
    parts = [_t(terms[0]), _t(terms[1])]
    syn_a, syn_b = _t(terms[2]), _t(terms[3])
    # the first synonym term also lists the words of part 0 (lower procs, behind its own), the second one a word of part 1
    syn_a["subs"] = sorted(syn_a["subs"] + [(w, p * 0.5) for w, p in parts[0]["subs"]], key=lambda x: -x[1])
    syn_b["subs"] = sorted(syn_b["subs"] + [(parts[1]["subs"][0][0], 33.0)], key=lambda x: -x[1])
    real, m = _engines(hostapi, nf, words, avg, removed, store)
    try:
        for limit in (20000, 90):
            _compare(real, m, ft, nf, limit, parts, [[syn_a, syn_b]], [[0], []], excluded)
            _compare(real, m, ft, nf, limit, [dict(parts[0], op=2), parts[1]], [[syn_a, syn_b]], [[0], []], excluded)
    finally:
        real.close()
        m.close()


def test_gpu_synonyms_with_a_phrase_part(hostapi, ft):
    nf, total = 2, 3000
    _, words, avg, removed, excluded, terms, store = _multi_case(220, nf, total, 20000, (1, 1, 1, 1, 1), False, None, sizes=(400, 1500), nsub_range=(2, 4))
    parts = [_t(terms[0], phrase=0, distance=1), _t(terms[1], phrase=0, distance=20), _t(terms[2])]   # phrase (part 0), term (part 1)
    synonyms = [[_t(terms[3]), _t(terms[4])]]
    real, m = _engines(hostapi, nf, words, avg, removed, store)
    try:
        _compare(real, m, ft, nf, 20000, parts, synonyms, [[], [0]], excluded)
        _compare(real, m, ft, nf, 20000, parts, synonyms, [[0], []], excluded)                      # the synonym hangs on the phrase
        and_phrase = [dict(parts[0], op=2), dict(parts[1], op=2), parts[2]]
        _compare(real, m, ft, nf, 20000, and_phrase, synonyms, [[0], []], excluded)                 # ... an AND phrase: its mask takes the synonym's in
    finally:
        real.close()
        m.close()


def test_gpu_synonyms_long_lists(hostapi, ft):
    """Posting lists of 20-60 K documents over 200 K: many document ranges, the synonym masks and the removal of partial documents at scale."""
    nf, total = 2, 200_000
    _, words, avg, removed, excluded, terms, store = _multi_case(230, nf, total, 20000, (2, 1, 1, 1, 1), False, None, sizes=(20_000, 60_000), nsub_range=(2, 4))
    parts = [_t(terms[0]), _t(terms[1])]
    synonyms = [[_t(terms[2], op=2), _t(terms[3], op=2)], [_t(terms[4])]]
    real, m = _engines(hostapi, nf, words, avg, removed, store)
    try:
        _compare(real, m, ft, nf, 20000, parts, synonyms, [[0], [1]], excluded, variants=((1.0, 0.5),), min_results=1000)
        _compare(real, m, ft, nf, 1500, parts, synonyms, [[0], [1]], excluded, variants=((1.0, 0.5),), min_results=100)
    finally:
        real.close()
        m.close()


def test_gpu_random_query_shapes_equal_real_merger(hostapi, ft):
    """Randomised differential run: 60 queries of random shape over one corpus — 1-5 parts (terms and phrases, OR / AND / NOT), 0-3 multi-word
    synonyms of 1-3 terms hung on random parts (some sharing words with the query's own terms), zero field boosts, merge limits from 30
    to 20000, docsExcluded on and off — each against the real merger."""
    nf, total = 3, 6000
    rng = np.random.default_rng(777)
    _, words, avg, removed, excluded, pool, store = _multi_case(778, nf, total, 20000, (1,) * 18, True, None, sizes=(200, 2500), nsub_range=(1, 5))
    real, m = _engines(hostapi, nf, words, avg, removed, store)
    boosts = [[1.0, 1.0, 1.0], [1.0, 0.0, 2.0], [0.0, 1.0, 0.5], [1.0, 1.0, 0.0]]
    try:
        checked = nonempty = 0
        for qi in range(60):
            order = rng.permutation(len(pool))
            take = iter(order)
            parts, nparts = [], int(rng.integers(1, 6))
            part_ops = []
            for pi in range(nparts):
                op = int(rng.choice([1, 1, 1, 2, 2, 3]))
                part_ops.append(op)
                if rng.random() < 0.3:   # a phrase of 2-3 terms
                    k = int(rng.integers(2, 4))
                    dist = int(rng.choice([1, 3, 10, 40]))
                    for j in range(k):
                        t = pool[next(take)]
                        parts.append(dict(_t(t, op=op), phrase=pi, distance=1 if j == 0 else dist))
                else:
                    parts.append(_t(pool[next(take)], op=op))
            if all(o == 3 for o in part_ops):
                parts[0]["op"] = 1
                part_ops[0] = 1
                if parts[0]["phrase"] >= 0:
                    for p_ in parts:
                        if p_["phrase"] == parts[0]["phrase"]:
                            p_["op"] = 1
            for p_ in parts:
                fb = boosts[int(rng.integers(0, len(boosts)))] if rng.random() < 0.3 else [1.0] * nf
                p_["opts"] = dict(p_["opts"], field_boost=fb)
            nsyn = int(rng.integers(0, 4))
            synonyms, part_syn = [], [[] for _ in range(nparts)]
            for sid in range(nsyn):
                owner = int(rng.integers(0, nparts))
                syn = []
                for _ in range(int(rng.integers(1, 4))):
                    st = _t(pool[next(take)], op=part_ops[owner] if part_ops[owner] != 3 else 1)
                    if rng.random() < 0.4:   # shares a word with one of the query's own terms
                        src = parts[int(rng.integers(0, len(parts)))]
                        if src["subs"]:
                            st["subs"] = sorted(st["subs"] + [(src["subs"][0][0], float(rng.choice([15.0, 33.0, 61.0])))], key=lambda x: -x[1])
                    syn.append(st)
                synonyms.append(syn)
                part_syn[owner].append(sid)
            limit = int(rng.choice([30, 150, 1000, 20000]))
            cfg = ft.default_config(nf, merge_limit=limit, min_rank=int(rng.choice([0, 5, 40])))
            dboost, dweight = (1.0, 0.5) if qi % 3 else (1.7, 0.8)
            cfg["distance_boost"], cfg["distance_weight"] = dboost, dweight
            real.set_config(cfg, distance_boost=dboost, distance_weight=dweight)
            exc = excluded if qi % 2 else None
            kw = dict(synonyms=synonyms, part_synonyms=part_syn) if synonyms else {}
            wd, wp, wf, wn = real.merge(parts, exc, rank_sort_type=1, **kw)
            gd, gp, gf, gn, _ = m.merge_query(cfg, parts, exc, sort_by_rank=False, **kw)
            tag = (qi, part_ops, [p_["phrase"] for p_ in parts], part_syn, limit)
            assert np.array_equal(gd, wd), (tag, len(gd), len(wd), gd[:6], wd[:6])
            assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32)), tag
            assert np.array_equal(gn, wn) and np.array_equal(gf, wf), tag
            checked += 1
            nonempty += len(wd) > 0
        assert checked == 60 and nonempty >= 30, (checked, nonempty)
    finally:
        real.close()
        m.close()


@pytest.mark.parametrize("seed,ops,syn_sizes,part_syn,limit", [(231, (1, 1), [2, 3], [[0], [1]], 20000), (232, (2, 1), [2], [[0], []], 20000),
                                                               (233, (1, 1), [2, 2], [[0], [1]], 60)])
def test_hybrid_query_with_synonyms_stays_resident(hostapi, ft, seed, ops, syn_sizes, part_syn, limit):
    """A hybrid query whose FT half has multi-word synonyms (round 4: rxgpu_ft_merge_query2_resident): the merge stays in HBM with the
    documents Merger::Merge removes (they hold only parts of a synonym, mergerimpl.h:533-555) still marked, the fusion kernels treat them
    as absent — before postProcessResults, as the reference does.  Bar: the fusion of the separate product calls (MergeQuery with the
    synonyms — itself pinned to the real merger above — + the Map's select + MergeRanked), RRF and linear, union and intersection."""
    from .conftest import make_corpus
    nf, total, d, k = 2, 3000, 32, 40
    n_syn_terms = sum(syn_sizes)
    _, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, limit, tuple(ops) + (1,) * n_syn_terms, False, None, sizes=(300, 1200),
                                                                 nsub_range=(1, 4))
    parts = [_t(t) for t in terms[:len(ops)]]
    synonyms, at = [], len(ops)
    owner_op = {sid: parts[pi]["op"] for pi, ids in enumerate(part_syn) for sid in ids}
    for sid, n_ in enumerate(syn_sizes):
        synonyms.append([_t(t, op=owner_op.get(sid, 1)) for t in terms[at:at + n_]])
        at += n_
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg, removed)
    for s in store:
        m.set_word_fpos(s["word"], s)
    rows = make_corpus(seed, total, d)
    vm = hostapi.GpuBruteforceMap(2, d, total)
    vm.add(rows, np.arange(total, dtype=np.uint64) << np.uint64(32))
    cfg = ft.default_config(nf, merge_limit=limit, min_rank=5)
    try:
        removed_some = False
        for qi in range(4):
            key = make_corpus(1000 * seed + qi, 1, d)[0]
            fid, fproc, _, _, _ = m.merge_query(cfg, parts, None, sort_by_rank=True, synonyms=synonyms, part_synonyms=part_syn)
            plain = m.merge_query(cfg, parts, None, sort_by_rank=True)[0]
            removed_some = removed_some or len(fid) != len(plain)
            kid, krank = vm.select(key, k=k, need_sort=False)
            for kind, params in (("rrf", [60.0]), ("linear", [0.7, 0.3, 1.0, 0.0, 0.0])):
                for union in (True, False):
                    got = hostapi.hybrid_query_resident(vm, m, cfg, parts, key, k, kind=kind, params=params, union=union, desc=True, synonyms=synonyms,
                                                        part_synonyms=part_syn)
                    ids, ranks = hostapi.merge_ranked(kind, params, kid, krank, fid, fproc, union=union, desc=True, metric=2, ft_order="rank")
                    assert np.array_equal(got[0], ids), (qi, kind, union, len(got[0]), len(ids))
                    assert np.array_equal(got[1].view(np.uint32), ranks.view(np.uint32)), (qi, kind, union)
        assert len(fid) > 0
    finally:
        m.close()
        vm.close()
