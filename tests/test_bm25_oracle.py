"""CPU: pins the BM25 restatement (oracle/oracle_bm25.c) to the reference's OWN golden vectors — the debug_rank() strings of
cpp_src/gtests/tests/unit/ft/ft_generic.cc:326-443 (test FTGenericApi.DebugInfo).  fmt prints floats in shortest round-trip
form, so parsing the printed value back to float32 must give EXACTLY the float the merger computed.

Corpus of that test (5 docs + the empty sentinel vdoc, one FT field, default FTConfig):
  1 "Маша ела кашу. Каша кушалась сама. Машу ругали."            8 words
  2 "Коля, Сеня гуляли."                                         3 words
  3 "слово простая фраза что то еще."                            6 words
  4 "слово начало простая фраза конец что то еще простая фраза слово слово."   12 words
  5 "жил пил гулял"                                              3 words          => average 6.4 words
"""
from pathlib import Path

import numpy as np
import pytest

from oracle.pyoracle import FtOracle

AVG = np.float32(32.0 / 5.0)
N_DOCS = 5   # totalNumDocs_ - 1 ("first doc is always empty", mergerimpl.h:122-124)

# (matched docs M, tf, words in field, first position, proc, termLenBoost fed by the DSL)  ->  golden strings
KATS = [
    # ft_generic.cc:326-327  "маша"
    dict(M=1, tf=1, words=8, pos=0, proc=100.0, tlb_in=1.0, bm25_norm="0.979844", tlb="1", prank="1", term_rank="97.9844"),
    dict(M=1, tf=1, words=8, pos=6, proc=80.0, tlb_in=1.0, bm25_norm="0.979844", tlb="1", prank="0.994", term_rank="77.91719"),
    # :346-347  "коля сеня"
    dict(M=1, tf=1, words=3, pos=0, proc=100.0, tlb_in=1.0, bm25_norm="1.0223141", tlb="1", prank="1", term_rank="102.23141"),
    dict(M=1, tf=1, words=3, pos=1, proc=100.0, tlb_in=1.0, bm25_norm="1.0223141", tlb="1", prank="0.999", term_rank="102.12917"),
    # :363-375  phrases over docs 3 and 4
    dict(M=2, tf=1, words=6, pos=1, proc=100.0, tlb_in=1.0, bm25_norm="0.9399332", tlb="1", prank="0.999", term_rank="93.89933"),
    dict(M=2, tf=1, words=6, pos=2, proc=100.0, tlb_in=5.0 / 7.0, bm25_norm="0.9399332", tlb="0.9142857", prank="0.998", term_rank="85.76488"),
    dict(M=1, tf=1, words=12, pos=1, proc=100.0, tlb_in=6.0 / 7.0, bm25_norm="0.96248657", tlb="0.95714283", prank="0.999", term_rank="92.031586"),
    dict(M=2, tf=2, words=12, pos=2, proc=100.0, tlb_in=1.0, bm25_norm="0.9436916", tlb="1", prank="0.998", term_rank="94.18042"),
    dict(M=2, tf=2, words=12, pos=3, proc=100.0, tlb_in=5.0 / 7.0, bm25_norm="0.9436916", tlb="0.9142857", prank="0.997", term_rank="86.02153"),
    dict(M=1, tf=1, words=12, pos=4, proc=100.0, tlb_in=5.0 / 7.0, bm25_norm="0.96248657", tlb="0.9142857", prank="0.996", term_rank="87.646774"),
    # :402-410  stemmed variants with fractional proc
    dict(M=2, tf=1, words=6, pos=1, proc=79.0, tlb_in=1.0, bm25_norm="0.9399332", tlb="1", prank="0.999", term_rank="74.180466"),
    dict(M=2, tf=1, words=6, pos=2, proc=81.25, tlb_in=7.0 / 8.0, bm25_norm="0.9399332", tlb="0.9625", prank="0.998", term_rank="73.3587"),
    dict(M=2, tf=2, words=12, pos=2, proc=79.0, tlb_in=1.0, bm25_norm="0.9436916", tlb="1", prank="0.998", term_rank="74.402534"),
    dict(M=2, tf=2, words=12, pos=3, proc=81.25, tlb_in=7.0 / 8.0, bm25_norm="0.9436916", tlb="0.9625", prank="0.997", term_rank="73.57823"),
    # :441-443  "жил~ пил" (typo variant with proc 75.99915)
    dict(M=1, tf=1, words=3, pos=1, proc=75.99915, tlb_in=1.0, bm25_norm="1.0223141", tlb="1", prank="0.999", term_rank="77.61731"),
]


def f32(s):
    return np.float32(float(s))


@pytest.fixture(scope="module")
def ft(oracle):
    return FtOracle(oracle)


@pytest.mark.parametrize("kat", KATS, ids=[f"{k['term_rank']}" for k in KATS])
def test_term_rank_matches_reference_golden_strings(ft, kat):
    cfg, opts = ft.default_config(1), ft.default_opts(1, term_len_boost=np.float32(kat["tlb_in"]))
    idf = ft.idf(N_DOCS, kat["M"])
    rank, field, bm25n, tlb, prank = ft.term_rank(cfg, opts, idf, np.float32(kat["proc"]), [0], [kat["tf"]], [kat["pos"]],
                                                  [np.float32(kat["words"])], [AVG])
    assert field == 0
    assert bm25n == f32(kat["bm25_norm"]), (bm25n, kat["bm25_norm"])
    assert tlb == f32(kat["tlb"])
    assert prank == f32(kat["prank"])
    assert rank == f32(kat["term_rank"]), (rank, kat["term_rank"])


def test_idf_floor_and_shape(ft):
    assert ft.idf(5, 1) == pytest.approx(np.log(5) / np.log(6))
    assert ft.idf(1000, 999) == 0.2          # "saturate min to 0.2" (bm25.h:22-25)
    assert ft.idf(10, 10) == 0.2


def make_postings(rng, total_docs, nfields, n, max_tf=4):
    docs = np.sort(rng.choice(np.arange(1, total_docs), n, replace=False)).astype(np.uint32)
    ent_off = [0]
    ef, et, ep = [], [], []
    for _ in range(n):
        fields = np.sort(rng.choice(nfields, rng.integers(1, min(nfields, 3) + 1), replace=False))
        for f in fields:
            ef.append(f)
            et.append(rng.integers(1, max_tf + 1))
            ep.append(rng.integers(0, 300))
        ent_off.append(len(ef))
    return dict(doc=docs, ent_off=np.array(ent_off, np.uint32), ent_field=np.array(ef, np.uint8), ent_tf=np.array(et, np.uint32),
                ent_first_pos=np.array(ep, np.uint32))


def test_merge_simple_semantics(ft):
    """max over sub-terms with the first max winning, mergeLimit in sequence order, removed/excluded docs, minRank filter,
    full-match boost, 0..255 normalisation."""
    rng = np.random.default_rng(1)
    total, nf = 400, 3
    words = rng.integers(1, 30, (total, nf)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    removed = np.zeros(total, np.uint8)
    removed[rng.choice(total, 20, replace=False)] = 1
    excluded = np.zeros(total, np.uint8)
    excluded[rng.choice(total, 20, replace=False)] = 1
    subs = []
    for proc in (100.0, 85.0, 60.0):
        s = make_postings(rng, total, nf, 150)
        s["proc"] = proc
        subs.append(s)
    cfg = ft.default_config(nf, merge_limit=120)
    opts = ft.default_opts(nf, field_boost=[1.0, 0.5, 0.0])
    doc, proc, field, norm = ft.merge_simple(cfg, opts, total, words, avg, removed, excluded, subs, sort_by_rank=False)
    assert 0 < doc.shape[0] <= 120
    assert not removed[doc].any() and not excluded[doc].any()
    assert np.all(field != 2)                                      # zero-boost field never wins
    assert norm.max() == 255 or proc.max() <= 255                  # scaled only when the raw max exceeds 255
    assert np.all(norm == proc.astype(np.uint8))
    # admitted docs are the first `merge_limit` distinct valid docs in (sub-term, posting) order
    seen = []
    for s in subs:
        for i, d in enumerate(s["doc"]):
            fields = s["ent_field"][s["ent_off"][i]:s["ent_off"][i + 1]]
            if removed[d] or excluded[d] or d in seen or np.all(fields == 2):   # only zero-boost fields => rank 0 => skipped
                continue
            seen.append(int(d))
    assert set(doc.tolist()) <= set(seen[:120])
    doc2, proc2, field2, norm2 = ft.merge_simple(cfg, opts, total, words, avg, removed, excluded, subs, sort_by_rank=True)
    assert sorted(doc2.tolist()) == sorted(doc.tolist()) and np.all(np.diff(norm2.astype(int)) <= 0)


# ------------------------------------------------------------------------------------------- vs the REAL ft::Merger (oracle/_ref)
def _by_doc(ids, norm, field):
    o = np.argsort(ids, kind="stable")
    return ids[o], norm[o], field[o]


@pytest.mark.parametrize("nf,limit", [(1, 20000), (3, 20000), (3, 120), (4, 37)])
def test_restated_merge_simple_equals_real_merger(ft, nf, limit):
    """The plain-C restatement vs reindexer::ft::Merger itself (compiled in place from the reference tree, libref_ft.so)."""
    from oracle.pyoracle import ref_ft_or_none
    real = ref_ft_or_none(nf)
    if real is None:
        pytest.skip("oracle/_ref/libref_ft.so not available")
    rng = np.random.default_rng(nf * 1000 + limit)
    total = 2500
    words = rng.integers(1, 40, (total, nf)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    removed = np.zeros(total, np.uint8)
    removed[rng.choice(total, 80, replace=False)] = 1
    excluded = np.zeros(total, np.uint8)
    excluded[rng.choice(total, 80, replace=False)] = 1
    real.set_docs(words, avg, removed)
    subs = []
    for wid, proc in enumerate((100.0, 91.5, 77.0, 60.25)):
        s = make_postings(rng, total, nf, int(rng.integers(150, 1200)))
        s["proc"] = proc
        subs.append(s)
        real.set_word_flat(wid, s)
    for variant in range(3):
        cfg = ft.default_config(nf, merge_limit=limit)
        opts = ft.default_opts(nf, field_boost=[1.0, 0.6, 0.0, 1.4][:nf], boost=1.0 + 0.3 * variant, term_len_boost=0.85)
        if variant == 2 and nf > 1:
            cfg["summation_ratio"] = 0.4
            opts["need_sum_rank"] = [1] * nf
        real.set_config(cfg)
        for exc in (None, excluded):
            wd, wp, wf, wn = real.merge([dict(op=real.OP_OR, opts=opts, subs=[(i, s["proc"]) for i, s in enumerate(subs)])], exc, rank_sort_type=1)
            gd, gp, gf, gn = ft.merge_simple(cfg, opts, total, words, avg, removed, exc, subs, sort_by_rank=False)
            assert np.array_equal(gd.astype(np.int32), wd), (nf, limit, variant)      # same docs in the same merge order
            assert np.array_equal(gn, wn) and np.array_equal(gf, wf)
            assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32))
            # sorted flavour: pdqsort is unstable in the reference => compare as (doc -> rank) maps
            wd2, _, wf2, wn2 = real.merge([dict(op=real.OP_OR, opts=opts, subs=[(i, s["proc"]) for i, s in enumerate(subs)])], exc, rank_sort_type=0)
            assert np.all(np.diff(wn2.astype(int)) <= 0)
            a, b = _by_doc(wd2, wn2, wf2), _by_doc(gd.astype(np.int32), gn, gf)
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
    real.close()


# ------------------------------------------------------------------------------------------- multi-term (mergeTerm) vs the real merger
def make_pos_postings(rng, total, nf, n, proc, array_fields=False, max_pos=40):
    """A sub-term in positions format: ascending docs, per doc 1..5 PosType words sorted like IdRelType::SortAndUnique."""
    from oracle.pyoracle import make_fpos
    doc = np.sort(rng.choice(np.arange(1, total), n, replace=False)).astype(np.uint32)
    k = rng.integers(1, 6, n)
    owner = np.repeat(np.arange(n), k)
    m = owner.shape[0]
    w = make_fpos(rng.integers(0, max_pos, m), rng.integers(0, nf, m), rng.integers(0, 3, m) if array_fields else np.zeros(m, np.int64))
    order = np.lexsort((w, owner))
    owner, w = owner[order], w[order]
    keep = np.ones(m, bool)
    keep[1:] = (owner[1:] != owner[:-1]) | (w[1:] != w[:-1])          # SortAndUnique per posting
    owner, w = owner[keep], w[keep]
    pos_off = np.zeros(n + 1, np.int64)
    np.add.at(pos_off, owner + 1, 1)
    return dict(doc=doc, pos_off=np.cumsum(pos_off).astype(np.uint32), fpos=w.astype(np.uint64), proc=proc)


def _multi_case(seed, nf, total, limit, ops, array_fields=False, field_boosts=None, sizes=(150, 900), nsub_range=(1, 4)):
    from oracle.pyoracle import ref_ft_or_none
    rng = np.random.default_rng(seed)
    words = rng.integers(1, 6, (total, nf)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    removed = np.zeros(total, np.uint8)
    removed[rng.choice(total, total // 30, replace=False)] = 1
    excluded = np.zeros(total, np.uint8)
    excluded[rng.choice(total, total // 30, replace=False)] = 1
    terms, wid, word_store = [], 0, []
    for ti, op in enumerate(ops):
        nsub = int(rng.integers(*nsub_range))
        procs = sorted((float(rng.choice([100.0, 85.0, 70.5, 55.0, 40.0])) for _ in range(nsub)), reverse=True)
        subs = []
        for pr in procs:
            s = make_pos_postings(rng, total, nf, int(rng.integers(*sizes)), pr, array_fields)
            s["word"] = wid
            word_store.append(s)
            wid += 1
            subs.append(s)
        fb = field_boosts[ti] if field_boosts else [1.0] * nf
        terms.append(dict(op=op, opts=dict(boost=float(rng.choice([1.0, 1.3, 0.7])), term_len_boost=float(rng.choice([1.0, 0.8])),
                                            field_boost=fb, need_sum_rank=[0] * nf), subs=subs))
    return ref_ft_or_none, words, avg, removed, excluded, terms, word_store


MULTI_CASES = [
    # (seed, nf, total, merge_limit, ops, array_fields, field_boosts)
    (1, 1, 3000, 20000, (1, 1), False, None),
    (2, 3, 3000, 20000, (1, 1, 1), False, None),
    (3, 3, 3000, 20000, (2, 1), False, None),
    (4, 2, 3000, 20000, (1, 2, 3), False, None),
    (5, 4, 3000, 20000, (2, 2), True, None),
    (6, 3, 3000, 20000, (1, 3, 1), True, [[1.0, 0.0, 2.0], [1.0, 1.0, 1.0], [0.0, 0.5, 1.0]]),
    (7, 3, 3000, 20000, (2, 1), False, [[0.0, 1.0, 0.0], [1.0, 0.0, 0.5]]),
    (8, 2, 3000, 150, (1, 1), False, None),            # mergeLimit reached inside mergeTerm AND the preselect path
    (9, 3, 3000, 97, (1, 1, 1), True, [[1.0, 0.5, 2.0], [1.0, 1.0, 1.0], [3.0, 0.5, 1.0]]),
    (10, 2, 3000, 60, (2, 1, 3), False, None),
    (11, 1, 3000, 500, (1, 1, 1, 1), False, None),
    (13, 2, 3000, 40, (2, 2), False, None),             # AND-only: the estimate is 0 => no preselect, the limit cuts inside mergeTerm
    (14, 3, 3000, 25, (2, 2, 3), True, None),
    (12, 2, 3000, 20000, (3, 1), False, None),          # a NOT term first: queryParts.size() > #merged terms => no full-match boost
]


@pytest.mark.parametrize("seed,nf,total,limit,ops,arr,fbs", MULTI_CASES)
def test_restated_multi_term_merge_equals_real_merger(ft, seed, nf, total, limit, ops, arr, fbs):
    ref_ft_or_none, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, limit, ops, arr, fbs)
    real = ref_ft_or_none(nf)
    if real is None:
        pytest.skip("oracle/_ref/libref_ft.so not available")
    real.set_docs(words, avg, removed)
    for s in store:
        real.set_word_fpos(s["word"], s)
    saw_pre = False
    for variant, (dboost, dweight) in enumerate([(1.0, 0.5), (1.7, 0.8), (0.0, 1.0)]):
        cfg = ft.default_config(nf, merge_limit=limit, min_rank=5 if variant != 1 else 60)
        real.set_config(cfg, distance_boost=dboost, distance_weight=dweight)
        rterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
        for exc in (None, excluded):
            wd, wp, wf, wn = real.merge(rterms, exc, rank_sort_type=1)
            gd, gp, gf, gn, pre = ft.merge_query(cfg, terms, total, words, avg, removed, exc, sort_by_rank=False, distance_boost=dboost,
                                                 distance_weight=dweight)
            saw_pre |= pre
            assert np.array_equal(gd.astype(np.int32), wd), (seed, variant, len(gd), len(wd))
            assert np.array_equal(gn, wn) and np.array_equal(gf, wf)
            assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32))
    assert saw_pre == (limit < 1000 and 1 in ops)     # the small-limit cases must have gone through preselectMostRelevantDocs
    real.close()


def test_positions_distance_restatement(ft):
    from oracle.pyoracle import make_fpos
    a = make_fpos([3, 9], [0, 0])
    assert ft.positions_distance(a, make_fpos([5], [0])) == 2
    assert ft.positions_distance(a, make_fpos([9], [0])) == 0
    assert ft.positions_distance(a, make_fpos([10], [0])) == 1
    assert ft.positions_distance(a, make_fpos([4], [1])) == 0          # other field: no distance -> 0 ("zero for first occurence in field")
    assert ft.positions_distance(make_fpos([], []), make_fpos([4], [1])) == 0


# ------------------------------------------------------------------------------------------- committed fixtures of the real reference (tests/golden/ft.npz)
FT_GOLDEN = Path(__file__).resolve().parent / "golden" / "ft.npz"


def _unpack(data, afp):
    import ctypes as C
    from reindexer_amd import hostapi
    L = hostapi.lib()
    L.rxhost_ft_unpack.restype = C.c_long
    L.rxhost_ft_unpack.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    data = np.ascontiguousarray(data, np.uint8)
    npos = C.c_size_t(0)
    n = L.rxhost_ft_unpack(data.ctypes.data, data.shape[0], afp, None, None, None, C.byref(npos))
    if n < 0:
        raise ValueError(hostapi.last_error() if hasattr(hostapi, "last_error") else "unpack failed")
    doc, po, fp = np.zeros(n, np.uint32), np.zeros(n + 1, np.uint32), np.zeros(npos.value, np.uint64)
    assert L.rxhost_ft_unpack(data.ctypes.data, data.shape[0], afp, doc.ctypes.data, po.ctypes.data, fp.ctypes.data, None) == n
    return doc, po, fp


@pytest.mark.parametrize("name", ["plain", "arrays"])
def test_packed_posting_decoder_against_reference_packer_golden(name):
    """Byte streams written by the reference's own PackedIdRelVec::insert_back (fixture made by tests/golden/make_golden.py) decode to
    exactly the postings that were packed — both element formats (with / without array indexes) in one stream."""
    z = np.load(FT_GOLDEN)
    afp = int(z[f"packed_{name}_afp"])
    assert (afp < len(z[f"packed_{name}_bytes"])) == (name == "arrays")
    doc, po, fp = _unpack(z[f"packed_{name}_bytes"], afp)
    assert np.array_equal(doc, z[f"packed_{name}_doc"]) and np.array_equal(po, z[f"packed_{name}_pos_off"])
    assert np.array_equal(fp, z[f"packed_{name}_fpos"])
    with pytest.raises(Exception):
        _unpack(z[f"packed_{name}_bytes"][:-1], afp)       # truncated stream: loud error


def test_packed_posting_decoder_against_live_reference_packer():
    from oracle.pyoracle import ref_ft_or_none
    real = ref_ft_or_none(4)
    if real is None:
        pytest.skip("oracle/_ref/libref_ft.so not available")
    rng = np.random.default_rng(12)
    for arr in (False, True):
        for _ in range(3):
            s = make_pos_postings(rng, 300000, 4, int(rng.integers(1, 3000)), 1.0, array_fields=arr, max_pos=1 << 20)
            data, afp = real.pack(s)
            doc, po, fp = _unpack(data, afp)
            assert np.array_equal(doc, s["doc"]) and np.array_equal(po, s["pos_off"]) and np.array_equal(fp, s["fpos"])
    real.close()


@pytest.mark.parametrize("case", MULTI_CASES[:6])
def test_restated_multi_term_merge_equals_reference_golden(ft, case):
    """The same comparison as above against COMMITTED outputs of the real ft::Merger, so it also runs where oracle/_ref is absent."""
    seed, nf, total, limit, ops, arr, fbs = case
    z = np.load(FT_GOLDEN)
    _, words, avg, removed, excluded, terms, _ = _multi_case(seed, nf, total, limit, ops, arr, fbs)
    cfg = ft.default_config(nf, merge_limit=limit)
    gd, gp, gf, gn, _ = ft.merge_query(cfg, terms, total, words, avg, removed, excluded, sort_by_rank=False)
    assert np.array_equal(gd.astype(np.int32), z[f"merge{seed}_doc"]) and np.array_equal(gn, z[f"merge{seed}_norm"])
    assert np.array_equal(gf, z[f"merge{seed}_field"]) and np.array_equal(gp.view(np.uint32), z[f"merge{seed}_proc"].view(np.uint32))


# ------------------------------------------------------------------------------------------- the other Bm25Calculator variants
@pytest.mark.parametrize("bm25_type", ["classic", "word_count"])
def test_restated_merges_with_bm25_classic_and_word_count_equal_real_merger(ft, bm25_type):
    """FTConfig::Bm25Config::bm25Type = classic / wordCount (bm25.h:38-68, dispatch selecterimpl.h:615-624): the restatement against the real
    Merger::Merge<Bm25Classic> / Merge<TermCount>, single-term (mergeSimple) and multi-term (mergeTerm + preselect).  The GPU merger evaluates
    Bm25Rx only and refuses these types (GpuFtMerger::Supports) — this pins the checker for extending it."""
    from oracle.pyoracle import ref_ft_or_none
    nf = 3
    real = ref_ft_or_none(nf)
    if real is None:
        pytest.skip("oracle/_ref/libref_ft.so not available")
    if not hasattr(real.L, "ref_ft_set_bm25_type"):
        pytest.skip("oracle/_ref/libref_ft.so predates the bm25Type switch")
    rng = np.random.default_rng(7 if bm25_type == "classic" else 8)
    total = 2500
    words = rng.integers(1, 40, (total, nf)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    removed = np.zeros(total, np.uint8)
    removed[rng.choice(total, 60, replace=False)] = 1
    real.set_docs(words, avg, removed)
    subs = []
    for wid, proc in enumerate((100.0, 88.0, 61.5)):
        s = make_postings(rng, total, nf, int(rng.integers(200, 900)))
        s["proc"] = proc
        subs.append(s)
        real.set_word_flat(wid, s)
    for limit in (20000, 150):
        cfg = ft.default_config(nf, merge_limit=limit, bm25_type=bm25_type)
        opts = ft.default_opts(nf, field_boost=[1.0, 0.7, 1.3], boost=1.2, term_len_boost=0.9)
        real.set_config(cfg)
        wd, wp, wf, wn = real.merge([dict(op=real.OP_OR, opts=opts, subs=[(i, s["proc"]) for i, s in enumerate(subs)])], None, rank_sort_type=1)
        gd, gp, gf, gn = ft.merge_simple(cfg, opts, total, words, avg, removed, None, subs, sort_by_rank=False)
        assert np.array_equal(gd.astype(np.int32), wd) and np.array_equal(gn, wn) and np.array_equal(gf, wf)
        assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32))
    real.close()
    # multi-term: two of the standing cases, re-run with the other calculator
    for case in (MULTI_CASES[0], MULTI_CASES[3]):
        seed, nf2, total2, limit, ops, arr, fbs = case
        ref_ft, words, avg, removed, excluded, terms, store = _multi_case(seed, nf2, total2, limit, ops, arr, fbs)
        real = ref_ft(nf2)
        real.set_docs(words, avg, removed)
        for s in store:
            real.set_word_fpos(s["word"], s)
        cfg = ft.default_config(nf2, merge_limit=limit, bm25_type=bm25_type)
        real.set_config(cfg)
        rterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
        wd, wp, wf, wn = real.merge(rterms, excluded, rank_sort_type=1)
        gd, gp, gf, gn, _ = ft.merge_query(cfg, terms, total2, words, avg, removed, excluded, sort_by_rank=False)
        assert np.array_equal(gd.astype(np.int32), wd), (bm25_type, seed)
        assert np.array_equal(gn, wn) and np.array_equal(gf, wf) and np.array_equal(gp.view(np.uint32), wp.view(np.uint32))
        real.close()
