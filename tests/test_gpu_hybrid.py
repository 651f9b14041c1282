"""-m gpu: BASELINE configs[4] end to end at test scale — ft_fast BM25 merge (GPU) + cosine KNN (GPU) fused with RRF on the host —
against the same pipeline assembled from the CPU checkers (restated merger + restated brute force + restated fusion)."""
import numpy as np
import pytest

from oracle.pyoracle import FtOracle
from .conftest import make_corpus
from .test_bm25_oracle import make_postings
from .test_hybrid_rerank import restated

pytestmark = pytest.mark.gpu


def test_hybrid_rrf_pipeline(rxgpu, oracle):
    from reindexer_amd import hostapi
    ft = FtOracle(oracle)
    rng = np.random.default_rng(42)
    n_docs, d, k = 6000, 512, 100
    total = n_docs + 1                                   # vdoc 0 is the empty sentinel; vdoc i <-> row id i
    # --- full text side
    words = rng.integers(20, 61, (total, 1)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    subs = []
    m = hostapi.GpuFtMerger(1)
    m.set_docs(words, avg)
    for wid, (proc, cnt) in enumerate(((100.0, 900), (80.0, 400))):
        s = make_postings(rng, total, 1, cnt)
        s["proc"] = proc
        subs.append(s)
        m.set_word_flat(wid, s)
    cfg, opts = hostapi.default_ft_config(1), hostapi.default_ft_opts(1)
    g_id, g_proc, _, _ = m.merge(cfg, opts, [(i, s["proc"]) for i, s in enumerate(subs)], sort_by_rank=True)
    w_id, w_proc, _, _ = ft.merge_simple(cfg, opts, total, words, avg, None, None, subs, sort_by_rank=True)
    assert np.array_equal(np.sort(g_id.astype(np.uint32)), np.sort(w_id))
    # --- vector side (row id = vdoc id; row 0 unused)
    rows = make_corpus(43, total, d)
    labels = np.arange(total, dtype=np.uint64) << np.uint64(32)
    vm = hostapi.GpuBruteforceMap(2, d, total)
    vm.add(rows, labels)
    key = make_corpus(44, 1, d)[0]
    knn_ids, knn_ranks = vm.select(key, k=k, need_sort=False)          # selectRaw: best first, cosine similarity descending
    qn, _ = oracle.normalize_copy(key)
    wd, wl = oracle.bf_search_knn(2, rows, labels, oracle.l2_modules(rows), qn, k)
    assert np.array_equal(knn_ids, (wl >> np.uint64(32)).astype(np.int32))
    # --- fusion: FT ids ascending with their ranks (IndexText::afterSelect hands the id set sorted by id for RRF positions)
    o = np.argsort(g_id, kind="stable")
    ft_ids, ft_ranks = g_id[o].astype(np.int32), g_proc[o]
    ow = np.argsort(w_id, kind="stable")
    for union in (False, True):
        gi, gr = hostapi.merge_ranked("rrf", [60.0], knn_ids, knn_ranks, ft_ids, ft_ranks, union=union, desc=True, metric=2)
        wi, wr = restated("rrf", [60.0], (wl >> np.uint64(32)).astype(np.int32), -wd, w_id[ow].astype(np.int32), w_proc[ow], union, True, 2)
        assert np.array_equal(gi, wi) and np.array_equal(gr.view(np.uint32), wr.view(np.uint32))
        assert (len(gi) > 0) if union else True
    m.close()
    vm.close()


def test_hybrid_resident_wide_k_and_sharded_mirror_take_the_host_pieces(rxgpu, oracle):
    """HybridQueryResident with k + 1 > 128 (the resident KNN list holds 128 entries) or over a sharded mirror: no exception — the query is
    assembled from the host-side pieces (KnnSelectRaw + MergeQuery + MergeRanked), the list a narrow-k resident query gives for the same
    inputs when k fits."""
    from reindexer_amd import hostapi
    from .test_bm25_oracle import _multi_case
    n_docs, d = 5000, 32
    total = n_docs + 1
    _, words, avg, removed, excluded, terms_all, store = _multi_case(17, 1, total, 20000, (1, 1), False, None, sizes=(300, 1500), nsub_range=(1, 3))
    ftm = hostapi.GpuFtMerger(1)
    ftm.set_docs(words, avg, None)
    for s in store:
        ftm.set_word_fpos(s["word"], s)
    rows = make_corpus(18, total, d)
    labels = np.arange(total, dtype=np.uint64) << np.uint64(32)
    cfg = hostapi.default_ft_config(1)
    terms = [dict(op=1, opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms_all]
    key = make_corpus(19, 1, d)[0]
    one = hostapi.GpuBruteforceMap(2, d, total)
    one.add(rows, labels)
    many = hostapi.GpuBruteforceMap(2, d, total, devices=[0, 0])
    many.add(rows, labels)
    narrow = hostapi.hybrid_query_resident(one, ftm, cfg, terms, key, 50, kind="rrf", params=[60.0], union=True, desc=True)
    sharded = hostapi.hybrid_query_resident(many, ftm, cfg, terms, key, 50, kind="rrf", params=[60.0], union=True, desc=True)
    assert np.array_equal(narrow[0], sharded[0]) and np.array_equal(narrow[1].view(np.uint32), sharded[1].view(np.uint32))
    # ... and with the text index over a device list (document-range shards): the FT half is merged over the shards, the fusion runs on the host
    ftmany = hostapi.GpuFtMerger(1, devices=[0, 0])
    ftmany.set_docs(words, avg, None)
    for s in store:
        ftmany.set_word_fpos(s["word"], s)
    for vm in (one, many):
        both = hostapi.hybrid_query_resident(vm, ftmany, cfg, terms, key, 50, kind="rrf", params=[60.0], union=True, desc=True)
        assert np.array_equal(narrow[0], both[0]) and np.array_equal(narrow[1].view(np.uint32), both[1].view(np.uint32))
    ftmany.close()
    for k in (127, 128, 300):
        wide = hostapi.hybrid_query_resident(one, ftm, cfg, terms, key, k, kind="rrf", params=[60.0], union=True, desc=True)
        # the same fusion from the separate product calls
        fid, fproc, _, _, _ = ftm.merge_query(cfg, terms, None, sort_by_rank=True)
        kid, krank = one.select(key, k=k, need_sort=False)
        ids, ranks = hostapi.merge_ranked("rrf", [60.0], kid, krank, fid, fproc, union=True, desc=True, metric=2, ft_order="rank")
        assert np.array_equal(wide[0], ids) and np.array_equal(wide[1].view(np.uint32), ranks.view(np.uint32)), k
    for m in (one, many, ftm):
        m.close()
