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
