"""-m gpu: the hybrid rank fusion ON THE DEVICE (hybrid_fuse.hip) against
  * the Python restatement of MergerRankedImpl / mergeRanked / InitRRFPositions that tests/test_hybrid_rerank.py pins to the reference's own
    code (oracle/_ref/libref_rank.so: selectiteratorcontainer.cc compiled in place), and that library itself where it is present;
  * the host fusion (hybrid_rerank.h) on the same inputs;
and the fully resident hybrid query (FT merge left in HBM -> postProcessResults + fusion on the device with a KNN list that lies in HBM)
against the pipeline assembled from the separate product calls (GpuFtMerger::MergeQuery + search + host fusion), which the other suites
hold to the reference.  Bar: identical ids in identical order, identical rank bits."""
import numpy as np
import pytest

from .conftest import make_corpus
from .test_bm25_oracle import make_pos_postings
from .test_hybrid_rerank import _positions_by_id, restated

pytestmark = pytest.mark.gpu

KINDS = [("rrf", [60.0]), ("rrf", [1.0]), ("linear", [0.7, 0.1, 0.3, 5.0, 2.0]), ("linear", [1.0, 0.0, -1.0, 0.0, 0.0]),
         ("linear", [0.5, 0.0, 0.0, 3.0, 1.0])]   # kFt < 0: classes reverse; kFt = 0: every class one group (pure id order)


def _same(got, want, tag):
    assert np.array_equal(got[0], want[0]), (tag, len(got[0]), len(want[0]))
    assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32)), tag


@pytest.mark.parametrize("kind,params", KINDS)
@pytest.mark.parametrize("union", [False, True])
@pytest.mark.parametrize("metric", [0, 1])
def test_device_fusion_equals_restated_merger_small(rxgpu, kind, params, union, metric):
    """Small id space, massive rank ties (shared RRF positions on both sides), duplicate row ids in the KNN list (array field), empty sides."""
    from reindexer_amd import capi
    rng = np.random.default_rng(abs(hash((kind, len(params), union, metric))) % 10007)
    for it in range(30):
        nk, nf = int(rng.integers(0, 60)), int(rng.integers(0, 300))
        if it == 0:
            nk = 0
        if it == 1:
            nf = 0
        knn_ids = rng.choice(500, nk, replace=False).astype(np.int32)
        if it % 3 == 0 and nk > 4:
            knn_ids[rng.integers(0, nk, 3)] = knn_ids[rng.integers(0, nk, 3)]
        kr = np.sort(rng.integers(0, 12, nk).astype(np.float32))
        knn_ranks = kr if metric == 0 else kr[::-1].copy()
        ft_ids = rng.permutation(np.sort(rng.choice(500, nf, replace=False))).astype(np.int32)    # the merge order is not the id order
        ft_ranks = rng.integers(0, 20 if it % 2 else 256, nf).astype(np.uint8)
        o = np.argsort(ft_ids, kind="stable")
        for desc in (True, False):
            want = restated(kind, params, knn_ids, knn_ranks, ft_ids[o], ft_ranks[o].astype(np.float32), union, desc, metric)
            got = capi.hybrid_fuse(kind, params, knn_ids, knn_ranks, ft_ids, ft_ranks, union=union, desc=desc, metric=metric)
            _same(got, want, (it, desc))


@pytest.mark.parametrize("kind,params", [("rrf", [60.0]), ("linear", [0.7, 0.1, -0.3, 5.0, 2.0]), ("linear", [1.0, 0.0, 1.0, 0.0, 0.0])])
@pytest.mark.parametrize("union", [False, True])
def test_device_fusion_at_hybrid_sizes(rxgpu, kind, params, union):
    """configs[4] sizes: up to 65 000 FT hits with uint8 ranks, ids up to 2^31, k up to 1024 — every radix pass count (1..4 id bytes), the
    head larger than a wavefront, head entries whose fused rank equals a tail group's (position inside the group by id)."""
    from reindexer_amd import capi, hostapi
    rng = np.random.default_rng(11)
    for nf, id_space, nk in ((1, 200, 3), (3000, 250, 64), (3000, 60_000, 100), (20000, 5_000_000, 100), (20000, 5_000_000, 1024),
                             (65000, 2_000_000_000, 300)):
        id_space = max(id_space, nf + nk + 10)
        ft_ids = rng.choice(id_space, nf, replace=False).astype(np.int32)
        ft_ranks = rng.integers(0, 256, nf).astype(np.uint8)
        knn_ids = rng.choice(id_space, nk, replace=False).astype(np.int32)
        take = min(nk * 2 // 5, nf)
        knn_ids[:take] = ft_ids[rng.choice(nf, take, replace=False)]
        knn_ids = np.unique(knn_ids)
        rng.shuffle(knn_ids)
        knn_ranks = np.sort(rng.integers(0, 200, knn_ids.size).astype(np.float32) / np.float32(64))[::-1].copy()
        o = np.argsort(ft_ids, kind="stable")
        for desc in (True, False):
            want = hostapi.merge_ranked(kind, params, knn_ids, knn_ranks, ft_ids[o], ft_ranks[o].astype(np.float32), union=union, desc=desc, metric=1)
            got = capi.hybrid_fuse(kind, params, knn_ids, knn_ranks, ft_ids, ft_ranks, union=union, desc=desc, metric=1)
            _same(got, want, (nf, nk, desc))


def test_device_fusion_equals_reference_merger(rxgpu):
    """Directly against SelectIteratorContainer::MergerRankedImpl (libref_rank.so travels to the GPU box)."""
    from oracle.pyoracle import ref_rank_or_none
    from reindexer_amd import capi
    refrank = ref_rank_or_none()
    if refrank is None:
        pytest.skip("oracle/_ref/libref_rank.so not available")
    rng = np.random.default_rng(5)
    for kind, params in KINDS[:3]:
        for metric in (0, 2):
            for union in (False, True):
                nf, nk = 4000, 100
                ft_ids = rng.choice(300_000, nf, replace=False).astype(np.int32)
                ft_ranks = rng.integers(1, 256, nf).astype(np.uint8)
                knn_ids = np.concatenate([ft_ids[rng.choice(nf, 40, replace=False)], rng.choice(300_000, 60, replace=False).astype(np.int32)])
                knn_ids[97] = knn_ids[3]   # the same row twice
                rng.shuffle(knn_ids)
                kr = np.sort(rng.integers(0, 30, nk).astype(np.float32) / np.float32(8))
                knn_ranks = kr if metric == 0 else kr[::-1].copy()
                o = np.argsort(ft_ids, kind="stable")
                fr = ft_ranks[o].astype(np.float32)
                for desc in (True, False):
                    want = refrank.merge(kind, params, knn_ids, knn_ranks, ft_ids[o], fr, union=union, desc=desc, metric=metric, ft_positions=_positions_by_id(fr))
                    got = capi.hybrid_fuse(kind, params, knn_ids, knn_ranks, ft_ids, ft_ranks, union=union, desc=desc, metric=metric)
                    _same(got, want, (kind, metric, union, desc))


@pytest.mark.parametrize("metric", [0, 2])
def test_resident_hybrid_query_equals_assembled_pipeline(rxgpu, oracle, metric):
    """The whole hybrid query with nothing leaving HBM in between: GpuFtMerger::MergeQueryResident + rxgpu_search_knn_device +
    rxgpu_hybrid_fuse_resident, vs MergeQuery (host postProcessResults) + the same KNN list + the host fusion.  min_rank high enough to
    drop documents, ranks above 255 (the 255 / max scaling), single-term and multi-term queries, a vdoc -> row id table."""
    import torch
    from reindexer_amd import capi, hostapi
    rng = np.random.default_rng(3 + metric)
    n_docs, d, k = 9000, 128, 100
    total = n_docs + 1
    nf = 2
    words = rng.integers(3, 30, (total, nf)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg)
    store = []
    for w, (proc, cnt) in enumerate(((100.0, 2500), (85.0, 1200), (100.0, 1800), (70.0, 600))):
        s = make_pos_postings(rng, total, nf, cnt, proc)
        s["word"] = w
        m.set_word_fpos(w, s)
        store.append(s)
    dev = torch.device("cuda", 0)
    rows = make_corpus(43, total, d)
    d_rows = torch.from_numpy(rows).to(dev)
    inv = oracle.l2_modules(rows) if metric == 2 else None
    d_inv = torch.from_numpy(inv).to(dev) if inv is not None else None
    ix = capi.VectorIndex(metric, d, device=0)
    ix.adopt_device_rows(d_rows.data_ptr(), total, d, d_inv.data_ptr() if d_inv is not None else None, keepalive=(d_rows, d_inv))
    row_of_doc = rng.permutation(total).astype(np.int32)          # texts and vectors of a row are numbered differently
    d_map = torch.from_numpy(row_of_doc).to(dev)
    opts = hostapi.default_ft_opts(nf)
    queries = {
        "simple": [dict(op=1, opts=opts, subs=[(0, 100.0), (1, 85.0)])],
        "or_or": [dict(op=1, opts=opts, subs=[(0, 100.0), (1, 85.0)]), dict(op=1, opts=opts, subs=[(2, 100.0), (3, 70.0)])],
        "and": [dict(op=1, opts=opts, subs=[(0, 100.0)]), dict(op=2, opts=opts, subs=[(2, 100.0)])],
    }
    for qname, terms in queries.items():
        for min_rank, limit in ((5, 20000), (90, 20000), (5, 700)):
            cfg = hostapi.default_ft_config(nf, min_rank=min_rank, merge_limit=limit)
            key = make_corpus(50 + len(qname), 1, d)[0]
            if metric == 2:
                key, _ = oracle.normalize_copy(key)
            for kind, params in (("rrf", [60.0]), ("linear", [0.7, 0.1, 0.3, 5.0, 2.0])):
                for union in (True, False):
                    p_dist, p_row, p_cnt, p_stream, entries = ix.search_knn_resident(key, k + 1)      # enqueued, left in HBM
                    assert entries == k + 1
                    gi, gr, tie = m.hybrid_query(cfg, terms, p_dist, p_row, entries, k, metric, kind=kind, params=params, union=union,
                                                 knn_count_ptr=p_cnt, knn_stream=p_stream, row_of_doc_ptr=d_map.data_ptr())
                    assert not tie
                    # the assembled pipeline
                    fid, fproc, _, fnorm, _ = m.merge_query(cfg, terms, sort_by_rank=False)
                    hd, hr, _ = ix.search_knn(key[None, :], k + 1)
                    kd, kr_ = hd[0, :k], hr[0, :k].astype(np.int32)
                    ranks = kd if metric == 0 else -kd
                    ft_rows = row_of_doc[fid]
                    o = np.argsort(ft_rows, kind="stable")
                    wi, wr = hostapi.merge_ranked(kind, params, kr_, ranks.astype(np.float32), ft_rows[o], fnorm[o].astype(np.float32), union=union, desc=True,
                                                  metric=metric)
                    _same((gi, gr), (wi, wr), (qname, min_rank, limit, kind, union))
    m.close()
    ix.close()


def test_resident_fusion_reports_boundary_ties_and_empty_ft(rxgpu, oracle):
    import torch
    from reindexer_amd import capi, hostapi
    dev = torch.device("cuda", 0)
    total, d, k = 600, 32, 10
    rows = make_corpus(9, total, d)
    rows[100] = rows[7]                    # two identical rows: with the query at that point the k-th / (k+1)-th distances can tie
    d_rows = torch.from_numpy(rows).to(dev)
    ix = capi.VectorIndex(0, d, device=0)
    ix.adopt_device_rows(d_rows.data_ptr(), total, d, None, keepalive=(d_rows,))
    m = hostapi.GpuFtMerger(1)
    words = np.full((total, 1), 5, np.float32)
    words[0] = 0
    m.set_docs(words, words[1:].mean(axis=0).astype(np.float32))
    rng = np.random.default_rng(1)
    s = make_pos_postings(rng, total, 1, 50, 100.0)
    m.set_word_fpos(0, s)
    cfg, opts = hostapi.default_ft_config(1), hostapi.default_ft_opts(1)
    stream = torch.cuda.current_stream(dev).cuda_stream
    d_key = torch.from_numpy(rows[7].copy()).to(dev)
    od = torch.empty(2, dtype=torch.float32, device=dev)
    orow = torch.empty(2, dtype=torch.int32, device=dev)
    ix.search_knn_device(d_key.data_ptr(), 1, 2, od.data_ptr(), orow.data_ptr(), None, stream)
    _, _, tie = m.hybrid_query(cfg, [dict(op=1, opts=opts, subs=[(0, 100.0)])], od.data_ptr(), orow.data_ptr(), 2, 1, 0, knn_stream=stream)
    assert tie                              # dist[0] == dist[1] == 0: the Map's label replay has to pick the k-th
    # an FT side that merges nothing (a NOT-only query): the fused list is the KNN list under RRF
    gi, gr, tie = m.hybrid_query(cfg, [dict(op=3, opts=opts, subs=[(0, 100.0)])], od.data_ptr(), orow.data_ptr(), 2, 2, 0, knn_stream=stream)
    torch.cuda.synchronize(dev)
    want = restated("rrf", [60.0], orow.cpu().numpy().astype(np.int32), od.cpu().numpy(), np.zeros(0, np.int32), np.zeros(0, np.float32), True, True, 0)
    _same((gi, gr), want, "empty ft")
    m.close()
    ix.close()


@pytest.mark.parametrize("metric", [1, 2])
def test_hybrid_query_through_map_and_merger(rxgpu, oracle, metric):
    """rxgpu::host::HybridQueryResident (hybrid_query.h): the Map's resident search + the Merger's resident merge + the fusion, vs the same
    query assembled from Map::select + MergeQuery + the host fusion — with swap-deletes in between (the device needs the row-id table then:
    labels are no longer row << 32) and a key placed ON a duplicated row (boundary tie -> the label-aware replay path)."""
    from reindexer_amd import hostapi
    rng = np.random.default_rng(17 + metric)
    n_docs, d, k = 5000, 96, 100
    total = n_docs + 1
    words = rng.integers(10, 40, (total, 1)).astype(np.float32)
    words[0] = 0
    ftm = hostapi.GpuFtMerger(1)
    ftm.set_docs(words, words[1:].mean(axis=0).astype(np.float32))
    for w, (proc, cnt) in enumerate(((100.0, 1500), (75.0, 700), (100.0, 400))):
        s = make_pos_postings(rng, total, 1, cnt, proc)
        ftm.set_word_fpos(w, s)
    rows = make_corpus(61, total, d)
    labels = np.arange(total, dtype=np.uint64) << np.uint64(32)
    vm = hostapi.GpuBruteforceMap(metric, d, total)
    vm.add(rows, labels)
    opts = hostapi.default_ft_opts(1)
    cfg = hostapi.default_ft_config(1)
    terms = [dict(op=1, opts=opts, subs=[(0, 100.0), (1, 75.0)]), dict(op=1, opts=opts, subs=[(2, 100.0)])]

    def check(tag, key, expect_tie=False):
        for kind, params in (("rrf", [60.0]), ("linear", [0.7, 0.1, 0.3, 5.0, 2.0])):
            for union in (True, False):
                gi, gr, tie = hostapi.hybrid_query_resident(vm, ftm, cfg, terms, key, k, kind=kind, params=params, union=union)
                assert tie == expect_tie, (tag, tie)
                kid, krank = vm.select(key, k=k, need_sort=False)
                fid, _, _, fnorm, _ = ftm.merge_query(cfg, terms, sort_by_rank=False)
                o = np.argsort(fid, kind="stable")
                wi, wr = hostapi.merge_ranked(kind, params, kid, krank, fid[o], fnorm[o].astype(np.float32), union=union, desc=True, metric=metric)
                _same((gi, gr), (wi, wr), (tag, kind, union))

    check("identity labels", make_corpus(70, 1, d)[0])
    for victim in rng.choice(np.arange(1, total), 40, replace=False):      # swap-with-last deletes: rows move, labels stay with them
        vm.remove(labels[victim])
    check("after swap-deletes", make_corpus(71, 1, d)[0])
    # the same vector stored under two labels, the key ON it, k = 1: the 1st and 2nd distances are equal -> the device reports the
    # boundary tie and the query is redone through the label-aware replay (the larger label is evicted, bruteforce.cc:103-127)
    dup = make_corpus(72, 1, d)
    vm.add(dup, np.array([(total + 5) << 32], np.uint64))
    vm.add(dup, np.array([(total + 3) << 32], np.uint64))
    k = 1
    check("boundary tie", dup[0], expect_tie=True)
    ftm.close()
    vm.close()
