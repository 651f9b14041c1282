"""-m gpu: the pre-filtered brute-force search (`WHERE cond AND KNN(...)`) through the C-ABI and through the Map.
Contract: the result is what BruteforceSearch::SearchKnn returns over an index holding ONLY the allowed rows — same distance bits, same
(dist, row) order — so the checker is the ordinary oracle (and the real reference engine where oracle/_ref is present) run on the sub-corpus."""
import numpy as np
import pytest

from .conftest import lex_topk, make_corpus

pytestmark = pytest.mark.gpu


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def to_words(ids, n):
    words = np.zeros((n + 31) // 32, np.uint32)
    ids = np.asarray(ids, np.int64)
    np.bitwise_or.at(words, ids >> 5, (np.uint32(1) << (ids & 31).astype(np.uint32)))
    return words


def want_subset(oracle, metric, q, rows, inv, ids, kk):
    sub_inv = inv[ids] if inv is not None else None
    d = oracle.dist_many(metric, q, rows[ids], sub_inv)
    wd, wpos = lex_topk(d, kk)
    return wd, ids[wpos].astype(np.uint32)


def subsets(rng, n):
    yield "one_first", np.array([0], np.uint32)
    yield "one_last", np.array([n - 1], np.uint32)
    yield "three", np.array([1, n // 2, n - 2], np.uint32)
    for dens in (0.002, 0.05, 0.5):
        yield f"dens{dens}", np.flatnonzero(rng.random(n) < dens).astype(np.uint32)
    yield "all", np.arange(n, dtype=np.uint32)
    yield "block", np.arange(n // 3, n // 3 + 777, dtype=np.uint32)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("d", [5, 64, 100, 128, 256, 512, 768, 1024])
def test_subset_and_bitmap_match_oracle(rxgpu, oracle, metric, d):
    n = 6007
    rng = np.random.default_rng(1000 * metric + d)
    rows = make_corpus(d, n, d)
    inv = oracle.l2_modules(rows) if metric == 2 else None
    q = make_corpus(77 + d, 1, d)[0]
    if metric == 2:
        q, _ = oracle.normalize_copy(q)
    with rxgpu.VectorIndex(metric, d, n) as ix:
        ix.upload_rows(0, rows, inv)
        for name, ids in subsets(rng, n):
            for kk in (1, 11, 64, 100, 128, 300):
                eff = min(kk, ids.size)
                wd, wr = want_subset(oracle, metric, q, rows, inv, ids, eff)
                dist, row, cnt = ix.search_knn_subset(q, kk, ids)
                assert int(cnt[0]) == eff, (name, kk)
                assert np.array_equal(row[0, :eff], wr), (name, kk)
                assert np.array_equal(bits(dist[0, :eff]), bits(wd)), (name, kk)
                dist, row, cnt, allowed = ix.search_knn_bitmap(q, kk, to_words(ids, n))
                assert allowed == ids.size and int(cnt[0]) == eff, (name, kk)
                assert np.array_equal(row[0, :eff], wr), (name, kk)
                assert np.array_equal(bits(dist[0, :eff]), bits(wd)), (name, kk)


@pytest.mark.parametrize("metric,d,kks", [(1, 768, (11, 100)), (0, 128, (11, 64)), (2, 512, (10, 101)), (0, 1024, (7,)), (1, 256, (33,))])
def test_long_lists_take_the_chunked_kernel(rxgpu, oracle, metric, d, kks):
    """Lists of >= 2 * (waves on the chip) * 64 entries go through knn_scan_subset (64-entry chunks, double-buffered rows); the list length
    is ragged (not a multiple of 64 or 4) and several queries share the launch."""
    n = 400_000 if d <= 256 else 330_000
    rng = np.random.default_rng(d)
    rows = make_corpus(4242 + d, n, d)
    inv = oracle.l2_modules(rows) if metric == 2 else None
    queries = make_corpus(9 + d, 3, d)
    if metric == 2:
        queries = np.stack([oracle.normalize_copy(q)[0] for q in queries])
    keep = rng.random(n) < 0.9
    keep[-1] = True
    ids = np.flatnonzero(keep).astype(np.uint32)
    if ids.size % 64 == 0:
        ids = ids[:-1]
    assert ids.size >= 2 * 2048 * 64
    with rxgpu.VectorIndex(metric, d, n) as ix:
        ix.upload_rows(0, rows, inv)
        for kk in kks:
            dist, row, cnt = ix.search_knn_subset(queries, kk, ids)
            bdist, brow, bcnt, allowed = ix.search_knn_bitmap(queries, kk, to_words(ids, n))
            assert allowed == ids.size
            for qi in range(queries.shape[0]):
                wd, wr = want_subset(oracle, metric, queries[qi], rows, inv, ids, kk)
                assert int(cnt[qi]) == kk and int(bcnt[qi]) == kk
                assert np.array_equal(row[qi], wr), (kk, qi)
                assert np.array_equal(bits(dist[qi]), bits(wd)), (kk, qi)
                assert np.array_equal(brow[qi], wr) and np.array_equal(bits(bdist[qi]), bits(wd)), (kk, qi)
        # the full list == the unfiltered search
        dist, row, cnt = ix.search_knn_subset(queries[0], 11, np.arange(n, dtype=np.uint32))
        fd, fr, _ = ix.search_knn(queries[0], 11)
        assert np.array_equal(row, fr) and np.array_equal(bits(dist), bits(fd))


def test_subset_ties_follow_dist_row_order(rxgpu, oracle):
    rng = np.random.default_rng(5)
    n, d = 300_000, 8
    rows = rng.integers(-1, 2, (n, d)).astype(np.float32)
    ids = np.flatnonzero(rng.random(n) < 0.95).astype(np.uint32)
    small = ids[:: 41]
    with rxgpu.VectorIndex("l2", d, n) as ix:
        ix.upload_rows(0, rows)
        for qi in range(4):
            q = rng.integers(-1, 2, d).astype(np.float32)
            for lst in (ids, small):
                for kk in (1, 10, 64, 128, 500):
                    wd, wr = want_subset(oracle, 0, q, rows, None, lst, kk)
                    dist, row, cnt = ix.search_knn_subset(q, kk, lst)
                    assert np.array_equal(row[0], wr), (qi, kk, lst.size)
                    assert np.array_equal(bits(dist[0]), bits(wd))


def test_subset_edge_cases_and_errors(rxgpu, oracle):
    n, d = 1000, 64
    rows = make_corpus(3, n, d)
    q = make_corpus(4, 1, d)[0]
    with rxgpu.VectorIndex("ip", d, n) as ix:
        ix.upload_rows(0, rows)
        dist, row, cnt = ix.search_knn_subset(q, 10, np.empty(0, np.uint32))
        assert int(cnt[0]) == 0
        dist, row, cnt, allowed = ix.search_knn_bitmap(q, 10, np.zeros((n + 31) // 32, np.uint32))
        assert int(cnt[0]) == 0 and allowed == 0
        # bits at and above the row count are ignored
        words = np.full((n + 31) // 32 + 2, 0xFFFFFFFF, np.uint32)
        dist, row, cnt, allowed = ix.search_knn_bitmap(q, 10, words)
        fd, fr, _ = ix.search_knn(q, 10)
        assert allowed == n and np.array_equal(row, fr) and np.array_equal(bits(dist), bits(fd))
        with pytest.raises(rxgpu.RxGpuError):
            ix.search_knn_subset(q, 10, np.array([5, 5], np.uint32))        # not strictly increasing
        with pytest.raises(rxgpu.RxGpuError):
            ix.search_knn_subset(q, 10, np.array([7, 3], np.uint32))
        with pytest.raises(rxgpu.RxGpuError):
            ix.search_knn_subset(q, 10, np.array([3, n], np.uint32))        # beyond the index
        with pytest.raises(rxgpu.RxGpuError):
            ix.search_knn_bitmap(q, 10, np.zeros(n // 32 - 1, np.uint32))   # does not cover every row
        # after a truncate the bitmap is interpreted against the new count
        ix.truncate(500)
        dist, row, cnt, allowed = ix.search_knn_bitmap(q, 600, words)
        assert allowed == 500 and int(cnt[0]) == 500 and int(row[0, :500].max()) == 499


def test_subset_device_variant(rxgpu, oracle):
    import torch
    n, d, kk = 50_000, 128, 20
    rows = make_corpus(8, n, d)
    queries = make_corpus(9, 4, d)
    rng = np.random.default_rng(0)
    ids = np.flatnonzero(rng.random(n) < 0.3).astype(np.uint32)
    dev = torch.device("cuda", 0)
    with rxgpu.VectorIndex("l2", d, n) as ix:
        ix.upload_rows(0, rows)
        tq = torch.from_numpy(queries).to(dev)
        tids = torch.from_numpy(ids.astype(np.int32)).to(dev)   # same bits as uint32
        od = torch.empty((4, kk), dtype=torch.float32, device=dev)
        orow = torch.empty((4, kk), dtype=torch.int32, device=dev)
        ocnt = torch.empty(4, dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream(dev)
        assert ix.check_row_list_device(tids.data_ptr(), ids.size, stream.cuda_stream)
        ix.search_knn_subset_device(tq.data_ptr(), 4, kk, tids.data_ptr(), ids.size, od.data_ptr(), orow.data_ptr(), ocnt.data_ptr(),
                                    stream.cuda_stream)
        torch.cuda.synchronize(dev)
        for qi in range(4):
            wd, wr = want_subset(oracle, 0, queries[qi], rows, None, ids, kk)
            assert np.array_equal(orow[qi].cpu().numpy().view(np.uint32), wr)
            assert np.array_equal(bits(od[qi].cpu().numpy()), bits(wd))
            assert int(ocnt[qi]) == kk
        bad = tids.clone()
        bad[10] = bad[9]
        assert not ix.check_row_list_device(bad.data_ptr(), ids.size, stream.cuda_stream)
        bad = tids.clone()
        bad[-1] = n
        assert not ix.check_row_list_device(bad.data_ptr(), ids.size, stream.cuda_stream)


# ------------------------------------------------------------------------------------------------ through the Map
@pytest.fixture(scope="module")
def hostapi(rxgpu):
    from reindexer_amd import hostapi as h
    h.lib()
    return h


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("kind", ["gauss", "ties"])
def test_map_filtered_equals_search_over_the_sub_index(hostapi, oracle, metric, kind):
    """GpuBruteforceMap::SearchKnnFiltered == BruteforceSearch::SearchKnn over an index holding only the allowed points (same relative
    insertion order), incl. swap-deletes, exact ties across the k-th boundary, unknown and duplicate labels, dense and sparse filters."""
    rng = np.random.default_rng(23 + metric)
    n, d = (3000, 128) if kind == "gauss" else (2000, 8)
    rows = make_corpus(6, n, d) if kind == "gauss" else rng.integers(-1, 2, (n, d)).astype(np.float32)
    labels = (rng.permutation(n).astype(np.uint64) << np.uint64(32)) | rng.integers(0, 3, n).astype(np.uint64)
    m = hostapi.GpuBruteforceMap(metric, d, n)
    m.add(rows, labels)
    live_rows, live_labels = rows.copy(), labels.copy()
    cnt = n
    for lab in labels[rng.choice(n, 50, replace=False)]:
        m.remove(lab)
        pos = int(np.nonzero(live_labels[:cnt] == lab)[0][0])
        if pos + 1 != cnt:
            live_rows[pos] = live_rows[cnt - 1]
            live_labels[pos] = live_labels[cnt - 1]
        cnt -= 1
    live_rows, live_labels = live_rows[:cnt].copy(), live_labels[:cnt].copy()
    for dens in (0.01, 0.2, 0.9):
        keep = rng.random(cnt) < dens
        sub_rows, sub_labels = live_rows[keep], live_labels[keep]
        sub_inv = oracle.l2_modules(sub_rows) if metric == 2 else None
        allowed = np.concatenate([sub_labels, sub_labels[:5], np.array([2**63 + 5, 12345], np.uint64)])   # duplicates + unknown labels
        rng.shuffle(allowed)
        for qi in range(8):
            q = make_corpus(200 + qi, 1, d)[0] if kind == "gauss" else rng.integers(-1, 2, d).astype(np.float32)
            if metric == 2:
                q, _ = oracle.normalize_copy(q)
            for k in (1, 10, 64, 100, 200, sub_rows.shape[0] + 3):
                wd, wl = oracle.bf_search_knn(metric, sub_rows, sub_labels, sub_inv, q, k)
                gd, gl = m.search_knn_filtered(q, k, allowed)
                assert np.array_equal(gl, wl), (kind, metric, dens, qi, k)
                assert np.array_equal(bits(gd), bits(wd))
    gd, gl = m.search_knn_filtered(q, 5, np.array([7], np.uint64))   # nothing allowed is present
    assert gl.size == 0
    if kind == "ties":
        assert m.tie_replays > 0
    m.close()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_map_filtered_equals_the_real_reference_engine(hostapi, ref, metric):
    """The same statement against the reference's own BruteforceSearch (oracle/_ref) built over the allowed points only."""
    from oracle.pyoracle import RefBruteforce
    rng = np.random.default_rng(31)
    n, d = 5000, 256
    rows = make_corpus(11, n, d)
    labels = (np.arange(n, dtype=np.uint64) << np.uint64(32))
    m = hostapi.GpuBruteforceMap(metric, d, n)
    m.add(rows, labels)
    for dens in (0.03, 0.6):
        keep = rng.random(n) < dens
        rb = RefBruteforce(ref, metric, d, int(keep.sum()))
        rb.add(rows[keep], labels[keep])
        for qi in range(6):
            q = make_corpus(300 + qi, 1, d)[0]
            if metric == 2:
                q, _ = hostapi.normalize_copy(q)
            for k in (1, 10, 100):
                wd, wl = rb.search_knn(q, k)
                gd, gl = m.search_knn_filtered(q, k, labels[keep])
                assert np.array_equal(gl, wl), (metric, dens, qi, k)
                assert np.array_equal(bits(gd), bits(wd))
        rb.close()
    m.close()
