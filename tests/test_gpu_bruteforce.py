"""-m gpu: parity of the HIP brute-force path (through the C-ABI) with the CPU oracle on seeded inputs.
Bar: bit-exact distances and identical rows (integer/index work), for every metric and dimension class."""
import numpy as np
import pytest

from .conftest import lex_topk, make_corpus

pytestmark = pytest.mark.gpu


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def oracle_all(oracle, metric, q, rows, inv):
    return oracle.dist_many(metric, q, rows, inv)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("d", [1, 5, 16, 33, 64, 100, 128, 130, 256, 384, 512, 768, 1000, 1024, 1536])
def test_topk_matches_oracle_all_dims(rxgpu, oracle, metric, d):
    n = 3001  # not a multiple of 4: exercises the ragged last quad
    rows = make_corpus(d, n, d)
    inv = oracle.l2_modules(rows) if metric == 2 else None
    queries = make_corpus(1000 + d, 3, d)
    with rxgpu.VectorIndex(metric, d, n) as ix:
        ix.upload_rows(0, rows, inv)
        for qi in range(queries.shape[0]):
            q = queries[qi]
            if metric == 2:
                q, _ = oracle.normalize_copy(q)
            want_all = oracle_all(oracle, metric, q, rows, inv)
            for kk in (1, 11, 64):
                dist, row, cnt = ix.search_knn(q, kk)
                wd, wr = lex_topk(want_all, kk)
                assert int(cnt[0]) == kk
                assert np.array_equal(row[0], wr), (metric, d, kk)
                assert np.array_equal(bits(dist[0]), bits(wd))


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_config1_100k_x_128(rxgpu, oracle, metric):
    """BASELINE.json configs[0]: 100k x 128, k = 10, single queries — ids and distance bits vs the oracle scan."""
    n, d = 100_000, 128
    rows = make_corpus(20260924, n, d)
    inv = oracle.l2_modules(rows) if metric == 2 else None
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    queries = make_corpus(7, 40, d)
    with rxgpu.VectorIndex(metric, d, n) as ix:
        ix.upload_rows(0, rows, inv)
        qs = np.stack([oracle.normalize_copy(q)[0] if metric == 2 else q for q in queries])
        dist, row, cnt = ix.search_knn(qs, 11)  # batched call: nq = 40
        for qi in range(qs.shape[0]):
            wd, wl = oracle.bf_search_knn(metric, rows, labels, inv, qs[qi], 10)
            assert np.array_equal(labels[row[qi, :10]], wl)
            assert np.array_equal(bits(dist[qi, :10]), bits(wd))


def test_ties_follow_dist_row_order(rxgpu, oracle):
    """Massive exact ties: the C-ABI contract is the (dist,row) total order."""
    rng = np.random.default_rng(3)
    n, d = 5000, 8
    rows = rng.integers(-1, 2, (n, d)).astype(np.float32)
    with rxgpu.VectorIndex("l2", d, n) as ix:
        ix.upload_rows(0, rows)
        for qi in range(10):
            q = rng.integers(-1, 2, d).astype(np.float32)
            want_all = oracle.dist_many(0, q, rows)
            for kk in (1, 10, 64, 200, 1000):
                dist, row, cnt = ix.search_knn(q, kk)
                wd, wr = lex_topk(want_all, kk)
                assert np.array_equal(row[0], wr), (qi, kk)
                assert np.array_equal(bits(dist[0]), bits(wd))


def test_duplicate_rows_and_zero_vectors(rxgpu, oracle):
    n, d = 2000, 64
    rows = make_corpus(1, n, d)
    rows[100:200] = rows[0]       # 100 duplicates of row 0
    rows[500:600] = 0.0           # zero vectors (cosine norm coefficient = 1)
    for metric in (0, 1, 2):
        inv = oracle.l2_modules(rows) if metric == 2 else None
        with rxgpu.VectorIndex(metric, d, n) as ix:
            ix.upload_rows(0, rows, inv)
            q = rows[0].copy()
            if metric == 2:
                q, _ = oracle.normalize_copy(q)
            want_all = oracle.dist_many(metric, q, rows, inv)
            dist, row, cnt = ix.search_knn(q, 64)
            wd, wr = lex_topk(want_all, 64)
            assert np.array_equal(row[0], wr)
            assert np.array_equal(bits(dist[0]), bits(wd))


def test_large_k_radix_select_path(rxgpu, oracle):
    """kk > 64 takes the distance-pass + radix-select path (the reference bench uses k = 1000)."""
    n, d = 20_000, 32
    rows = make_corpus(2, n, d)
    q = make_corpus(3, 1, d)[0]
    for metric in (0, 1):
        want_all = oracle.dist_many(metric, q, rows)
        with rxgpu.VectorIndex(metric, d, n) as ix:
            ix.upload_rows(0, rows)
            for kk in (65, 1000, 1001, n, n + 10):
                dist, row, cnt = ix.search_knn(q, kk)
                c = int(cnt[0])
                assert c == min(kk, n)
                wd, wr = lex_topk(want_all, c)
                assert np.array_equal(row[0, :c], wr)
                assert np.array_equal(bits(dist[0, :c]), bits(wd))


def test_small_and_empty_indexes(rxgpu, oracle):
    d = 128
    rows = make_corpus(4, 7, d)
    q = make_corpus(5, 1, d)[0]
    with rxgpu.VectorIndex("l2", d, 16) as ix:
        dist, row, cnt = ix.search_knn(q, 10)       # empty index (bruteforce.cc:106-108)
        assert int(cnt[0]) == 0
        rd, rr = ix.search_range(q, 100.0)
        assert rd.size == 0
        for n in range(1, 8):
            ix.upload_rows(n - 1, rows[n - 1: n])
            want_all = oracle.dist_many(0, q, rows[:n])
            dist, row, cnt = ix.search_knn(q, 10)   # k > N: k = min(k, N) (bruteforce.cc:111)
            assert int(cnt[0]) == n
            wd, wr = lex_topk(want_all, n)
            assert np.array_equal(row[0, :n], wr) and np.array_equal(bits(dist[0, :n]), bits(wd))
        dist, row, cnt = ix.search_knn(q, 0)        # k == 0
        assert int(cnt[0]) == 0


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_range_matches_oracle(rxgpu, oracle, metric):
    n, d = 6000, 100
    rows = make_corpus(6, n, d)
    inv = oracle.l2_modules(rows) if metric == 2 else None
    labels = np.arange(n, dtype=np.uint64)
    with rxgpu.VectorIndex(metric, d, n) as ix:
        ix.upload_rows(0, rows, inv)
        for qi in range(5):
            q = make_corpus(50 + qi, 1, d)[0]
            if metric == 2:
                q, _ = oracle.normalize_copy(q)
            all_d = np.sort(oracle.dist_many(metric, q, rows, inv))
            for radius in (float(all_d[0]), float(all_d[37]), float(all_d[-1]) + 1.0, float(all_d[0]) - 1.0):
                wd, wl = oracle.bf_search_range(metric, rows, labels, inv, q, radius)
                gd, gr = ix.search_range(q, radius, cap=64)     # small cap: exercises the OVERFLOW retry
                assert np.array_equal(gr.astype(np.uint64), wl)
                assert np.array_equal(bits(gd), bits(wd))
            gd, gr = ix.search_range(q, float(all_d[37]), inclusive=True)
            assert gd.size == int((all_d <= all_d[37]).sum())


def test_swap_delete_mirror(rxgpu, oracle):
    """move_row + truncate mirror RemovePoint's swap-with-last (bruteforce.cc:70-86)."""
    n, d = 1000, 64
    rows = make_corpus(8, n, d)
    live = rows.copy()
    cnt = n
    rng = np.random.default_rng(1)
    with rxgpu.VectorIndex("cosine", d, n) as ix:
        ix.upload_rows(0, rows, oracle.l2_modules(rows))
        for _ in range(100):
            pos = int(rng.integers(0, cnt))
            if pos + 1 != cnt:
                live[pos] = live[cnt - 1]
                ix.move_row(cnt - 1, pos)
            cnt -= 1
            ix.truncate(cnt)
        q, _ = oracle.normalize_copy(make_corpus(9, 1, d)[0])
        want_all = oracle.dist_many(2, q, live[:cnt], oracle.l2_modules(live[:cnt]))
        dist, row, c = ix.search_knn(q, 20)
        wd, wr = lex_topk(want_all, 20)
        assert np.array_equal(row[0], wr) and np.array_equal(bits(dist[0]), bits(wd))


def test_reserve_preserves_rows(rxgpu, oracle):
    d = 128
    rows = make_corpus(10, 300, d)
    q = make_corpus(11, 1, d)[0]
    with rxgpu.VectorIndex("ip", d, 100) as ix:
        ix.upload_rows(0, rows[:100])
        with pytest.raises(rxgpu.RxGpuError):
            ix.upload_rows(100, rows[100:101])      # "The number of elements exceeds the specified limit"
        ix.reserve(300)
        ix.upload_rows(100, rows[100:])
        with pytest.raises(rxgpu.RxGpuError):
            ix.reserve(10)                           # cannot shrink below count (bruteforce.cc:89-91)
        want_all = oracle.dist_many(1, q, rows)
        dist, row, cnt = ix.search_knn(q, 10)
        wd, wr = lex_topk(want_all, 10)
        assert np.array_equal(row[0], wr) and np.array_equal(bits(dist[0]), bits(wd))


def test_denormals_and_extremes(rxgpu, oracle):
    """x86 keeps f32 subnormals; so must the kernels. Also huge magnitudes (inf distances sort last)."""
    n, d = 512, 64
    rng = np.random.default_rng(12)
    rows = (rng.normal(0, 1, (n, d)) * 1e-22).astype(np.float32)   # squares are subnormal / underflow
    rows[7] = 3e19                                                  # squares overflow to inf
    q = (rng.normal(0, 1, d) * 1e-22).astype(np.float32)
    for metric in (0, 1):
        want_all = oracle.dist_many(metric, q, rows)
        with rxgpu.VectorIndex(metric, d, n) as ix:
            ix.upload_rows(0, rows)
            got = ix.distances(q, np.arange(n, dtype=np.uint32))
            assert np.array_equal(bits(got), bits(want_all))
            dist, row, cnt = ix.search_knn(q, 16)
            wd, wr = lex_topk(want_all, 16)
            assert np.array_equal(row[0], wr) and np.array_equal(bits(dist[0]), bits(wd))


def test_property_full_size_roundtrip(rxgpu):
    """Size-independent properties at a size the CPU oracle cannot scan in seconds (2M x 768 = 6 GB):
    every row queried against itself (L2) must come back first with distance exactly 0, sorted output,
    and the result must equal a re-score of the returned rows (rxgpu_distances)."""
    import torch
    n, d = 2_000_000, 768
    g = torch.Generator(device="cuda").manual_seed(5)
    rows = torch.randn((n, d), device="cuda", dtype=torch.float32, generator=g) * 0.25
    with rxgpu.VectorIndex("l2", d) as ix:
        ix.adopt_device_rows(rows.data_ptr(), n, d, None, keepalive=rows)
        probe = torch.tensor([0, 1, 12345, n // 2, n - 1], device="cuda")
        queries = rows[probe].cpu().numpy()
        dist, row, cnt = ix.search_knn(queries, 11)
        for i, p in enumerate(probe.tolist()):
            assert row[i, 0] == p and dist[i, 0] == 0.0
            assert np.all(np.diff(dist[i]) >= 0)
            re = ix.distances(queries[i], row[i])
            assert np.array_equal(bits(re), bits(dist[i]))


@pytest.mark.parametrize("nq", [1, 5])
def test_device_shard_merge_equals_single_index(rxgpu, oracle, nq):
    """The N>1 step without the collective: two row-range shards searched on the device, their raw outputs laid out as the
    all-gather would ([world][2][nq][kk] words), merged by knn_merge_shards — must equal one index over all rows."""
    import torch
    n, d, kk, world = 30_000, 128, 11, 2
    rows = make_corpus(77, n, d)
    rows[n // 2 + 5] = rows[3]          # an exact cross-shard tie: (dist, global row) must order it
    queries = make_corpus(78, nq, d)
    shard = n // world
    t_q = torch.from_numpy(queries).cuda()
    gathered = torch.empty((world, 2, nq, kk), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    shards = []
    for w in range(world):
        ix = rxgpu.VectorIndex("ip", d, shard)
        ix.upload_rows(0, rows[w * shard:(w + 1) * shard])
        base = gathered[w].data_ptr()
        ix.search_knn_device(t_q.data_ptr(), nq, kk, base, base + nq * kk * 4, None, stream)
        shards.append(ix)
    od = torch.empty((nq, kk), dtype=torch.float32, device="cuda")
    orow = torch.empty((nq, kk), dtype=torch.int32, device="cuda")
    rxgpu.merge_shards_device(gathered.data_ptr(), world, nq, kk, shard, od.data_ptr(), orow.data_ptr(), None, stream)
    torch.cuda.synchronize()
    for qi in range(nq):
        wd, wr = lex_topk(oracle.dist_many(1, queries[qi], rows), kk)
        assert np.array_equal(orow[qi].cpu().numpy().view(np.uint32), wr)
        assert np.array_equal(bits(od[qi].cpu().numpy()), bits(wd))
    for ix in shards:
        ix.close()


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("d,n", [(768, 20_000), (128, 60_000), (100, 9_000), (64, 90)])
def test_fused_scan_with_two_entries_per_lane(rxgpu, oracle, metric, d, n):
    """64 < k <= 128 (a hybrid query asks for k = 100): the fused scan keeps two list entries per lane instead of falling to the radix-select
    path.  Same contract: exact rows and distance bits, ties by row — checked on gaussian and on quantised (massively tied) data."""
    from .conftest import lex_topk
    rng = np.random.default_rng(d + metric)
    for style in ("gauss", "quant"):
        rows = make_corpus(d + 3, n, d) if style == "gauss" else rng.integers(-1, 2, (n, d)).astype(np.float32)
        if metric == 2:
            rows[np.all(rows == 0, axis=1)] = 1.0
        inv = oracle.l2_modules(rows) if metric == 2 else None
        queries = make_corpus(4000 + d, 3, d) if style == "gauss" else rng.integers(-1, 2, (3, d)).astype(np.float32)
        if metric == 2:
            queries[np.all(queries == 0, axis=1)] = 1.0
            queries = np.stack([oracle.normalize_copy(q)[0] for q in queries])
        with rxgpu.VectorIndex(metric, d, n) as ix:
            ix.upload_rows(0, rows, inv)
            for k in (65, 100, 101, 128, 129):
                for nq in (1, 3):
                    dist, row, cnt = ix.search_knn(queries[:nq], k)
                    c = min(k, n)
                    for qi in range(nq):
                        wd, wr = lex_topk(oracle.dist_many(metric, queries[qi], rows, inv), c)
                        assert int(cnt[qi]) == c
                        assert np.array_equal(row[qi, :c], wr), (metric, d, style, k, qi)
                        assert np.array_equal(dist[qi, :c].view(np.uint32), wd.view(np.uint32))
