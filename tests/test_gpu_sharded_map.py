"""-m gpu: GpuBruteforceMap over a DEVICE LIST (BASELINE configs[3] behind the C++ seam: rxgpu_index_create_sharded, row-range shards, one
worker thread and one device per shard, per-shard exact top-k lists merged under (dist, global row)).  The 1-GPU test box lists the same
device several times — the code path is the multi-GPU one.  Bar: identical to the single-device Map and to the reference engine — labels
and distance bits — including exact distance ties that straddle the k-th boundary ACROSS a shard boundary with labels that are not in
row order (the reference evicts by (dist, label), bruteforce.cc:117-124), swap-with-last deletes that move rows between shards,
pre-filtered and range searches."""
import numpy as np
import pytest

from .conftest import make_corpus

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def hostapi(rxgpu):
    from reindexer_amd import hostapi as h
    h.lib()
    return h


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("shards", [2, 3, 8])
def test_sharded_map_equals_single_device_map_and_oracle(hostapi, oracle, metric, shards):
    rng = np.random.default_rng(100 * metric + shards)
    n, d = 5000, 40
    rows = make_corpus(7 + metric, n, d)
    labels = (rng.permutation(n).astype(np.uint64) << np.uint64(32)) | np.uint64(3)   # label order != row order
    one = hostapi.GpuBruteforceMap(metric, d, n + 16)
    many = hostapi.GpuBruteforceMap(metric, d, n + 16, devices=[0] * shards)
    for m in (one, many):
        m.add(rows, labels)
    live_rows, live_labels = rows.copy(), labels.copy()
    for step in range(3):
        inv = oracle.l2_modules(live_rows) if metric == 2 else None
        for qi in range(6):
            q = make_corpus(300 + qi + 10 * step, 1, d)[0]
            if metric == 2:
                q, _ = oracle.normalize_copy(q)
            for k in (1, 10, 100):
                wd, wl = oracle.bf_search_knn(metric, live_rows, live_labels, inv, q, k)
                ad, al = one.search_knn(q, k)
                bd, bl = many.search_knn(q, k)
                assert np.array_equal(bl, wl) and np.array_equal(bits(bd), bits(wd)), (metric, shards, step, qi, k)
                assert np.array_equal(bl, al) and np.array_equal(bits(bd), bits(ad))
            radius = float(np.sort(oracle.dist_many(metric, q, live_rows, inv))[40])
            ad, al = one.search_range(q, radius)
            bd, bl = many.search_range(q, radius)
            assert np.array_equal(bl, al) and np.array_equal(bits(bd), bits(ad))
            allowed = live_labels[rng.random(live_labels.shape[0]) < 0.15]
            ad, al = one.search_knn_filtered(q, 10, allowed)
            bd, bl = many.search_knn_filtered(q, 10, allowed)
            assert np.array_equal(bl, al) and np.array_equal(bits(bd), bits(ad))
        # swap-with-last deletes: the last row moves into the hole, often from another shard
        victims = rng.choice(live_labels.shape[0], 120, replace=False)
        for v in sorted(victims.tolist(), reverse=True):
            lab = live_labels[v]
            one.remove(lab)
            many.remove(lab)
            last = live_labels.shape[0] - 1
            live_rows[v], live_labels[v] = live_rows[last], live_labels[last]
            live_rows, live_labels = live_rows[:last], live_labels[:last]
    one.close()
    many.close()


@pytest.mark.parametrize("metric", [0, 1])
def test_cross_shard_tie_at_the_kth_boundary_is_resolved_by_label(hostapi, oracle, metric):
    """Quantised data: hundreds of rows share each distance, spread over every shard, with permuted labels.  The k-th boundary then cuts
    through a group of equal distances whose members live in different shards; the reference keeps the ones its (dist, label) heap keeps."""
    rng = np.random.default_rng(5 + metric)
    n, d = 4096, 16
    rows = rng.integers(-1, 2, (n, d)).astype(np.float32)
    labels = (rng.permutation(n).astype(np.uint64) << np.uint64(32)) | np.uint64(1)
    many = hostapi.GpuBruteforceMap(metric, d, n, devices=[0, 0, 0, 0])
    many.add(rows, labels)
    for qi in range(8):
        q = rng.integers(-1, 2, d).astype(np.float32)
        for k in (5, 37, 64, 200):
            wd, wl = oracle.bf_search_knn(metric, rows, labels, None, q, k)
            gd, gl = many.search_knn(q, k)
            assert np.array_equal(gl, wl), (metric, qi, k)
            assert np.array_equal(bits(gd), bits(wd))
    assert many.tie_replays > 0
    many.close()


def test_sharded_c_abi_contract(rxgpu):
    """The C-ABI directly: global rows in and out, no holes, fixed capacity, single-device-only entry points refused loudly."""
    import ctypes as C
    from reindexer_amd import capi
    L = capi.lib()
    h = C.c_void_p()
    devs = (C.c_int * 3)(0, 0, 0)
    assert L.rxgpu_index_create_sharded(1, 24, 1000, 3, devs, C.byref(h)) == 0
    assert L.rxgpu_index_shard_count(h) == 3 and L.rxgpu_index_shard_rows(h) == 352   # ceil(1000 / 3) rounded up to 32
    rows = make_corpus(1, 900, 24)
    assert L.rxgpu_index_upload_rows(h, 0, 900, rows.ctypes.data, None) == 0
    assert L.rxgpu_index_count(h) == 900 and L.rxgpu_index_capacity(h) == 1000
    assert L.rxgpu_index_upload_rows(h, 950, 10, rows.ctypes.data, None) != 0            # a hole
    assert L.rxgpu_index_reserve(h, 2000) == capi.RXGPU_ERR_LOGIC
    q = make_corpus(2, 4, 24)
    dist, row, cnt = np.zeros((4, 11), np.float32), np.zeros((4, 11), np.uint32), np.zeros(4, np.uint32)
    assert L.rxgpu_search_knn(h, q.ctypes.data, 4, 11, dist.ctypes.data, row.ctypes.data, cnt.ctypes.data) == 0
    with capi.VectorIndex(1, 24, 1000) as ix:
        ix.upload_rows(0, rows)
        d1, r1, c1 = ix.search_knn(q, 11)
    assert np.array_equal(row, r1) and np.array_equal(dist.view(np.uint32), d1.view(np.uint32)) and np.array_equal(cnt, c1)
    assert row.max() > 704, "hits must come from the last shard too (global rows)"
    out = np.zeros(5, np.float32)
    pick = np.array([899, 0, 352, 351, 704], np.uint32)
    assert L.rxgpu_distances(h, q[0].ctypes.data, pick.ctypes.data, 5, out.ctypes.data) == 0
    with capi.VectorIndex(1, 24, 1000) as ix:
        ix.upload_rows(0, rows)
        assert np.array_equal(out.view(np.uint32), ix.distances(q[0], pick).view(np.uint32))
    assert L.rxgpu_search_knn_device(h, None, 1, 1, None, None, None, None) != 0
    L.rxgpu_index_destroy(h)


@pytest.mark.parametrize("mode", ["rccl", "host"])
def test_knn_lists_meet_in_one_rccl_all_gather_behind_the_c_abi(rxgpu, oracle, monkeypatch, mode):
    """the per-shard top-k lists travel in ONE all-gather inside librxgpu.so (the path the C++ Map reaches), the merge runs on
    the first device; RXGPU_SHARD_MERGE=host keeps the host merge.  Both equal the single-device index bit for bit — batch 1 and a batch,
    k + 1 = 11 and 64, partially filled last shard, an EMPTY last shard, equal distances straddling shard boundaries."""
    from reindexer_amd import capi
    if mode == "host":
        monkeypatch.setenv("RXGPU_SHARD_MERGE", "host")
    else:
        monkeypatch.delenv("RXGPU_SHARD_MERGE", raising=False)
    n, d = 3000, 48
    rows = make_corpus(21, n, d)
    rows[1500] = rows[10]            # exact ties across shards: (dist, global row) decides
    rows[2900] = rows[10]
    for metric in (0, 1, 2):
        inv = oracle.l2_modules(rows) if metric == 2 else None
        for shards, fill in ((2, n), (3, n), (5, 2000), (4, 700)):
            with capi.ShardedVectorIndex(metric, d, n, [0] * shards) as sx, capi.VectorIndex(metric, d, n) as one:
                assert sx.merge_mode == mode and sx.ranks == (1 if mode == "rccl" else 0)
                sx.upload_rows(0, rows[:fill], inv[:fill] if inv is not None else None)
                one.upload_rows(0, rows[:fill], inv[:fill] if inv is not None else None)
                q = make_corpus(40 + metric, 9, d)
                q[0] = rows[10]
                if metric == 2:
                    q = np.stack([oracle.normalize_copy(v)[0] for v in q])
                before = sx.collectives
                calls = 0
                for kk in (1, 11, 64):
                    for qs in (q[:1], q):
                        ad, ar, ac = one.search_knn(qs, kk)
                        bd, br, bc = sx.search_knn(qs, kk)
                        calls += 1
                        assert np.array_equal(br, ar) and np.array_equal(bits(bd), bits(ad)) and np.array_equal(bc, ac), (metric, shards, fill, kk)
                # every shard non-empty holds >= 64 rows here, so every call above went through the exchange
                assert sx.collectives - before == (calls if mode == "rccl" else 0)
                ad, ar, ac = one.search_knn(q, 100)     # kk > 64: host merge in either mode
                bd, br, bc = sx.search_knn(q, 100)
                assert np.array_equal(br, ar) and np.array_equal(bits(bd), bits(ad))


def test_rccl_exchange_with_concurrent_callers(rxgpu):
    """Several planner threads on one sharded index: every fan-out takes its own lane (streams + buffers), the collectives of the one
    communicator are enqueued under a lock — results stay those of the single-device index."""
    import threading
    from reindexer_amd import capi
    n, d = 4000, 64
    rows = make_corpus(5, n, d)
    q = make_corpus(6, 32, d)
    with capi.ShardedVectorIndex(1, d, n, [0, 0, 0]) as sx, capi.VectorIndex(1, d, n) as one:
        assert sx.merge_mode == "rccl"
        sx.upload_rows(0, rows)
        one.upload_rows(0, rows)
        want = one.search_knn(q, 11)
        bad = []

        def worker(t):
            for it in range(20):
                i = (t * 7 + it) % 32
                gd, gr, gc = sx.search_knn(q[i:i + 1], 11)
                if not (np.array_equal(gr[0], want[1][i]) and np.array_equal(bits(gd[0]), bits(want[0][i]))):
                    bad.append((t, it))
        ts = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not bad
        assert sx.collectives >= 120


def test_shards_filled_in_place_from_device_memory(rxgpu):
    """bench.py --in-process: every shard adopts rows generated on ITS device; the sharded handle takes the counts over."""
    import torch
    from reindexer_amd import capi
    n_per, d = 2048, 32
    with capi.ShardedVectorIndex(1, d, 3 * n_per, [0, 0, 0]) as sx, capi.VectorIndex(1, d, 3 * n_per) as one:
        parts = [torch.randn((n_per, d), device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(s)) * 0.25 for s in range(3)]
        for s, p in enumerate(parts):
            sx.shard(s).adopt_device_rows(p.data_ptr(), n_per, d)
        sx.sync_count()
        assert sx.count == 3 * n_per
        one.upload_rows(0, torch.cat(parts).cpu().numpy())
        q = make_corpus(9, 5, d)
        ad, ar, ac = one.search_knn(q, 11)
        bd, br, bc = sx.search_knn(q, 11)
        assert np.array_equal(br, ar) and np.array_equal(bits(bd), bits(ad))
        del parts


def test_in_tree_constructor_shape_takes_its_device_list_from_the_environment(hostapi, oracle, monkeypatch):
    """§8(e) "Host topology" through the seam: the reference constructs `Map(metric, dim, maxElements)` (hnsw_index.cc:61-66) and the in-tree
    adapter (rx_seam.h GpuBruteforceMapInTree) reads the device list from RX_GPU_VECTOR_INDEXES.  "0,0,0" builds the three-shard Map — the path
    an 8-GPU node takes with "0-7" — and must answer like the single-device Map and the reference engine."""
    n, d = 3000, 32
    rows = make_corpus(77, n, d)
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    monkeypatch.setenv("RX_GPU_VECTOR_INDEXES", "0,0,0")
    many = hostapi.GpuBruteforceMap.from_env(1, d, n)
    monkeypatch.setenv("RX_GPU_VECTOR_INDEXES", "0")
    one = hostapi.GpuBruteforceMap.from_env(1, d, n)
    monkeypatch.delenv("RX_GPU_VECTOR_INDEXES")
    dflt = hostapi.GpuBruteforceMap.from_env(1, d, n)
    assert many.sharded and not one.sharded and not dflt.sharded
    for m in (one, many, dflt):
        m.add(rows, labels)
    for qi in range(8):
        q = make_corpus(500 + qi, 1, d)[0]
        wd, wl = oracle.bf_search_knn(1, rows, labels, None, q, 10)
        for m in (one, many, dflt):
            gd, gl = m.search_knn(q, 10)
            assert np.array_equal(gl, wl) and np.array_equal(bits(gd), bits(wd))
    for m in (one, many, dflt):
        m.close()
