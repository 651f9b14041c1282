"""-m gpu: the visited set of an HNSW search as a HASH SET sized by ef (hnsw_visit in hnsw_search.hip; the reference's visited list is a
tag array over all nodes, visited_list_pool.h:13-36 — the semantics are "seen before", hnswalg.h:904-931) instead of a bitset over the
nodes zeroed by a memset in front of every launch.  Bar: the same answers as the bitset path, bit for bit, the same number of distance
evaluations and hops (= the same traversal); a search that would fill the set leaves as an overflow and is re-run on the bitset; the
sorted-list search, its in-kernel restart on the heaps, the heap kernel, graphs with deleted nodes and quantised graphs all go through it."""
import numpy as np
import pytest

from .conftest import make_corpus
from .test_gpu_hnsw_sorted import as_sorted_pairs, bits, build

pytestmark = pytest.mark.gpu


def _batch(ix, queries, k, ef):
    ix.hnsw_read_stats()
    dist, row, cnt = ix.hnsw_search_knn(queries, k, ef)
    return dist.copy(), row.copy(), cnt.copy(), ix.hnsw_read_stats()


def _same_batches(a, b, nq):
    assert np.array_equal(a[2], b[2])
    for qi in range(nq):
        c = int(a[2][qi])
        x, y = as_sorted_pairs(a[0][qi, :c], a[1][qi, :c]), as_sorted_pairs(b[0][qi, :c], b[1][qi, :c])
        assert np.array_equal(x[1], y[1]) and np.array_equal(bits(x[0]), bits(y[0])), qi


@pytest.mark.parametrize("metric,d", [(0, 128), (2, 768), (1, 96)])
def test_hash_set_equals_bitset(rxgpu, oracle, monkeypatch, metric, d):
    n, nq = 20_000, 600
    m, rows, labels = build(metric, n, d, M=12, efc=60, seed=31 + d)
    g = m.export_graph()
    inv = oracle.l2_modules(rows) if metric == 2 else None
    queries = make_corpus(33, nq, d)
    if metric == 2:
        queries = np.stack([oracle.normalize_copy(q)[0] for q in queries])
    with rxgpu.VectorIndex(metric, d, n) as ix:
        ix.upload_rows(0, rows, inv)
        ix.hnsw_attach_graph(g)
        for sorted_mode in ("1", "0"):                      # the list in registers / the reference's heaps
            monkeypatch.setenv("RXGPU_HNSW_SORTED", sorted_mode)
            for k, ef in ((10, 128), (10, 16), (100, 256), (300, 300)):
                monkeypatch.setenv("RXGPU_HNSW_VISITED", "bitset")
                want = _batch(ix, queries, k, ef)
                monkeypatch.setenv("RXGPU_HNSW_VISITED", "hash")
                got = _batch(ix, queries, k, ef)
                _same_batches(got, want, nq)
                assert got[3] == want[3], (sorted_mode, k, ef, got[3], want[3])   # evaluations and hops: the same traversal
    m.close()


def test_a_search_that_fills_the_set_is_rerun_on_the_bitset(rxgpu, oracle, monkeypatch):
    """2^7 = 128 words: a search may hold 64 nodes — every ef = 128 search overflows and comes back through the global-heap tier (bitset),
    small-ef searches stay; the answers do not change."""
    n, d, nq = 12_000, 64, 200
    m, rows, labels = build(0, n, d, M=8, efc=60, seed=77)
    g = m.export_graph()
    queries = make_corpus(78, nq, d)
    with rxgpu.VectorIndex(0, d, n) as ix:
        ix.upload_rows(0, rows)
        ix.hnsw_attach_graph(g)
        for k, ef in ((10, 128), (3, 4)):
            monkeypatch.setenv("RXGPU_HNSW_VISITED", "bitset")
            monkeypatch.delenv("RXGPU_HNSW_VISITED_LOG2", raising=False)
            want = _batch(ix, queries, k, ef)
            monkeypatch.setenv("RXGPU_HNSW_VISITED", "hash")
            monkeypatch.setenv("RXGPU_HNSW_VISITED_LOG2", "7")
            got = _batch(ix, queries, k, ef)
            _same_batches(got, want, nq)
    m.close()


def test_equal_keys_restart_and_deleted_nodes_with_the_hash_set(rxgpu, oracle, monkeypatch):
    """Rows on an integer grid, every row four times (most searches meet equal keys and start over on the heaps INSIDE the kernel: the set is
    zeroed and refilled), then a third of the nodes marked deleted (HnswSortedListDel / heaps with delete marks): vs the restated engine."""
    from oracle.pyoracle import oracle_hnsw_search_knn
    monkeypatch.setenv("RXGPU_HNSW_VISITED", "hash")
    n, d = 6000, 32
    rng = np.random.default_rng(5)
    base = rng.integers(-3, 4, size=(n // 4, d)).astype(np.float32)
    base[np.all(base == 0, axis=1)] = 1.0
    rows = np.ascontiguousarray(np.repeat(base, 4, axis=0)[rng.permutation(n)])
    m, rows, labels = build(0, n, d, M=8, efc=60, rows=rows)
    g = m.export_graph()
    g["vectors"] = rows
    m.tie_reruns()
    for qi in range(16):
        q = rng.integers(-3, 4, size=d).astype(np.float32)
        for k, ef in ((10, 64), (100, 200)):
            wd, wl = oracle_hnsw_search_knn(oracle, g, q, k, ef, None)
            gd, gl = m.search_knn(q, k, ef)
            assert np.array_equal(gl, wl) and np.array_equal(bits(gd), bits(wd)), (qi, k, ef)
    assert m.tie_reruns() > 0
    m.close()
    # deleted nodes
    m, rows, labels = build(2, 8000, 48, M=10, efc=80, seed=91)
    victims = rng.choice(8000, 2500, replace=False)
    for v in victims:
        m.mark_delete(int(labels[v]))
    queries = make_corpus(92, 12, 48)
    for q in queries:
        for k, ef in ((10, 96), (20, 200)):
            monkeypatch.setenv("RXGPU_HNSW_VISITED", "bitset")
            wd, wl = m.search_knn(q, k, ef)
            monkeypatch.setenv("RXGPU_HNSW_VISITED", "hash")
            gd, gl = m.search_knn(q, k, ef)
            assert np.array_equal(gl, wl) and np.array_equal(bits(gd), bits(wd))
    m.close()


@pytest.mark.parametrize("grid_rows", [False, True])
def test_small_batches_keep_the_set_in_lds(rxgpu, oracle, monkeypatch, grid_rows):
    """D = 768, at most 512 searches in a launch, ef <= 128: the latency form of the kernel keeps the hash set in the workgroup's LDS
    (HnswParams::vis_lds).  Same answers and the same traversal as the bitset; rows on an integer grid make most searches start over on the
    heaps inside the kernel (the LDS set is zeroed and refilled); a set of 2^7 words overflows and the search comes back through the re-run
    with the largest LDS heap (a bitset)."""
    n, d = 6000, 768
    rng = np.random.default_rng(17)
    if grid_rows:
        base = rng.integers(-2, 3, size=(n // 4, d)).astype(np.float32)
        rows = np.ascontiguousarray(np.repeat(base, 4, axis=0)[rng.permutation(n)])
        m, rows, labels = build(0, n, d, M=8, efc=40, rows=rows)
    else:
        m, rows, labels = build(2, n, d, M=12, efc=60, seed=19)
    metric = 0 if grid_rows else 2
    g = m.export_graph()
    inv = oracle.l2_modules(rows) if metric == 2 else None
    with rxgpu.VectorIndex(metric, d, n) as ix:
        ix.upload_rows(0, rows, inv)
        ix.hnsw_attach_graph(g)
        for nq in (1, 7, 300):
            if grid_rows:
                queries = rng.integers(-2, 3, size=(nq, d)).astype(np.float32)
            else:
                queries = np.stack([oracle.normalize_copy(q)[0] for q in make_corpus(40 + nq, nq, d)])
            for k, ef in ((10, 128), (5, 16), (50, 200)):   # ef = 200: the set would take 64 KB and stays in global memory
                monkeypatch.delenv("RXGPU_HNSW_VISITED_LOG2", raising=False)
                monkeypatch.setenv("RXGPU_HNSW_VISITED", "bitset")
                want = _batch(ix, queries, k, ef)
                monkeypatch.delenv("RXGPU_HNSW_VISITED")
                ix.hnsw_read_lds_reruns()
                got = _batch(ix, queries, k, ef)
                _same_batches(got, want, nq)
                assert got[3] == want[3], (nq, k, ef, got[3], want[3])
                assert ix.hnsw_read_lds_reruns() == 0 or grid_rows
                if ef == 128:
                    monkeypatch.setenv("RXGPU_HNSW_VISITED_LOG2", "7")
                    small = _batch(ix, queries, k, ef)
                    _same_batches(small, want, nq)
                    assert ix.hnsw_read_lds_reruns() > 0
    m.close()


@pytest.mark.parametrize("d", [32, 768])
def test_overflowing_searches_of_a_batch_run_on_helper_workgroups(rxgpu, oracle, monkeypatch, d):
    """Batches of >= 2048 queries get helper workgroups on a second stream (hnsw_helper_kernel): a search whose in-kernel restart outgrows
    its LDS heap area is queued and runs with the largest LDS heap while the batch is still going.  Rows on an integer grid, every row four
    times, make most searches restart; a restart area of 6 entries makes every restart overflow (more than the queue holds: the rest takes
    the tiers behind the batch).  Bar: the answers of the same batch with the helpers switched off, and the counter says they ran."""
    n, nq = 6000, 2300
    rng = np.random.default_rng(23)
    base = rng.integers(-2, 3, size=(n // 4, d)).astype(np.float32)
    base[np.all(base == 0, axis=1)] = 1.0
    rows = np.ascontiguousarray(np.repeat(base, 4, axis=0)[rng.permutation(n)])
    m, rows, labels = build(0, n, d, M=8, efc=40, rows=rows)
    g = m.export_graph()
    queries = rng.integers(-2, 3, size=(nq, d)).astype(np.float32)
    with rxgpu.VectorIndex(0, d, n) as ix:
        ix.upload_rows(0, rows, None)
        ix.hnsw_attach_graph(g)
        for cap in (None, "6"):
            if cap:
                monkeypatch.setenv("RXGPU_HNSW_RESTART_CAND", cap)
            for k, ef in ((10, 64), (5, 128)):
                monkeypatch.setenv("RXGPU_HNSW_HELPER", "0")
                want = _batch(ix, queries, k, ef)
                monkeypatch.delenv("RXGPU_HNSW_HELPER")
                ix.hnsw_read_lds_reruns()
                got = _batch(ix, queries, k, ef)
                _same_batches(got, want, nq)
                if cap:
                    assert ix.hnsw_read_lds_reruns() > 0
    m.close()


@pytest.mark.parametrize("metric,d", [(0, 32), (2, 768)])
def test_large_batches_are_searched_in_two_halves(rxgpu, oracle, monkeypatch, metric, d):
    """Batches of >= 8192 queries: the second half of the query block is uploaded on a second stream while the first half's searches run
    (two launches on two streams, RXGPU_HNSW_SPLIT_UPLOAD=0: one upload, one launch).  Same answers — with the bitset, the hash set, SQ8-less
    float graphs, an odd batch size."""
    n, nq = 5000, 8193
    m, rows, labels = build(metric, n, d, M=8, efc=40, seed=61 + d)
    g = m.export_graph()
    inv = oracle.l2_modules(rows) if metric == 2 else None
    queries = make_corpus(62, nq, d)
    if metric == 2:
        queries = queries / np.linalg.norm(queries, axis=1, keepdims=True).astype(np.float32)
    with rxgpu.VectorIndex(metric, d, n) as ix:
        ix.upload_rows(0, rows, inv)
        ix.hnsw_attach_graph(g)
        for visited in ("bitset", "hash"):
            monkeypatch.setenv("RXGPU_HNSW_VISITED", visited)
            monkeypatch.setenv("RXGPU_HNSW_SPLIT_UPLOAD", "0")
            want = _batch(ix, queries, 10, 64)
            monkeypatch.delenv("RXGPU_HNSW_SPLIT_UPLOAD")
            got = _batch(ix, queries, 10, 64)
            _same_batches(got, want, nq)
            assert got[3] == want[3]
    m.close()


def test_concurrent_large_batches_with_helpers_do_not_wait_for_each_other(rxgpu, oracle, monkeypatch):
    """Four searcher threads, each with batches of >= 2048 queries (the size from which helper workgroups are launched), on ONE index at
    once.  A helper is a kernel that polls a queue until its batch says it is over: it is enqueued BEHIND the batch's own launches, so that
    it can never sit in front of them on a hardware queue its stream shares with another caller's batch (where the batch would wait for the
    helper's multi-second bail-out).  Bar: every thread's answers equal the sequential ones, and no call takes anywhere near a bail-out."""
    import threading
    import time
    d, n, nq = 32, 6000, 2300
    rng = np.random.default_rng(29)
    base = rng.integers(-2, 3, size=(n // 4, d)).astype(np.float32)
    base[np.all(base == 0, axis=1)] = 1.0
    rows = np.ascontiguousarray(np.repeat(base, 4, axis=0)[rng.permutation(n)])
    m, rows, labels = build(0, n, d, M=8, efc=40, rows=rows)
    g = m.export_graph()
    queries = [rng.integers(-2, 3, size=(nq, d)).astype(np.float32) for _ in range(4)]
    monkeypatch.setenv("RXGPU_HNSW_RESTART_CAND", "6")   # every restart overflows: the helpers have work
    with rxgpu.VectorIndex(0, d, n) as ix:
        ix.upload_rows(0, rows, None)
        ix.hnsw_attach_graph(g)
        want = [ix.hnsw_search_knn(q, 10, 64) for q in queries]
        want = [(a.copy(), b.copy(), c.copy()) for a, b, c in want]
        ix.hnsw_read_lds_reruns()
        errs, slowest = [], [0.0] * 4

        def work(t):
            try:
                for _ in range(3):
                    t0 = time.perf_counter()
                    got = ix.hnsw_search_knn(queries[t], 10, 64)
                    slowest[t] = max(slowest[t], time.perf_counter() - t0)
                    _same_batches((got[0], got[1], got[2]), want[t], nq)
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))

        ths = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        assert not errs, errs
        assert ix.hnsw_read_lds_reruns() > 0
        assert max(slowest) < 1.5, slowest   # (a batch of 2300 searches over 6000 nodes takes milliseconds; the helpers' bail-out is 3 s)
    m.close()
