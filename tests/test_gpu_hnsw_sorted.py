"""-m gpu: the sorted-list form of the HNSW search (HnswSortedList in hnsw_search.hip) — graphs without deleted nodes, ef <= 256.
Bar: the same labels and distance bits as the kernel that replays the reference's binary heaps, and as the restated engine; a query that
meets EQUAL distances is handed to the heap kernel (the reference's order among equal keys is its sift order), which the tie counter shows."""
import numpy as np
import pytest

from .conftest import make_corpus

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sq8(oracle):
    from oracle.pyoracle import Sq8Oracle
    return Sq8Oracle(oracle)


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def as_sorted_pairs(dist, ids):
    order = np.lexsort((ids, dist))
    return dist[order], ids[order]


def build(metric, n, d, M=16, efc=100, seed=51, rows=None):
    from reindexer_amd import hostapi
    rows = make_corpus(seed, n, d) if rows is None else rows
    labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(9)
    m = hostapi.GpuHnswMap(metric, d, n, M=M, ef_construction=efc)
    m.add(rows, labels)
    return m, rows, labels


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("d", [128, 96, 768])
def test_sorted_list_equals_heap_kernel_and_restated_engine(rxgpu, oracle, monkeypatch, metric, d):
    """Every ef the list holds (1 .. 256: two or four entries a lane), one past it (257: heaps), k above and below the list length."""
    from oracle.pyoracle import oracle_hnsw_search_knn
    n = 5000 if d == 768 else 9000
    m, rows, labels = build(metric, n, d, seed=51 + d)
    g = m.export_graph()
    g["vectors"] = rows
    inv = oracle.l2_modules(rows) if metric == 2 else None
    m.tie_reruns()
    plans = ((10, 128), (10, 1), (1, 0), (10, 10), (64, 64), (65, 65), (40, 127), (100, 129), (256, 256), (10, 200), (300, 257))
    for qi in range(8):
        q = make_corpus(2100 + qi, 1, d)[0]
        if metric == 2:
            q, _ = oracle.normalize_copy(q)
        for k, ef in plans:
            monkeypatch.setenv("RXGPU_HNSW_SORTED", "1")
            gd, gl = m.search_knn(q, k, ef)
            monkeypatch.setenv("RXGPU_HNSW_SORTED", "0")
            hd, hl = m.search_knn(q, k, ef)
            assert np.array_equal(gl, hl) and np.array_equal(bits(gd), bits(hd)), (metric, d, qi, k, ef)
            if qi < 4:
                wd, wl = oracle_hnsw_search_knn(oracle, g, q, k, ef, inv)
                assert np.array_equal(gl, wl) and np.array_equal(bits(gd), bits(wd)), (metric, d, qi, k, ef)
    # gaussian rows meet equal float32 distances now and then (128 keys in a narrow band); only the consequential ones leave the list
    assert m.tie_reruns() < 8 * len(plans) // 2
    m.close()


def test_sorted_list_batches_through_the_c_abi(rxgpu, oracle, monkeypatch):
    """A batch large enough for the throughput form of the D = 768 kernel (> 3072 searches in one launch), against the heap kernel."""
    n, d, nq = 4000, 768, 3200
    m, rows, labels = build(2, n, d, M=12, efc=60, seed=61)
    g = m.export_graph()
    inv = oracle.l2_modules(rows)
    queries = np.stack([oracle.normalize_copy(q)[0] for q in make_corpus(62, nq, d)])
    with rxgpu.VectorIndex(2, d, n) as ix:
        ix.upload_rows(0, rows, inv)
        ix.hnsw_attach_graph(g)
        out = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("RXGPU_HNSW_SORTED", mode)
            ix.hnsw_read_stats()
            dist, row, cnt = ix.hnsw_search_knn(queries, 10, 128)
            out[mode] = (dist.copy(), row.copy(), cnt.copy(), ix.hnsw_read_stats())
        assert np.array_equal(out["1"][2], out["0"][2])
        assert out["1"][3] == out["0"][3]   # the same distance evaluations and hops: the same traversal
        for qi in range(nq):
            c = int(out["1"][2][qi])
            a = as_sorted_pairs(out["1"][0][qi, :c], out["1"][1][qi, :c])
            b = as_sorted_pairs(out["0"][0][qi, :c], out["0"][1][qi, :c])
            assert np.array_equal(a[1], b[1]) and np.array_equal(bits(a[0]), bits(b[0])), qi
        assert ix.hnsw_read_tie_reruns() < nq // 4   # most searches stay on the list
    m.close()


@pytest.mark.parametrize("restart", [None, "0", "6"])
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_equal_distances_are_rerun_on_the_heaps(rxgpu, oracle, monkeypatch, metric, restart):
    """Rows on a small integer grid, every row four times: most searches meet equal keys.  The answers must still be the restated
    engine's (its heaps decide the order among equal keys), and the counter must show that the heaps served them — restarted inside the
    sorted-list kernel (default), sent back to the host for a launch of their own (RXGPU_HNSW_RESTART_CAND=0), or restarted with a heap
    area so small (6 entries) that the restart overflows: those run once more on the heap kernel with the largest LDS heap, in front of the
    global-heap tiers (which tests/test_gpu_hnsw.py forces)."""
    from oracle.pyoracle import oracle_hnsw_search_knn
    if restart is not None:
        monkeypatch.setenv("RXGPU_HNSW_RESTART_CAND", restart)
    n, d = 6000, 32
    rng = np.random.default_rng(71)
    base = rng.integers(-3, 4, size=(n // 4, d)).astype(np.float32)
    base[np.all(base == 0, axis=1)] = 1.0
    rows = np.ascontiguousarray(np.repeat(base, 4, axis=0)[rng.permutation(n)])
    m, rows, labels = build(metric, n, d, M=8, efc=60, rows=rows)
    g = m.export_graph()
    g["vectors"] = rows
    inv = oracle.l2_modules(rows) if metric == 2 else None
    m.tie_reruns()
    m.lds_reruns()
    searches = 0
    for qi in range(24):
        q = rng.integers(-3, 4, size=d).astype(np.float32) if qi % 2 else rows[rng.integers(n)].copy()
        if metric == 2:
            q, _ = oracle.normalize_copy(q)
        for k, ef in ((10, 64), (5, 8), (100, 200)):
            wd, wl = oracle_hnsw_search_knn(oracle, g, q, k, ef, inv)
            gd, gl = m.search_knn(q, k, ef)
            assert np.array_equal(gl, wl) and np.array_equal(bits(gd), bits(wd)), (metric, qi, k, ef)
            searches += 1
    reruns = m.tie_reruns()
    assert 0 < reruns <= searches, reruns
    assert m.tie_reruns() == 0   # reading resets
    lds = m.lds_reruns()
    assert lds <= searches and (lds > 0 or restart != "6"), (restart, lds)
    m.close()


def test_non_finite_distances_leave_the_sorted_list(rxgpu, oracle):
    """An overflowing inner product (inf) is not a key the list can order: such a search takes the heaps."""
    from oracle.pyoracle import oracle_hnsw_search_knn
    n, d = 3000, 16
    rows = make_corpus(81, n, d)
    rows[::50] *= np.float32(3e19)
    m, rows, labels = build(0, n, d, M=8, efc=40, rows=rows)
    g = m.export_graph()
    g["vectors"] = rows
    m.tie_reruns()
    for qi in range(6):
        q = make_corpus(2300 + qi, 1, d)[0] * np.float32(3e19 if qi % 2 else 1.0)
        wd, wl = oracle_hnsw_search_knn(oracle, g, q, 10, 32)
        gd, gl = m.search_knn(q, 10, 32)
        assert np.array_equal(gl, wl) and np.array_equal(bits(gd), bits(wd)), qi
    m.close()


@pytest.mark.parametrize("metric", [0, 2])
def test_quantised_graph_on_the_sorted_list(rxgpu, oracle, sq8, monkeypatch, metric):
    """SQ8 distances are sums of small integers: equal keys are common, so this walks both the list and the hand-over on one graph."""
    from oracle.pyoracle import oracle_hnsw_search_knn_sq8
    n, d = 5000, 128
    m, rows, labels = build(metric, n, d, seed=91)
    min_q, max_q = float(np.quantile(rows, 0.005)), float(np.quantile(rows, 0.995))
    m.quantize(min_q, max_q)
    p = sq8.params(min_q, max_q, d)
    g = m.export_graph(with_views=True)
    vecs = np.array(g["vectors"])
    stored = [sq8.quantize(metric, p, x) for x in vecs]
    sq = dict(min_q=p["min_q"], alpha=p["alpha"], alpha_2=p["alpha_2"], delta=p["delta"], codes=np.stack([c for c, _ in stored]),
              corr=np.array([o for _, o in stored], np.float32))
    inv = oracle.l2_modules(vecs) if metric == 2 else None
    m.tie_reruns()
    for qi in range(16):
        q = make_corpus(2500 + qi, 1, d)[0]
        norm = None
        if metric == 2:
            q, k_ = oracle.normalize_copy(q)
            norm = float(np.float32(1.0) / np.float32(k_))
        for k, ef in ((10, 128), (10, 16), (60, 250)):
            wd, wl = oracle_hnsw_search_knn_sq8(oracle, g, sq, q, k, ef, inv, norm)
            monkeypatch.setenv("RXGPU_HNSW_SORTED", "1")
            gd, gl = m.search_knn_norm(q, k, ef, norm)
            assert np.array_equal(gl, wl) and np.array_equal(bits(gd), bits(wd)), (metric, qi, k, ef)
            monkeypatch.setenv("RXGPU_HNSW_SORTED", "0")
            hd, hl = m.search_knn_norm(q, k, ef, norm)
            assert np.array_equal(hl, wl) and np.array_equal(bits(hd), bits(wd))
    m.close()


def test_range_search_and_tiny_graphs_on_the_sorted_list(rxgpu, oracle, monkeypatch):
    """SearchRange starts from the same ef-search (k = ef); graphs smaller than ef and a single-node graph."""
    from oracle.pyoracle import oracle_hnsw_search_knn
    from reindexer_amd import hostapi
    for n in (1, 2, 50):
        m, rows, labels = build(0, n, 24, M=4, efc=20, seed=95 + n)
        g = m.export_graph()
        g["vectors"] = rows
        q = make_corpus(96, 1, 24)[0]
        for k, ef in ((10, 128), (1, 1), (n, 256)):
            wd, wl = oracle_hnsw_search_knn(oracle, g, q, k, ef)
            gd, gl = m.search_knn(q, k, ef)
            assert np.array_equal(gl, wl) and np.array_equal(bits(gd), bits(wd)), (n, k, ef)
        radius = float(np.sort(((rows - q) ** 2).sum(axis=1))[min(n, 5) - 1]) * 1.01
        monkeypatch.setenv("RXGPU_HNSW_SORTED", "1")
        rd, rl = m.search_range(q, radius, 64)
        monkeypatch.setenv("RXGPU_HNSW_SORTED", "0")
        hd, hl = m.search_range(q, radius, 64)
        a, b = as_sorted_pairs(rd, rl), as_sorted_pairs(hd, hl)
        assert np.array_equal(a[1], b[1]) and np.array_equal(bits(a[0]), bits(b[0])), n
        monkeypatch.delenv("RXGPU_HNSW_SORTED")
        m.close()


@pytest.mark.parametrize("metric", [0, 2])
@pytest.mark.parametrize("del_frac", [0.03, 0.4])
def test_graphs_with_deleted_nodes_on_the_sorted_list(rxgpu, oracle, monkeypatch, metric, del_frac):
    """HnswSortedListDel: deleted nodes are candidates, never results; 2 / 3 / 4 entries a lane by ef (96 / 160 / 224), heaps above.
    Against the heap kernel on every query and against the restated engine; the entry point is deleted in the second round."""
    from oracle.pyoracle import oracle_hnsw_search_knn
    n, d = 7000, 128
    m, rows, labels = build(metric, n, d, M=12, efc=80, seed=131)
    inv = oracle.l2_modules(rows) if metric == 2 else None
    rng = np.random.default_rng(9)
    plans = ((10, 64), (10, 96), (20, 97), (10, 128), (40, 160), (10, 161), (64, 224), (10, 225), (5, 0), (1, 1))
    for phase in range(2):
        victims = labels[rng.choice(n, int(n * del_frac / 2), replace=False)]
        g = m.export_graph()
        if phase:
            victims = np.unique(np.concatenate([victims, labels[[int(g["entry"])]]]))
        for lab in victims:
            try:
                m.mark_delete(lab)
            except Exception:
                pass   # already deleted in the first round
        g = m.export_graph()
        g["vectors"] = rows
        assert g["num_deleted"] > 0
        for qi in range(10):
            q = make_corpus(3100 + qi, 1, d)[0]
            if metric == 2:
                q, _ = oracle.normalize_copy(q)
            for k, ef in plans:
                monkeypatch.setenv("RXGPU_HNSW_SORTED", "1")
                gd, gl = m.search_knn(q, k, ef)
                monkeypatch.setenv("RXGPU_HNSW_SORTED", "0")
                hd, hl = m.search_knn(q, k, ef)
                assert np.array_equal(gl, hl) and np.array_equal(bits(gd), bits(hd)), (metric, phase, qi, k, ef)
                if qi < 4:
                    wd, wl = oracle_hnsw_search_knn(oracle, g, q, k, ef, inv)
                    assert np.array_equal(gl, wl) and np.array_equal(bits(gd), bits(wd)), (metric, phase, qi, k, ef)
    monkeypatch.delenv("RXGPU_HNSW_SORTED")
    m.close()
