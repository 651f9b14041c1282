"""GpuBruteforceMap (the drop-in `Map` for HnswIndexBase<Map>) against the CPU oracle / golden vectors.
CPU part: the host-side norm helpers.  -m gpu part: full Map semantics incl. the k-th-boundary tie replay,
swap-deletes, clone, select() post-processing and the reference's error behaviour."""
from pathlib import Path

import numpy as np
import pytest

from .conftest import make_corpus

G = Path(__file__).resolve().parent / "golden"


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("d", [1, 5, 64, 100, 768])
def test_host_norm_helpers_match_oracle(oracle, d):
    from reindexer_amd import hostapi
    rng = np.random.default_rng(d)
    for it in range(100):
        x = rng.normal(0, 0.25, d).astype(np.float32)
        if it % 4 == 0:
            x = (x / max(np.linalg.norm(x), 1e-9)).astype(np.float32)
        if it % 31 == 0:
            x[:] = 0
        assert bits(hostapi.l2_module(x)) == bits(oracle.l2_module(x))
        a, ka = hostapi.normalize_copy(x)
        b, kb = oracle.normalize_copy(x)
        assert np.array_equal(bits(a), bits(b)) and bits(ka) == bits(kb)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def hostapi(rxgpu):
    from reindexer_amd import hostapi as h
    h.lib()
    return h


def build_pair(hostapi, oracle, metric, rows, labels, victims=()):
    """Same mutation history for the GPU Map and for the oracle's flat arrays."""
    n, d = rows.shape
    m = hostapi.GpuBruteforceMap(metric, d, n)
    m.add(rows, labels)
    live_rows, live_labels = rows.copy(), labels.copy()
    cnt = n
    for lab in victims:
        m.remove(lab)
        pos = int(np.nonzero(live_labels[:cnt] == lab)[0][0])
        if pos + 1 != cnt:
            live_rows[pos] = live_rows[cnt - 1]
            live_labels[pos] = live_labels[cnt - 1]
        cnt -= 1
    live_rows, live_labels = live_rows[:cnt].copy(), live_labels[:cnt].copy()
    inv = oracle.l2_modules(live_rows) if metric == 2 else None
    assert m.count == cnt
    return m, live_rows, live_labels, inv


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("kind", ["gauss", "ties"])
def test_map_knn_range_match_oracle(hostapi, oracle, metric, kind):
    rng = np.random.default_rng(17)
    n, d = (3000, 128) if kind == "gauss" else (2000, 8)
    rows = make_corpus(5, n, d) if kind == "gauss" else rng.integers(-1, 2, (n, d)).astype(np.float32)
    labels = (rng.permutation(n).astype(np.uint64) << np.uint64(32)) | rng.integers(0, 3, n).astype(np.uint64)
    victims = labels[rng.choice(n, 60, replace=False)]
    m, lrows, llabels, inv = build_pair(hostapi, oracle, metric, rows, labels, victims)
    cnt = lrows.shape[0]
    for qi in range(20):
        q = make_corpus(100 + qi, 1, d)[0] if kind == "gauss" else rng.integers(-1, 2, d).astype(np.float32)
        if metric == 2:
            q, _ = oracle.normalize_copy(q)
        for k in (1, 2, 10, 37, 63, 64, 65, 200, cnt, cnt + 3):
            wd, wl = oracle.bf_search_knn(metric, lrows, llabels, inv, q, k)
            gd, gl = m.search_knn(q, k)
            assert np.array_equal(gl, wl), (kind, metric, qi, k)
            assert np.array_equal(bits(gd), bits(wd))
        alld = np.sort(oracle.dist_many(metric, q, lrows, inv))
        for radius in (float(alld[25]), float(alld[0]), float(alld[-1]) + 1):
            wd, wl = oracle.bf_search_range(metric, lrows, llabels, inv, q, radius)
            gd, gl = m.search_range(q, radius)
            assert np.array_equal(gl, wl) and np.array_equal(bits(gd), bits(wd))
    if kind == "ties":
        assert m.tie_replays > 0, "the tie dataset must exercise the k-th-boundary replay"
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_map_matches_golden_ties(hostapi, metric):
    """Golden vectors from the REAL reference incl. exact distance ties, swap-deletes and array labels."""
    g = np.load(G / "bruteforce.npz")
    rows, labels = g["ties_rows"], g["ties_labels"]
    n, d = rows.shape
    m = hostapi.GpuBruteforceMap(metric, d, n)
    m.add(rows, labels)
    for v in g["ties_victims"]:
        m.remove(labels[v])
    queries = g["ties_queries"]
    for qi in range(queries.shape[0]):
        q = queries[qi]
        if metric == 2:
            q, _ = hostapi.normalize_copy(q)
        for k in (1, 10, 64, 100, n):
            gd, gl = m.search_knn(q, k)
            assert np.array_equal(gl, g[f"ties_m{metric}_q{qi}_k{k}_label"]), (qi, k)
            assert np.array_equal(bits(gd), bits(g[f"ties_m{metric}_q{qi}_k{k}_dist"]))
        gd, gl = m.search_range(q, float(g[f"ties_m{metric}_q{qi}_radius"][0]))
        assert np.array_equal(gl, g[f"ties_m{metric}_q{qi}_range_label"])
        assert np.array_equal(bits(gd), bits(g[f"ties_m{metric}_q{qi}_range_dist"]))
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_select_postprocessing_matches_oracle(hostapi, oracle, metric):
    """HnswIndexBase::select: rank sign, rowId = label >> 32, equal-rank runs sorted by id, array dedupe, removeOverK."""
    rng = np.random.default_rng(23)
    n, d = 1500, 8
    rows = rng.integers(-1, 2, (n, d)).astype(np.float32)
    labels = (rng.integers(0, 400, n).astype(np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)  # array index: many rows per rowId
    m, lrows, llabels, inv = build_pair(hostapi, oracle, metric, rows, labels)
    for qi in range(15):
        key = rng.integers(-1, 2, d).astype(np.float32)
        if not key.any():
            key[0] = 1
        q = oracle.normalize_copy(key)[0] if metric == 2 else key
        for k in (1, 10, 50):
            wd, wl = oracle.bf_search_knn(metric, lrows, llabels, inv, q, k)
            for need_sort in (True, False):
                for is_array in (True, False):
                    wi, wr = oracle.select_postprocess(metric, wd, wl, need_sort, is_array, k=k)
                    gi, gr = m.select(key, k=k, need_sort=need_sort, is_array=is_array)
                    assert np.array_equal(gi, wi), (metric, qi, k, need_sort, is_array)
                    assert np.array_equal(bits(gr), bits(wr))
        # radius (user-facing sign: IP / cosine radius is negated by search(), hnsw_index.cc:185), alone and with k
        alld = np.sort(oracle.dist_many(metric, q, lrows, inv))
        user_radius = float(alld[30]) if metric == 0 else -float(alld[30])
        wd, wl = oracle.bf_search_range(metric, lrows, llabels, inv, q, float(alld[30]))
        wi, wr = oracle.select_postprocess(metric, wd, wl, True, True, k=None, has_radius=True)
        gi, gr = m.select(key, radius=user_radius, need_sort=True, is_array=True)
        assert np.array_equal(gi, wi) and np.array_equal(bits(gr), bits(wr))
        wi, wr = oracle.select_postprocess(metric, wd, wl, True, False, k=5, has_radius=True)
        gi, gr = m.select(key, k=5, radius=user_radius, need_sort=True, is_array=False)
        assert np.array_equal(gi, wi) and np.array_equal(bits(gr), bits(wr))
    m.close()


@pytest.mark.gpu
def test_map_error_behaviour_and_accessors(hostapi):
    d = 16
    rows = make_corpus(1, 8, d)
    m = hostapi.GpuBruteforceMap(0, d, 4)
    assert m.max_elements == 4 and m.count == 0 and m.element_size == d * 4 + 8
    assert m.search_knn(rows[0], 3)[0].size == 0                      # empty index
    m.add(rows[:4], np.arange(4, dtype=np.uint64))
    with pytest.raises(hostapi.HostError, match="exceeds the specified limit"):
        m.add(rows[4:5], np.array([4], np.uint64))
    m.add(rows[5:6], np.array([2], np.uint64))                        # existing label => overwrite in place
    assert m.count == 4 and np.array_equal(m.vector_by_label(2), rows[5])
    with pytest.raises(hostapi.HostLogicError, match="does not support concurrent insertions"):
        m.add_concurrent(rows[0], 9)
    with pytest.raises(hostapi.HostError, match="Label not found"):
        m.vector_by_label(77)
    with pytest.raises(hostapi.HostError, match="Cannot resize"):
        m.resize(2)
    m.remove(12345)                                                    # unknown label: silently ignored
    m.resize(8)
    m.add(rows[4:5], np.array([4], np.uint64))
    assert m.count == 5 and m.search_knn(rows[4], 1)[1][0] == 4
    with pytest.raises(hostapi.HostError, match="KNN limit should not be 0"):
        m.select(rows[0], k=0)
    with pytest.raises(hostapi.HostError, match="can not be empty both"):
        m.select(rows[0])
    with pytest.raises(hostapi.HostError, match="dimension"):
        m.select(rows[0][:8], k=1)
    m.close()


@pytest.mark.gpu
def test_map_clone_is_independent(hostapi, oracle):
    """Copy-with-capacity (Index::Clone for copy-on-write transactions, hnsw_index.cc:68-70,125-128)."""
    d, n = 64, 500
    rows = make_corpus(3, n + 10, d)
    labels = np.arange(n + 10, dtype=np.uint64) << np.uint64(32)
    m = hostapi.GpuBruteforceMap(1, d, n)
    m.add(rows[:n], labels[:n])
    c = m.clone(n + 10)
    c.add(rows[n:], labels[n:])
    m.remove(labels[0])
    q = rows[n + 3]
    wd, wl = oracle.bf_search_knn(1, rows, labels, None, q, 5)
    gd, gl = c.search_knn(q, 5)
    assert np.array_equal(gl, wl) and np.array_equal(bits(gd), bits(wd))
    assert c.count == n + 10 and m.count == n - 1
    m.close()
    c.close()


@pytest.mark.gpu
def test_concurrent_searches_share_one_map(hostapi, oracle):
    """Reads are re-entrant under the namespace shared lock (reference: runMultithreadQueries, float_vector_index.cc:258-294)."""
    import threading
    d, n = 128, 20000
    rows = make_corpus(4, n, d)
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    m = hostapi.GpuBruteforceMap(0, d, n)
    m.add(rows, labels)
    queries = make_corpus(5, 32, d)
    want = [oracle.bf_search_knn(0, rows, labels, None, q, 10)[1] for q in queries]
    errs = []

    def run(t):
        for i in range(t, 32, 4):
            for _ in range(3):
                if not np.array_equal(m.search_knn(queries[i], 10)[1], want[i]):
                    errs.append(i)
    ths = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs
    m.close()
