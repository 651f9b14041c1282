"""CPU-only: pins the plain-C restatement (oracle/oracle_knn.c) bit-for-bit to the REAL reference engines
compiled in place from /root/reference (oracle/_ref).  Skipped where the reference build is absent
(the GPU box relies on tests/golden/ instead, generated from the same reference build)."""
import numpy as np
import pytest

from oracle.pyoracle import RefBruteforce
from .conftest import make_corpus

DIMS = [1, 3, 7, 15, 16, 17, 31, 33, 48, 63, 64, 65, 100, 128, 130, 200, 256, 384, 512, 768, 1000, 1024, 1536]


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("d", DIMS)
def test_distance_bits_match_reference(oracle, ref, d):
    rows = make_corpus(100 + d, 4000, d)
    q = make_corpus(7 + d, 1, d)[0]
    for metric in (0, 1):
        want = ref.dist_many(metric, q, rows)
        got = oracle.dist_many(metric, q, rows) if metric == 0 else -oracle.dist_many(1, q, rows)
        assert np.array_equal(bits(want), bits(got)), f"metric={metric} d={d}"


def test_distance_bits_one_million_pairs(oracle, ref):
    """the equality is a property of the pinned oracle build; keep a large bitwise test."""
    for d in (128, 768):
        n = 1_000_000 * 128 // d // 2
        rows = make_corpus(11, n, d)
        for qi in range(2):
            q = make_corpus(1000 + qi, 1, d)[0]
            for metric in (0, 1):
                want = ref.dist_many(metric, q, rows)
                got = oracle.dist_many(0, q, rows) if metric == 0 else -oracle.dist_many(1, q, rows)
                assert np.array_equal(bits(want), bits(got))


@pytest.mark.parametrize("d", [1, 5, 64, 100, 128, 768])
def test_norm_coefficient_bits(oracle, ref, d):
    rng = np.random.default_rng(d)
    for it in range(300):
        x = rng.normal(0, 0.25, d).astype(np.float32)
        if it % 4 == 0:
            x = (x / max(np.linalg.norm(x), 1e-9)).astype(np.float32)  # exercises the |1-|x|^2| <= 1e-5 shortcut
        if it % 97 == 0:
            x[:] = 0
        assert bits(ref.l2_module(x)) == bits(oracle.l2_module(x))
        a, ka = ref.normalize_copy(x)
        b, kb = oracle.normalize_copy(x)
        assert np.array_equal(bits(a), bits(b)) and bits(ka) == bits(kb)


def quantized_corpus(seed, n, d):
    """Few distinct values per component => many exactly equal distances (tie semantics)."""
    rng = np.random.default_rng(seed)
    return rng.integers(-1, 2, (n, d)).astype(np.float32)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("kind", ["gauss", "ties"])
def test_bruteforce_knn_and_range_match_reference(oracle, ref, metric, kind):
    d, n = (128, 3000) if kind == "gauss" else (8, 1500)
    rows = make_corpus(5, n, d) if kind == "gauss" else quantized_corpus(5, n, d)
    labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(0)
    rng = np.random.default_rng(9)
    labels = labels[rng.permutation(n)]  # label order != insertion order
    bf = RefBruteforce(ref, metric, d, n)
    bf.add(rows, labels)
    # delete some rows: the reference swaps the last row into the hole (bruteforce.cc:70-86)
    live_rows, live_labels = rows.copy(), labels.copy()
    cnt = n
    for victim in rng.choice(n, 40, replace=False):
        lab = labels[victim]
        pos = int(np.nonzero(live_labels[:cnt] == lab)[0][0])
        bf.remove(lab)
        if pos + 1 != cnt:
            live_rows[pos] = live_rows[cnt - 1]
            live_labels[pos] = live_labels[cnt - 1]
        cnt -= 1
    assert bf.count == cnt
    live_rows, live_labels = live_rows[:cnt], live_labels[:cnt]
    inv = oracle.l2_modules(live_rows) if metric == 2 else None
    for qi in range(25):
        q = make_corpus(100 + qi, 1, d)[0] if kind == "gauss" else quantized_corpus(100 + qi, 1, d)[0]
        if metric == 2:
            q, _ = oracle.normalize_copy(q)
        for k in (1, 10, 37, cnt, cnt + 5):
            wd, wl = bf.search_knn(q, k)
            gd, gl = oracle.bf_search_knn(metric, live_rows, live_labels, inv, q, k)
            assert np.array_equal(wl, gl), f"labels differ q={qi} k={k}"
            assert np.array_equal(bits(wd), bits(gd))
        radius = float(np.sort(oracle.dist_many(metric, q, live_rows, inv))[50]) + 1e-6
        wd, wl = bf.search_range(q, radius)
        gd, gl = oracle.bf_search_range(metric, live_rows, live_labels, inv, q, radius)
        assert np.array_equal(wl, gl) and np.array_equal(bits(wd), bits(gd))
    bf.close()


def test_empty_and_k0(oracle, ref):
    bf = RefBruteforce(ref, 0, 16, 8)
    q = np.zeros(16, np.float32)
    assert bf.search_knn(q, 5)[0].size == 0
    rows = make_corpus(1, 4, 16)
    labels = np.arange(4, dtype=np.uint64)
    assert oracle.bf_search_knn(0, rows[:0], labels[:0], None, q, 5)[0].size == 0
    bf.add(rows, labels)
    assert bf.search_knn(q, 0)[0].size == 0
    assert oracle.bf_search_knn(0, rows, labels, None, q, 0)[0].size == 0
    bf.close()


# ------------------------------------------------------------------------------------------------ HNSW search
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_hnsw_search_restatement_matches_reference(oracle, ref, metric):
    """Graph built by the REAL engine, exported flat; the C restatement of the search must return exactly what the
    engine's own SearchKnn returns (same heaps, same tie mechanics) — with and without deleted nodes."""
    from oracle.pyoracle import RefHnsw, oracle_hnsw_search_knn
    n, d = 3000, 64
    rows = make_corpus(21, n, d)
    labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(1)
    h = RefHnsw(ref, metric, d, n, M=16, ef_construction=200)
    h.add(rows, labels)
    for phase in range(2):
        if phase == 1:
            for lab in labels[np.random.default_rng(3).choice(n, 150, replace=False)]:
                h.mark_delete(lab)
        g = h.export()
        assert g["num_deleted"] == (150 if phase else 0)
        inv = oracle.l2_modules(g["vectors"]) if metric == 2 else None
        for qi in range(40):
            q = make_corpus(500 + qi, 1, d)[0]
            if metric == 2:
                q, _ = oracle.normalize_copy(q)
            for k, ef in ((10, 128), (10, 10), (1, 0), (50, 64)):
                wd, wl = h.search_knn(q, k, ef)
                gd, gl = oracle_hnsw_search_knn(oracle, g, q, k, ef, inv)
                assert np.array_equal(wl, gl), (metric, phase, qi, k, ef)
                assert np.array_equal(bits(wd), bits(gd))
    h.close()


def _sorted_batch(d, l):
    o = np.lexsort((l, d))
    return d[o], l[o]


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_hnsw_streaming_restatement_matches_reference(oracle, ref, metric):
    """Whole streaming sessions (BeginStreamingSearch + ContinueStreamingSearch until exhausted, hnswalg.h:1865-1975) against the real
    engine: every batch must hold exactly the same (dist, label) pairs, and `exhausted` must flip at the same call."""
    from oracle.pyoracle import OracleHnswStream, RefHnsw
    n, d = 1200, 32
    rows = make_corpus(77, n, d)
    labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(2)
    h = RefHnsw(ref, metric, d, n, M=8, ef_construction=100)
    h.add(rows, labels)
    for phase in range(2):
        if phase == 1:
            for lab in labels[np.random.default_rng(4).choice(n, 100, replace=False)]:
                h.mark_delete(lab)
        g = h.export()
        inv = oracle.l2_modules(g["vectors"]) if metric == 2 else None
        for qi, (ef, batches) in enumerate([(0, [10] * 8), (16, [5, 40, 1, 300, 7]), (64, [64] * 40), (3, [1] * 30 + [2000])]):
            q = make_corpus(900 + qi, 1, d)[0]
            if metric == 2:
                q, _ = oracle.normalize_copy(q)
            rs, os_ = h.stream(q, ef), OracleHnswStream(oracle, g, q, ef, inv)
            total = 0
            for b in batches:
                wd, wl, wex = rs.next(b)
                gd, gl, gex = os_.next(b)
                assert wex == gex and len(wd) == len(gd), (metric, phase, qi, b)
                a, c = _sorted_batch(wd, wl), _sorted_batch(gd, gl)
                assert np.array_equal(a[1], c[1]) and np.array_equal(bits(a[0]), bits(c[0])), (metric, phase, qi, b)
                total += len(wd)
            if batches[-1] >= n:
                live = n - (100 if phase else 0)
                assert wex and 0.95 * live <= total <= live     # a full drain returns every REACHABLE live element exactly once
            rs.close()
            os_.close()
    h.close()


# ---------------------------------------------------------------- bench.py infrastructure: graph import + timed multi-thread baselines
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_ref_graph_import_roundtrip(ref, metric):
    """A graph written INTO the real engine by ref_hnsw_import_graph searches exactly like the engine that built it (labels + distance
    bits, deleted nodes included) and exports the same flat graph; a graph built by the product's concurrent builder imports too."""
    from oracle.pyoracle import RefHnsw
    from reindexer_amd import hostapi
    n, d, M, efc = 1500, 48, 8, 60
    rows = make_corpus(77 + metric, n, d)
    labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(1)
    a = RefHnsw(ref, metric, d, n, M=M, ef_construction=efc)
    a.add(rows, labels)
    for lab in labels[np.random.default_rng(2).choice(n, 40, replace=False)]:
        a.mark_delete(lab)
    g = a.export()
    b = RefHnsw(ref, metric, d, n + 10, M=M, ef_construction=efc)
    b.import_graph(g)
    g2 = b.export()
    for key in ("n", "maxlevel", "entry", "num_deleted"):
        assert g[key] == g2[key], key
    for key in ("links0", "levels", "labels", "deleted", "upper_off", "vectors"):
        assert np.array_equal(g[key], g2[key]), key
    assert np.array_equal(g["upper"][: int(g["upper_off"][-1])], g2["upper"][: int(g["upper_off"][-1])])
    qs = make_corpus(5, 24, d)
    od, ol, cnt = b.search_knn_many(qs, 10, 50)
    for i, q in enumerate(qs):
        wd, wl = a.search_knn(q, 10, 50)
        assert cnt[i] == len(wl) and np.array_equal(ol[i, : cnt[i]], wl) and np.array_equal(od[i, : cnt[i]].view(np.uint32), wd.view(np.uint32))
    # inserting after an import keeps working (label table, deleted-slot set, levels, norms are in place): the engine is constructed with
    # ReplaceDeleted_True, so the 5 new points take 5 of the 40 vacated slots (hnswalg.h:1410-1421) exactly as in the engine that built it
    extra = make_corpus(9, 5, d)
    # (WHICH slots is the iteration order of a hash set filled in a different order by initTree — also true of the reference's own LoadIndex)
    b.add(extra, np.arange(5, dtype=np.uint64) + np.uint64(7))
    gb = b.export()
    assert b.count == n and gb["num_deleted"] == 35
    for i in range(5):
        _, wl = b.search_knn(extra[i], 1, 50)
        assert wl[0] == 7 + i
    # the product's multithreaded builder -> import -> the reference's SearchKnn runs on it
    hg = hostapi.HnswGraph(metric, d, n, M=M, ef_construction=efc)
    hg.add(rows, labels, threads=4)
    e = hg.export()
    e["vectors"] = rows
    c = RefHnsw(ref, metric, d, n, M=M, ef_construction=efc)
    c.import_graph(e)
    from oracle.pyoracle import Oracle, oracle_hnsw_search_knn
    orc = Oracle()
    inv = orc.l2_modules(rows) if metric == 2 else None
    for q in qs[:8]:
        if metric == 2:
            q, _ = orc.normalize_copy(q)
        wd, wl = oracle_hnsw_search_knn(orc, e, q, 10, 50, inv_norms=inv)
        rd, rl = c.search_knn(q, 10, 50)
        assert np.array_equal(np.sort(rl), np.sort(wl)) and np.array_equal(np.sort(rd).view(np.uint32), np.sort(wd).view(np.uint32))
    secs, done = c.search_knn_mt(qs, 10, 50, threads=3, per_thread=5)
    assert done == 15 and secs > 0
    for x in (a, b, c):
        x.close()
    hg.close()


def test_ref_bruteforce_timed_threads(ref):
    from oracle.pyoracle import RefBruteforce
    rows = make_corpus(3, 4000, 64)
    bf = RefBruteforce(ref, 1, 64, 4000)
    bf.add(rows, np.arange(4000, dtype=np.uint64))
    secs, done = bf.search_knn_mt(make_corpus(4, 6, 64), 10, threads=4, per_thread=8)
    assert done == 32 and 0 < secs < 30
    secs, done = bf.search_knn_mt(make_corpus(4, 6, 64), 10, threads=2, per_thread=1_000_000, deadline_s=0.05)
    assert 2 <= done < 2_000_000
    bf.close()
