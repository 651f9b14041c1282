"""CPU: the SQ8 (scalar-quantised uint8) distance path of the reference — oracle/oracle_sq8.c against the REAL reference code
compiled in place (oracle/_ref: L2SqrDistance<uint8_t>, InnerProductDistance<uint8_t>, Quantizer::Quantize, DistCalculator<uint8_t>) and
against tests/golden/sq8.npz generated from it.  No GPU kernel reads uint8 vectors yet: this pins the checker for that next row.
The uint8 kernels are NOT order-free: the 16 int32 lane sums are added as floats, which rounds beyond 2^24."""
from pathlib import Path

import numpy as np
import pytest

G = Path(__file__).resolve().parent / "golden"
DIMS = [1, 2, 31, 32, 63, 64, 65, 100, 127, 128, 129, 256, 768, 1000, 1024, 1536, 4096]


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def sq8(oracle):
    from oracle.pyoracle import Sq8Oracle
    return Sq8Oracle(oracle)


@pytest.fixture(scope="module")
def sq8ref(ref):
    from oracle.pyoracle import Sq8Ref
    try:
        return Sq8Ref(ref)
    except AttributeError:
        pytest.skip("oracle/_ref/libref_oracle.so predates the SQ8 shims")


@pytest.mark.parametrize("d", DIMS)
def test_u8_distances_match_reference_bits(sq8, sq8ref, d):
    rng = np.random.default_rng(d)
    for style in ("uniform", "extreme", "narrow"):
        for _ in range(60):
            if style == "uniform":
                a, b = rng.integers(0, 256, d), rng.integers(0, 256, d)
            elif style == "extreme":
                a, b = rng.choice([0, 255], d), rng.choice([0, 255], d)
            else:
                a, b = rng.integers(100, 140, d), rng.integers(100, 140, d)
            a, b = a.astype(np.uint8), b.astype(np.uint8)
            assert bits(sq8.l2sqr_u8(a, b)) == bits(sq8ref.l2sqr_u8(a, b)), (style, d)
            assert bits(sq8.ip_u8(a, b)) == bits(sq8ref.ip_u8(a, b)), (style, d)


def test_float_reduction_really_rounds(sq8):
    """255^2 * 1536 = 9.99e7 > 2^24: the sequential float sum of the 16 lane sums is neither the exact integer (99 766 420) nor its
    correctly rounded float (99 766 416) — the lane assignment and the order of that sum are part of the contract."""
    d = 1536
    a, b = np.full(d, 255, np.uint8), np.zeros(d, np.uint8)
    a[::7] = 254
    exact = int(((a.astype(np.int64) - b) ** 2).sum())
    got = float(sq8.l2sqr_u8(a, b))
    assert exact == 99_766_420 and got == 99_766_408.0 and got != float(np.float32(exact))


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("d", [1, 8, 64, 100, 768, 1024])
def test_quantize_and_distcalculator_match_reference(oracle, sq8, sq8ref, metric, d):
    rng = np.random.default_rng(100 * metric + d)
    v = rng.normal(0, 0.25, (40, d)).astype(np.float32)
    v[0] = 10.0     # clamps to 255
    v[1] = -10.0    # clamps to 0
    for lo, hi in ((0.02, 0.98), (0.0, 1.0)):
        min_q, max_q = float(np.quantile(v[2:], lo)), float(np.quantile(v[2:], hi))
        p, pr = sq8.params(min_q, max_q, d), sq8ref.params(min_q, max_q, d)
        assert all(bits(p[k]) == bits(pr[k]) for k in ("alpha", "alpha_2", "delta"))
        stored = []
        for x in v:
            for scale in (1.0, 0.8, 3.5):
                c, o = sq8.quantize(metric, p, x, scale)
                rc, ro = sq8ref.quantize(metric, p, x, scale)
                assert np.array_equal(c, rc) and bits(o) == bits(ro), (metric, d, scale)
            stored.append(sq8.quantize(metric, p, x))
        for i in range(len(v) - 1):
            (a, ca), (b, cb) = stored[i], stored[i + 1]
            assert bits(sq8.dist_pair(metric, p, a, ca, v[i], b, cb, v[i + 1], oracle)) == bits(sq8ref.dist_pair(metric, p, a, ca, v[i], b, cb, v[i + 1]))
            qc, qo = sq8.quantize(metric, p, v[i], 1.25)
            assert bits(sq8.dist_query(metric, p, qc, qo, b, cb, v[i + 1], oracle)) == bits(sq8ref.dist_query(metric, p, qc, qo, b, cb, v[i + 1]))


def test_sq8_oracle_matches_golden_fixture(oracle, sq8):
    """The same pins on machines without /root/reference: values produced by the real reference, committed."""
    z = np.load(G / "sq8.npz")
    from .golden.make_golden import SQ8_DIMS
    for d in SQ8_DIMS:
        a, b = z[f"u8_a_{d}"], z[f"u8_b_{d}"]
        assert np.array_equal(bits([sq8.l2sqr_u8(x, y) for x, y in zip(a, b)]), bits(z[f"u8_l2_{d}"])), d
        assert np.array_equal(bits([sq8.ip_u8(x, y) for x, y in zip(a, b)]), bits(z[f"u8_ip_{d}"])), d
    for metric in (0, 1, 2):
        for d in (8, 100, 768):
            key = f"q_m{metric}_d{d}"
            v, (min_q, max_q) = z[key + "_vec"], z[key + "_minmax"]
            p = sq8.params(float(min_q), float(max_q), d)
            assert np.array_equal(bits([p["alpha"], p["alpha_2"], p["delta"]]), bits(z[key + "_params"]))
            codes, corr = z[key + "_codes"], z[key + "_corr"]
            for i, x in enumerate(v):
                c, o = sq8.quantize(metric, p, x)
                assert np.array_equal(c, codes[i]) and bits(o) == bits(corr[i])
                c, o = sq8.quantize(metric, p, x, 1.25)
                assert np.array_equal(c, z[key + "_qcodes"][i]) and bits(o) == bits(z[key + "_qcorr"][i])
            pair = [sq8.dist_pair(metric, p, codes[i], corr[i], v[i], codes[i + 1], corr[i + 1], v[i + 1], oracle) for i in range(15)]
            assert np.array_equal(bits(pair), bits(z[key + "_pair"]))
            query = [sq8.dist_query(metric, p, z[key + "_qcodes"][i], z[key + "_qcorr"][i], codes[i + 1], corr[i + 1], v[i + 1], oracle)
                     for i in range(15)]
            assert np.array_equal(bits(query), bits(z[key + "_query"]))
            # the row-batch form equals the scalar form
            inv = oracle.l2_modules(v) if metric == 2 else None
            many = sq8.dist_query_many(metric, p, z[key + "_qcodes"][0], z[key + "_qcorr"][0], codes, corr, inv)
            one = [sq8.dist_query(metric, p, z[key + "_qcodes"][0], z[key + "_qcorr"][0], codes[i], corr[i], v[i], oracle) for i in range(len(v))]
            assert np.array_equal(bits(many), bits(one))


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_quantised_hnsw_engine_is_reproduced(oracle, ref, sq8, sq8ref, metric):
    """The whole SQ8 search path against the REAL quantised engine (HierarchicalNSWImpl<uint8_t> built from a float graph the way
    HierarchicalNSW::Quantize does): with the engine's sampled (minQ, maxQ) the restatement reproduces every code, every corrective offset and
    the derived parameters, and SearchKnn over the codes returns the engine's labels and distance bits, deleted nodes included."""
    from oracle.pyoracle import RefHnsw, RefHnswQ, oracle_hnsw_search_knn_sq8
    from .conftest import make_corpus
    n, d = 3000, 64
    rows = make_corpus(61, n, d)
    labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(1)
    h = RefHnsw(ref, metric, d, n, M=16, ef_construction=100)
    h.add(rows, labels)
    for lab in labels[np.random.default_rng(4).choice(n, 100, replace=False)]:
        h.mark_delete(lab)
    g = h.export(with_vectors=False)
    hq = RefHnswQ(h, sample_size=2000)
    sq = hq.export()
    p = sq8.params(float(sq["min_q"]), float(sq["max_q"]), d)
    assert all(bits(p[k]) == bits(sq[k]) for k in ("alpha", "alpha_2", "delta"))
    for i in range(0, n, 7):
        c, o = sq8.quantize(metric, p, rows[i])
        assert np.array_equal(c, sq["codes"][i]) and bits(o) == bits(sq["corr"][i]), i
    inv = oracle.l2_modules(rows) if metric == 2 else None
    for qi in range(40):
        q = make_corpus(700 + qi, 1, d)[0]
        norm = None
        if metric == 2:
            q, k_ = oracle.normalize_copy(q)
            norm = float(np.float32(1.0) / np.float32(k_))   # hnsw_index.cc:168: normL2 = 1.f / NormalizeCopyVector(...)
        for k, ef in ((10, 64), (1, 10), (50, 0)):
            wd, wl = hq.search_knn(q, k, ef, norm)
            gd, gl = oracle_hnsw_search_knn_sq8(oracle, g, sq, q, k, ef, inv, norm)
            assert np.array_equal(gl, wl), (metric, qi, k, ef)
            assert np.array_equal(bits(gd), bits(wd))
    hq.close()
    h.close()


@pytest.mark.parametrize("metric", [0, 2])
def test_quantised_hnsw_search_matches_golden_engine_results(oracle, sq8, metric):
    """The end-to-end pin without /root/reference: a quantised graph exported from the real engine and its SearchKnn results."""
    from oracle.pyoracle import oracle_hnsw_search_knn_sq8
    z = np.load(G / "sq8.npz")
    key = f"hq_m{metric}"
    n, dim, M, maxM0, maxlevel, entry, num_deleted = (int(x) for x in z[key + "_meta"])
    g = dict(metric=metric, n=n, dim=dim, M=M, maxM0=maxM0, maxlevel=maxlevel, entry=entry, num_deleted=num_deleted,
             links0=z[key + "_links0"], upper_off=z[key + "_upper_off"], upper=z[key + "_upper"], levels=z[key + "_levels"],
             labels=z[key + "_labels"], deleted=z[key + "_deleted"])
    min_q, max_q, alpha, alpha_2, delta = z[key + "_params"]
    sq = dict(min_q=min_q, max_q=max_q, alpha=alpha, alpha_2=alpha_2, delta=delta, codes=z[key + "_codes"], corr=z[key + "_corr"])
    rows = z[key + "_rows"]
    p = sq8.params(float(min_q), float(max_q), dim)
    assert bits(p["alpha"]) == bits(alpha) and bits(p["alpha_2"]) == bits(alpha_2) and bits(p["delta"]) == bits(delta)
    for i in range(0, n, 5):
        c, o = sq8.quantize(metric, p, rows[i])
        assert np.array_equal(c, sq["codes"][i]) and bits(o) == bits(sq["corr"][i])
    inv = oracle.l2_modules(rows) if metric == 2 else None
    for i, q in enumerate(z[key + "_queries"]):
        gd, gl = oracle_hnsw_search_knn_sq8(oracle, g, sq, q, 10, 32, inv, float(z[key + "_qnorms"][i]) if metric == 2 else None)
        assert np.array_equal(gl, z[key + "_res_label"][i]) and np.array_equal(bits(gd), bits(z[key + "_res_dist"][i])), i


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_host_quantiser_matches_golden_reference_codes(metric):
    """The PRODUCT quantiser (reindexer_amd/host/sq8_quantizer.h, used by GpuHnswMap::Quantize and for the queries of a quantised Map)
    against codes, offsets and parameters produced by the real Quantizer (tests/golden/sq8.npz)."""
    from reindexer_amd.hostapi import sq8_quantize
    z = np.load(G / "sq8.npz")
    for d in (8, 100, 768):
        key = f"q_m{metric}_d{d}"
        v, (min_q, max_q) = z[key + "_vec"], z[key + "_minmax"]
        for i, x in enumerate(v):
            c, o, params = sq8_quantize(metric, float(min_q), float(max_q), x)
            assert np.array_equal(bits(params), bits(z[key + "_params"]))
            assert np.array_equal(c, z[key + "_codes"][i]) and bits(o) == bits(z[key + "_corr"][i])
            c, o, _ = sq8_quantize(metric, float(min_q), float(max_q), x, 1.25)
            assert np.array_equal(c, z[key + "_qcodes"][i]) and bits(o) == bits(z[key + "_qcorr"][i])


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_host_quantiser_matches_reference_quantizer(sq8ref, metric):
    from reindexer_amd.hostapi import sq8_quantize
    rng = np.random.default_rng(metric)
    for d in (1, 33, 128, 1000):
        v = rng.normal(0, 0.3, (30, d)).astype(np.float32)
        v[0], v[1] = 9.0, -9.0
        min_q, max_q = float(np.quantile(v[2:], 0.01)), float(np.quantile(v[2:], 0.99))
        pr = sq8ref.params(min_q, max_q, d)
        for x in v:
            for scale in (1.0, 0.7, 2.5):
                c, o, params = sq8_quantize(metric, min_q, max_q, x, scale)
                rc, ro = sq8ref.quantize(metric, pr, x, scale)
                assert np.array_equal(c, rc) and bits(o) == bits(ro)
                assert np.array_equal(bits(params), bits([pr["alpha"], pr["alpha_2"], pr["delta"]]))


def _srand(seed):
    import ctypes
    ctypes.CDLL(None).srand(int(seed))


@pytest.mark.parametrize("metric,n,d,sample,quantile", [(0, 2500, 32, 20000, 0.0), (2, 3000, 24, 700, 0.0), (1, 2010, 48, 1000, 0.97), (0, 400, 128, 100, 0.0),
                                                         (2, 61, 768, 20000, 0.0), (0, 1203, 16, 1203, 1.0)])
def test_sampled_quantisation_parameters_equal_the_reference(ref, metric, n, d, sample, quantile):
    """QuantizingParams(hnsw, config) (quantization_params.h:48-66) — reservoir sample by std::rand (hnsw_view_iterator.h:99-113), batches of 20
    rows with a shorter last one, FindNthMinMax per batch (:12-44), means — restated on the host (sq8_quantizer.h: Sq8SampleParams, what
    GpuHnswMap::Quantize(config) runs) against the reference's own constructor over its own float graph, from the same srand() state:
    minQ, maxQ, alpha, alpha_2 and delta bit for bit.  Rows with repeated components (equal minima / maxima: the FIRST occurrence leaves)."""
    from oracle.pyoracle import RefHnsw, RefHnswQ
    from reindexer_amd import hostapi
    rng = np.random.default_rng(n + d)
    rows = rng.normal(0.0, 0.25, (n, d)).astype(np.float32)
    rows[::7] = np.round(rows[::7], 1)            # ties among the extremes
    rows[5] = rows[5].max()
    if metric == 2:
        rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    g = RefHnsw(ref, metric, d, n, M=8, ef_construction=40)
    g.add(rows, labels)
    _srand(4242 + n)
    q = RefHnswQ(g, sample_size=sample, quantile=quantile)
    want = q.export()
    q.close()
    _srand(4242 + n)
    got = hostapi.sq8_sample_params(rows, sample_size=sample, quantile=quantile)
    wantp = np.array([want["min_q"], want["max_q"], want["alpha"], want["alpha_2"], want["delta"]], np.float32)
    assert np.array_equal(bits(got), bits(wantp)), (got, wantp)
    # the sample itself: the same ids from the same generator state
    _srand(99)
    a = hostapi.sq8_sample_indexes(sample, n)
    _srand(99)
    b = hostapi.sq8_sample_indexes(sample, n)
    assert np.array_equal(a, b) and len(a) == min(sample, n) and np.all(np.diff(a.astype(np.int64)) > 0)
    g.close()


def test_find_nth_min_max_walk():
    """FindNthMinMax by hand: n = trunc(0.5 (1 - q) dataSize) passes, each removes the current minimum and maximum (first occurrence)."""
    from reindexer_amd import hostapi
    v = np.array([5, 1, 9, 1, 7, 9, 3, 4], np.float32)
    assert hostapi.sq8_find_nth_min_max(v, 8, 1.0) == (1.0, 9.0)          # n = 0: one pass, nothing removed
    assert hostapi.sq8_find_nth_min_max(v, 8, 0.75) == (1.0, 9.0)         # n = 1
    assert hostapi.sq8_find_nth_min_max(v, 8, 0.5) == (1.0, 9.0)          # n = 2: the second 1 and the second 9
    assert hostapi.sq8_find_nth_min_max(v, 8, 0.25) == (3.0, 7.0)         # n = 3
    assert hostapi.sq8_find_nth_min_max(v, 16, 0.5) == (4.0, 5.0)         # dataSize is what the caller claims: n = 4 on 8 values
