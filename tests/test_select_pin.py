"""CPU: the KNN select post-processing pinned to the reference's own code.
oracle/_ref/libref_select.so = hnsw_index.cc of the reference compiled in place (oracle/ref/ref_select_shim.cc); its
HnswIndexBase<BruteforceSearch>::select / selectRaw (hnsw_index.cc:160-288) run over the reference's own BruteforceSearch map.  Checked against
it: (1) the C restatement oracle/oracle_knn.c::orc_select_postprocess (the checker the GPU tests use), (2) the PRODUCT's host code
reindexer_amd/host/knn_select.h (KnnSelect / KnnSelectRaw) fed with the same search result.  Tie-heavy integer data, array labels (several
vectors per rowId), every combination of K / radius / need_sort / is_array."""
import numpy as np
import pytest

from oracle import pyoracle

pytestmark = pytest.mark.skipif(not pyoracle.ref_select_available(), reason="oracle/_ref/libref_select.so not built (make -C oracle ref)")


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("is_array", [False, True])
def test_select_matches_reference_hnsw_index(oracle, metric, is_array):
    from reindexer_amd import hostapi
    rng = np.random.default_rng(10 * metric + int(is_array))
    n, d = 1200, 8
    rows = rng.integers(-1, 2, (n, d)).astype(np.float32)
    rows[np.all(rows == 0, axis=1), 0] = 1.0
    # array field: many vectors of one row (label = rowId << 32 | array index); scalar field: one vector per row
    row_ids = rng.integers(0, 300, n) if is_array else rng.permutation(n)
    labels = (row_ids.astype(np.uint64) << np.uint64(32)) | (np.arange(n, dtype=np.uint64) if is_array else np.uint64(0))
    ref = pyoracle.RefSelect(metric, d, n, is_array=is_array)
    ref.add(rows, labels)
    inv = oracle.l2_modules(rows) if metric == 2 else None
    checked = 0
    for qi in range(12):
        key = rng.integers(-1, 2, d).astype(np.float32)
        if not key.any():
            key[0] = 1
        q = oracle.normalize_copy(key)[0] if metric == 2 else key
        alld = np.sort(oracle.dist_many(metric, q, rows, inv))
        for k in (1, 7, 40, 300):
            sd, sl = oracle.bf_search_knn(metric, rows, labels, inv, q, k)   # = the Map's SearchKnn result (pinned: tests/test_oracle_vs_ref.py)
            for need_sort in (True, False):
                wi, wr = ref.select(key, k=k, need_sort=need_sort)
                oi, orr = oracle.select_postprocess(metric, sd, sl, need_sort, is_array, k=k)
                pi, pr = hostapi.select_postprocess(metric, sd, sl, k=k, need_sort=need_sort, is_array=is_array)
                assert np.array_equal(oi, wi) and np.array_equal(bits(orr), bits(wr)), ("oracle", metric, qi, k, need_sort)
                assert np.array_equal(pi, wi) and np.array_equal(bits(pr), bits(wr)), ("product", metric, qi, k, need_sort)
                checked += 1
            wi, wr = ref.select_raw(key, k=k)
            pi, pr = hostapi.select_postprocess(metric, sd, sl, k=k, is_array=is_array, raw=True)
            assert np.array_equal(pi, wi) and np.array_equal(bits(pr), bits(wr)), ("raw", metric, qi, k)
        # radius alone and radius + K (removeOverK); the user-facing radius of IP / cosine is the negated distance bound (hnsw_index.cc:185)
        for cut in (5, 60):
            bound = float(alld[cut])
            user_radius = bound if metric == 0 else -bound
            sd, sl = oracle.bf_search_range(metric, rows, labels, inv, q, bound)
            for k in (None, 3, 1000):
                wi, wr = ref.select(key, k=k, radius=user_radius, need_sort=True)
                oi, orr = oracle.select_postprocess(metric, sd, sl, True, is_array, k=k, has_radius=True)
                pi, pr = hostapi.select_postprocess(metric, sd, sl, k=k, has_radius=True, need_sort=True, is_array=is_array)
                assert np.array_equal(oi, wi) and np.array_equal(bits(orr), bits(wr)), ("oracle radius", metric, qi, cut, k)
                assert np.array_equal(pi, wi) and np.array_equal(bits(pr), bits(wr)), ("product radius", metric, qi, cut, k)
                checked += 1
    assert checked > 100
    ref.close()


def test_index_default_radius_applies_when_the_query_has_none(oracle):
    """Opts().FloatVector().Radius(): the index's own radius takes over when the query gives none (hnsw_index.cc:175-177) and still counts
    as 'radius present' for removeOverK."""
    from reindexer_amd import hostapi
    rng = np.random.default_rng(5)
    n, d = 500, 6
    rows = rng.integers(-2, 3, (n, d)).astype(np.float32)
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    ref = pyoracle.RefSelect(0, d, n, index_radius=9.0)
    ref.add(rows, labels)
    key = rng.integers(-2, 3, d).astype(np.float32)
    sd, sl = oracle.bf_search_range(0, rows, labels, None, key, 9.0)
    for k in (None, 4):
        wi, wr = ref.select(key, k=k)
        pi, pr = hostapi.select_postprocess(0, sd, sl, k=k, has_radius=True)
        assert np.array_equal(pi, wi) and np.array_equal(bits(pr), bits(wr))
    ref.close()
