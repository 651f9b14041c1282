"""-m gpu: the VECTOR half of the drop-in boundary, EXECUTED (the FT half: tests/test_gpu_ft_seam.py).

oracle/_ref/libref_knn_seam.so (oracle/ref/ref_knn_seam_shim.cc) is the reference's own cpp_src/core/index/float_vector/hnsw_index.cc with
integration/patches/0001-hnsw_index-gpu-maps.patch applied, instantiated four times: HnswIndexBase over the reference's BruteforceSearch and
HierarchicalNSW, and over the product's GpuBruteforceMapInTree and GpuHnswMapT<None> (the product's host sources compiled against the
reference's FloatVectorId / ConstFloatVectorView / SearchResultQueue).  The same rows go through upsert / del of both, and what the PATCHED
REFERENCE CODE returns from select / selectRaw / beginStreaming + continueStreaming over the GPU Map must be what it returns over its own
engine: the IdSet in the same order, the ranks to the bit, equal-distance runs by ascending row id, one entry per row of an array field,
the k + radius truncation (hnsw_index.cc:159-288, float_vector_index.h:140-160).

BASELINE configs[0] (the minimum slice): 100 000 x 128 fp32, L2, k = 10, 1000 single queries through that seam against the
reference's CPU float_vector index."""
import numpy as np
import pytest

from .conftest import make_corpus

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def seam(rxgpu):
    from oracle import pyoracle
    if not pyoracle.ref_knn_seam_available():
        pytest.skip("oracle/_ref/libref_knn_seam.so not available (built where /root/reference exists)")
    return pyoracle.RefKnnSeam


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))


def pair(seam, kinds, metric, dim, cap, is_array=False, **kw):
    return seam(kinds[0], metric, dim, cap, is_array, **kw), seam(kinds[1], metric, dim, cap, is_array, **kw)


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_brute_force_through_the_patched_hnsw_index(seam, metric):
    """upsert (with the resize of a full Map, hnsw_index.cc:90-92), del (swap-with-last in both engines), select / selectRaw with K, radius,
    both, with and without the id sort."""
    n, d = 6000, 96
    rows = make_corpus(300 + metric, n, d)
    rng = np.random.default_rng(metric)
    rows[rng.choice(n, 40, replace=False)] = rows[7]   # equal vectors: runs of equal distance, ordered by row id in the result
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    ref, gpu = pair(seam, ("ref_bf", "gpu_bf"), metric, d, 1000)   # 1000 < n: both Maps grow through HnswIndexBase::newSize
    for a in range(0, n, 1500):
        ref.upsert(rows[a:a + 1500], labels[a:a + 1500])
        gpu.upsert(rows[a:a + 1500], labels[a:a + 1500])
    assert ref.count == gpu.count == n
    queries = make_corpus(900 + metric, 12, d)
    queries[0] = rows[7]

    def check(tag):
        for qi, q in enumerate(queries):
            for k, radius in ((10, None), (1, None), (64, None), (300, None), (None, 4.0 if metric == 0 else 0.2), (50, 6.0 if metric == 0 else 0.05)):
                for need_sort in (True, False):
                    a, b = ref.select(q, k=k, radius=radius, need_sort=need_sort), gpu.select(q, k=k, radius=radius, need_sort=need_sort)
                    assert same(a, b), (tag, metric, qi, k, radius, need_sort, len(a[0]), len(b[0]))
                assert same(ref.select_raw(q, k=k, radius=radius), gpu.select_raw(q, k=k, radius=radius)), (tag, metric, qi, k, radius)

    check("full")
    for lab in labels[rng.choice(n, 700, replace=False)]:
        ref.delete(int(lab))
        gpu.delete(int(lab))
    assert ref.count == gpu.count == n - 700
    check("after deletes")
    more = make_corpus(950 + metric, 300, d)
    more_labels = (np.arange(n, n + 300, dtype=np.uint64)) << np.uint64(32)
    ref.upsert(more, more_labels)
    gpu.upsert(more, more_labels)
    check("after re-inserts")
    ref.close()
    gpu.close()


def test_brute_force_array_field_one_entry_per_row(seam):
    """An array-of-vectors field: the label carries the array position in its low word, select() keeps the best entry of a row
    (removeDuplicateRowId, float_vector_index.h:140-160) — so fewer than k rows may come back."""
    rows_n, per_row, d = 900, 4, 48
    vecs = make_corpus(77, rows_n * per_row, d)
    labels = ((np.repeat(np.arange(rows_n, dtype=np.uint64), per_row) << np.uint64(32)) | np.tile(np.arange(per_row, dtype=np.uint64), rows_n))
    ref, gpu = pair(seam, ("ref_bf", "gpu_bf"), 0, d, rows_n * per_row, is_array=True)
    ref.upsert(vecs, labels)
    gpu.upsert(vecs, labels)
    for q in make_corpus(78, 10, d):
        for k in (5, 40, 200):
            a, b = ref.select(q, k=k), gpu.select(q, k=k)
            assert same(a, b) and len(set(a[0].tolist())) == len(a[0]) <= k
            assert same(ref.select_raw(q, k=k), gpu.select_raw(q, k=k))
    ref.close()
    gpu.close()


@pytest.mark.parametrize("metric", [0, 2])
def test_hnsw_through_the_patched_hnsw_index(seam, metric):
    """Both engines build their graph from the same single-thread insert order (the product's builder restates the reference's link for
    link, tests/test_hnsw_builder.py), so the patched adapter must return the same lists from either: select / selectRaw at several ef,
    MarkDelete through del(), and streaming sessions batch for batch."""
    n, d = 3000, 64
    rows = make_corpus(500 + metric, n, d)
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    ref, gpu = pair(seam, ("ref_hnsw", "gpu_hnsw"), metric, d, n, M=8, ef_construction=100)
    ref.upsert(rows, labels)
    gpu.upsert(rows, labels)
    queries = make_corpus(600 + metric, 10, d)

    def check(tag):
        for qi, q in enumerate(queries):
            for k, ef in ((10, 64), (1, 16), (30, 128)):
                for need_sort in (True, False):
                    a, b = ref.select(q, k=k, ef=ef, need_sort=need_sort), gpu.select(q, k=k, ef=ef, need_sort=need_sort)
                    assert same(a, b), (tag, metric, qi, k, ef, need_sort, a[0][:5], b[0][:5])
                assert same(ref.select_raw(q, k=k, ef=ef), gpu.select_raw(q, k=k, ef=ef)), (tag, metric, qi, k, ef)
            a, b = ref.select(q, k=20, radius=(5.0 if metric == 0 else 0.1), ef=64), gpu.select(q, k=20, radius=(5.0 if metric == 0 else 0.1), ef=64)
            assert same(a, b), (tag, metric, qi, "radius")

    check("built")
    rng = np.random.default_rng(3 + metric)
    for lab in labels[rng.choice(n, 250, replace=False)]:
        ref.delete(int(lab))
        gpu.delete(int(lab))
    check("after MarkDelete")
    for q in queries[:4]:
        ref.begin_streaming(q, 48)
        gpu.begin_streaming(q, 48)
        for step in range(6):
            a, b = ref.continue_streaming(16), gpu.continue_streaming(16)
            assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and a[2] == b[2], (metric, step, a[0][:4], b[0][:4])
            if a[2]:
                break
    ref.close()
    gpu.close()


def test_config0_100k_x_128_l2_k10_1000_queries_through_the_seam(seam):
    """BASELINE configs[0]: the reference's own CPU-runnable case, as SURVEY §7.2 defines the minimum slice — through the Map behind the
    patched HnswIndexBase, against the reference's engine behind the unpatched members of the same class."""
    n, d, k, nq = 100_000, 128, 10, 1000
    rows = make_corpus(20260924 % 100000, n, d)
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    ref, gpu = pair(seam, ("ref_bf", "gpu_bf"), 0, d, n)
    ref.upsert(rows, labels)
    gpu.upsert(rows, labels)
    queries = make_corpus(7, nq, d)
    bad = 0
    for q in queries:
        a, b = ref.select(q, k=k, need_sort=False), gpu.select(q, k=k, need_sort=False)
        bad += int(not same(a, b))
    assert bad == 0, f"{bad} of {nq} queries differ"
    ref.close()
    gpu.close()
