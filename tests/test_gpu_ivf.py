"""-m gpu: GpuIvfFlat and the range scan over a row list.
The index is held to the DEFINITION of IVF-Flat in the engine's own arithmetic — the result of a query is the exact (dist,row)-ordered search
over the rows of the nprobe nearest lists, distances bit-identical to the brute-force oracle — and to recall against the exact search (what the
reference's own IVF tests assert).  That definition itself is pinned bit for bit against the reference's vendored FAISS, given FAISS's trained
state, in tests/test_ivf_oracle.py (CPU).  TRAINING is pinned here: centroids and inverted lists equal, bit for bit, those of the vendored FAISS
built in place (oracle/_ref/libref_ivf.so) with its k-means assignment on the exact distance functions."""
import numpy as np
import pytest

from .conftest import lex_topk, make_corpus
from .test_ivf_oracle import check_topk_whatever_the_tie_rule

pytestmark = pytest.mark.gpu


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def hostapi(rxgpu):
    from reindexer_amd import hostapi as h
    h.lib()
    return h


def clustered(seed, n, d, clusters=64):
    rng = np.random.default_rng(seed)
    centres = rng.normal(0, 0.25, (clusters, d)).astype(np.float32)
    return (centres[rng.integers(0, clusters, n)] + rng.normal(0, 0.05, (n, d))).astype(np.float32)


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_range_subset_matches_oracle(rxgpu, oracle, metric):
    n, d = 5000, 96
    rng = np.random.default_rng(metric)
    rows = make_corpus(12, n, d)
    inv = oracle.l2_modules(rows) if metric == 2 else None
    q = make_corpus(13, 1, d)[0]
    if metric == 2:
        q, _ = oracle.normalize_copy(q)
    with rxgpu.VectorIndex(metric, d, n) as ix:
        ix.upload_rows(0, rows, inv)
        for dens in (0.01, 0.3, 1.0):
            ids = np.flatnonzero(rng.random(n) < dens).astype(np.uint32) if dens < 1 else np.arange(n, dtype=np.uint32)
            dist = oracle.dist_many(metric, q, rows[ids], inv[ids] if inv is not None else None)
            srt = np.sort(dist)
            for radius in (float(srt[min(20, srt.size - 1)]), float(srt[0]), float(srt[-1]) + 1.0):
                for inclusive in (False, True):
                    keep = dist <= radius if inclusive else dist < radius
                    wd, wpos = lex_topk(np.where(keep, dist, np.inf), int(keep.sum()))
                    gd, gr = ix.search_range_subset(q, radius, ids, inclusive=inclusive, cap=8)   # cap too small on purpose: overflow + retry
                    assert np.array_equal(gr, ids[wpos]) and np.array_equal(bits(gd), bits(wd)), (dens, radius, inclusive)
        assert ix.search_range_subset(q, 1e9, np.empty(0, np.uint32))[0].size == 0
        with pytest.raises(rxgpu.RxGpuError):
            ix.search_range_subset(q, 1.0, np.array([4, 4], np.uint32))


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_ivf_equals_exact_search_over_the_probed_lists(hostapi, oracle, metric):
    n, d, nlist = 20000, 64, 64
    rows = clustered(3, n, d)
    ids = (np.arange(n, dtype=np.int64) * 7 + 1000)
    ivf = hostapi.GpuIvfFlat(metric, d, nlist)
    ivf.add_with_ids(rows[:2000], ids[:2000])
    assert not ivf.is_trained and ivf.ntotal == 2000
    inv_all = oracle.l2_modules(rows) if metric == 2 else None
    sign = 1.0 if metric == 0 else -1.0

    def exact(qp, cand_rows, k):
        dist = oracle.dist_many(metric, qp, rows[cand_rows], inv_all[cand_rows] if inv_all is not None else None)
        wd, wpos = lex_topk(dist, min(k, cand_rows.size))
        return wd, cand_rows[wpos]

    q0 = clustered(4, 1, d)[0]
    qp0 = oracle.normalize_copy(q0)[0] if metric == 2 else q0
    # flat phase == IndexFlat: exact over everything
    gd, gl = ivf.search(q0, 10, nprobe=1)
    wd, wr = exact(qp0, np.arange(2000), 10)
    assert np.array_equal(gl, ids[wr]) and np.array_equal(bits(gd * sign), bits(wd))
    ivf.add_with_ids(rows[2000:12000], ids[2000:12000])
    ivf.train()
    assert ivf.is_trained
    ivf.add_with_ids(rows[12000:], ids[12000:])   # after training: straight into the lists
    assert ivf.ntotal == n and int(ivf.list_sizes().sum()) == n and int((ivf.list_sizes() == 0).sum()) <= 4
    cent = ivf.centroids()
    if metric != 0:
        assert np.allclose(np.linalg.norm(cent, axis=1), 1.0, atol=1e-5)   # spherical k-means for inner product / cosine
    hits = total = 0
    queries = clustered(5, 30, d)
    for q in queries:
        qp = oracle.normalize_copy(q)[0] if metric == 2 else q
        for nprobe in (1, 4, 16, nlist):
            cand = ivf.probed_rows(q, nprobe)
            assert np.all(np.diff(cand.astype(np.int64)) > 0)
            full = oracle.dist_many(metric, qp, rows[cand], inv_all[cand] if inv_all is not None else None)
            for k in (1, 10, 100, 300):
                gd, gl = ivf.search(q, k, nprobe=nprobe)
                # (which of several vectors AT the k-th distance are kept, and the order of equal distances, follow FAISS's scanner:
                # test_ties_follow_the_faiss_scanner; here the answer is checked without assuming a tie rule)
                check_topk_whatever_the_tie_rule(gd * sign, gl, full, ids[cand], k)
            if nprobe == nlist:
                assert cand.size == n   # every list probed == exact search
            # range search over the same lists
            radius_internal = float(np.sort(full)[min(25, full.size - 1)])
            gd, gl = ivf.range_search(q, radius_internal * sign, nprobe=nprobe, cap=4)
            keep = full < radius_internal
            wd, wpos = lex_topk(np.where(keep, full, np.inf), int(keep.sum()))
            assert np.array_equal(gl, ids[cand[wpos]]) and np.array_equal(bits(gd * sign), bits(wd))
        # recall@10 at nprobe = 8 against the exact search over everything
        wd, wr = exact(qp, np.arange(n), 10)
        gd, gl = ivf.search(q, 10, nprobe=8)
        hits += len(set(ids[wr].tolist()) & set(gl.tolist()))
        total += 10
    # maximum-inner-product search through unit centroids is the weak case of IVF (the winner need not sit near its centroid's direction)
    assert hits / total >= (0.8 if metric == 1 else 0.9), hits / total
    # removals: swap-delete keeps lists, ids and the device mirror in step
    rng = np.random.default_rng(9)
    victims = rng.choice(n, 500, replace=False)
    assert ivf.remove_ids(ids[victims]) == 500 and ivf.remove_ids(ids[victims[:10]]) == 0
    assert ivf.ntotal == n - 500 and int(ivf.list_sizes().sum()) == n - 500
    alive = np.ones(n, bool)
    alive[victims] = False
    alive_rows = np.flatnonzero(alive)
    for q in queries[:10]:
        qp = oracle.normalize_copy(q)[0] if metric == 2 else q
        gd, gl = ivf.search(q, 20, nprobe=nlist)   # all lists probed: must equal the exact search over the survivors
        full = oracle.dist_many(metric, qp, rows[alive_rows], inv_all[alive_rows] if inv_all is not None else None)
        check_topk_whatever_the_tie_rule(gd * sign, gl, full, ids[alive_rows], 20)
        assert not (set(gl.tolist()) & set(ids[victims].tolist()))
    with pytest.raises(hostapi.HostError, match="already present"):
        ivf.add_with_ids(rows[:1], ids[alive_rows[:1]])
    ivf.reset()
    assert ivf.ntotal == 0 and not ivf.is_trained
    ivf.close()


def test_ivf_errors(hostapi):
    ivf = hostapi.GpuIvfFlat(0, 8, 16)
    with pytest.raises(hostapi.HostError, match="at least as large as number of clusters"):
        ivf.train()
    d, l = ivf.search(np.zeros(8, np.float32), 3, nprobe=2)
    assert np.all(l == -1) and np.all(np.isinf(d))
    ivf.close()


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d,nlist,clusters", [(2500, 32, 16, 16), (9000, 48, 32, 200), (700, 24, 16, 3), (3000, 40, 8, 64)])
def test_training_equals_vendored_faiss_bit_for_bit(hostapi, metric, n, d, nlist, clusters):
    """IvfIndex's trained state (ivf_index.cc:96-108, 469-487): IndexIVFFlat::train -> Level1Quantizer::train_q1 -> Clustering::train_encoded
    (rand_perm initialisation with seed 1234 + 1, 10 iterations, single-precision centroid sums in data order, split_clusters with its own
    RandomGenerator(1234), spherical renormalisation for inner product / cosine; (9000, 32 lists) exceeds 256 points per centroid: the
    rand_perm subsample; (700, 16 lists, 3 clusters) leaves clusters empty: split_clusters runs) and add_with_ids.  The product's centroids
    and lists must be FAISS's, bit for bit."""
    from oracle.pyoracle import RefIvf, ref_ivf_available
    if not ref_ivf_available():
        pytest.skip("oracle/_ref/libref_ivf.so not built")
    rows = clustered(50 + metric, n, d, clusters)
    ids = (np.arange(n, dtype=np.int64) * 7 + 3)
    ref = RefIvf(metric, d, nlist, rows, ids, exact_assignment=True)
    cent_ref, lists_ref = ref.export()
    g = hostapi.GpuIvfFlat(metric, d, nlist)
    g.add_with_ids(rows, ids)
    g.train()
    cent = g.centroids()
    assert np.array_equal(bits(cent), bits(cent_ref)), (metric, int((bits(cent) != bits(cent_ref)).sum()))
    for c in range(nlist):
        assert np.array_equal(np.sort(g.list_ids(c)), np.sort(lists_ref[c])), (metric, c)
    # searches then agree label for label (the search definition is pinned separately)
    q = clustered(99, 1, d, clusters)[0]
    qref = hostapi.normalize_copy(q)[0] if metric == 2 else q   # IvfIndex normalises a cosine query before it reaches FAISS
    for nprobe in (1, 4):
        gd, gl = g.search(q, 10, nprobe)
        rd, rl = ref.search(qref, 10, nprobe)
        assert np.array_equal(gl, rl) and np.array_equal(bits(gd), bits(rd))
    g.close()
    ref.close()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_ties_follow_the_faiss_scanner(hostapi, metric):
    """Exact copies among the vectors: equal distances inside a result and at its k-th place.  FAISS's scanner keeps the first-scanned of
    equal distances at the boundary and evicts by (distance, id) resp. (similarity, id); GpuIvfFlat replays that over the tied candidates
    (scan order = probed lists in coarse order, every list in ArrayInvertedLists order, which remove_ids reshuffles).  Labels AND their
    order must be the vendored FAISS's, before and after removals, through the device list search (nprobe <= 64)."""
    from oracle.pyoracle import RefIvf, ref_ivf_available
    if not ref_ivf_available():
        pytest.skip("oracle/_ref/libref_ivf.so not built")
    rng = np.random.default_rng(60 + metric)
    n, d, nlist = 4000, 12, 16
    base = clustered(80 + metric, n, d, 12)
    rows = np.where((rng.random(n) < 0.35)[:, None], base[rng.integers(0, n, n)], base).astype(np.float32)
    ids = rng.permutation(n * 3)[:n].astype(np.int64)
    ref = RefIvf(metric, d, nlist, rows, ids, exact_assignment=True)
    g = hostapi.GpuIvfFlat(metric, d, nlist)
    g.add_with_ids(rows, ids)
    g.train()
    alive = np.ones(n, bool)
    boundary_ties = 0
    for round_ in range(3):
        for _ in range(30):
            q = (rows[rng.integers(0, n)] + rng.normal(0, 0.02, d)).astype(np.float32)
            qref = hostapi.normalize_copy(q)[0] if metric == 2 else q
            nprobe, k = int(rng.integers(1, nlist + 1)), int(rng.choice([1, 3, 10, 50]))
            gd, gl = g.search(q, k, nprobe)
            rd, rl = ref.search(qref, k, nprobe)
            assert np.array_equal(gl, rl), (metric, round_, nprobe, k, gl[:12], rl[:12])
            assert np.array_equal(bits(gd[rl >= 0]), bits(rd[rl >= 0]))
            rd1, _ = ref.search(qref, k + 1, nprobe)
            boundary_ties += int(k < len(rd1) and rd1[k] == rd1[k - 1] and np.isfinite(rd1[k]))
        victims = ids[rng.choice(np.flatnonzero(alive), 500, replace=False)]
        assert ref.remove_ids(victims) == 500 and g.remove_ids(victims) == 500
        alive &= ~np.isin(ids, victims)
    assert boundary_ties > 5
    g.close()
    ref.close()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_c_abi_list_search_equals_subset_search_over_the_probed_lists(rxgpu, oracle, metric):
    """rxgpu_search_knn_lists (coarse search -> bitmap of the probed lists -> row list -> scan, one call) == rxgpu_search_knn_subset over the
    union of the lists of the nprobe nearest centroids; error behaviour of the two entries."""
    n, d, nlist = 6000, 40, 24
    rng = np.random.default_rng(200 + metric)
    rows = clustered(7, n, d, 30)
    cents = clustered(8, nlist, d, 30)
    owner = rng.integers(0, nlist, n)
    lists = [np.flatnonzero(owner == l).astype(np.uint32) for l in range(nlist)]
    lists[5] = np.empty(0, np.uint32)   # an empty list
    kept = np.concatenate(lists)
    inv = oracle.l2_modules(rows) if metric == 2 else None
    cinv = oracle.l2_modules(cents) if metric == 2 else None
    with rxgpu.VectorIndex(metric, d, n) as ix, rxgpu.VectorIndex(metric, d, nlist) as cx:
        ix.upload_rows(0, rows, inv)
        cx.upload_rows(0, cents, cinv)
        with pytest.raises(rxgpu.RxGpuError):   # no lists yet
            ix.search_knn_lists(cx, rows[0], 4, 10)
        ix.set_lists(lists)
        for qi in range(12):
            q = clustered(300 + qi, 1, d, 30)[0]
            if metric == 2:
                q, _ = oracle.normalize_copy(q)
            for nprobe, k in ((1, 10), (4, 10), (7, 100), (24, 33), (64, 5)):
                _, crow, ccnt = cx.search_knn(q, min(nprobe, nlist))
                probed = crow[0, :int(ccnt[0])]
                want_rows = np.sort(np.concatenate([lists[int(l)] for l in probed]))
                gd, gr, scanned = ix.search_knn_lists(cx, q, nprobe, k)
                assert scanned == want_rows.size
                if want_rows.size == 0:
                    assert gd.size == 0
                    continue
                wd, wr, wc = ix.search_knn_subset(q, k, want_rows)
                c = int(wc[0])
                assert np.array_equal(gr, wr[0, :c]) and np.array_equal(bits(gd), bits(wd[0, :c])), (metric, qi, nprobe, k)
        # (nprobe is clamped to the number of lists first; the 64-list limit of the device entry only matters above that)
        with pytest.raises(rxgpu.RxGpuError):   # a row that the index does not hold
            ix.set_lists([np.array([n], np.uint32)])
        ix.upload_rows(n - 1, rows[-1:], inv[-1:] if inv is not None else None)   # same count: lists stay valid
        ix.search_knn_lists(cx, rows[0] if metric != 2 else oracle.normalize_copy(rows[0])[0], 2, 3)
    assert kept.size <= n


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_c_abi_wide_probes_and_range_over_device_lists(rxgpu, oracle, metric):
    """More lists than the resident coarse search returns (nprobe 65..128: the wide fused select; above: list ids through the host, the
    lists themselves stay in HBM), and rxgpu_search_range_lists == rxgpu_search_range_subset over the union of the probed lists."""
    n, d, nlist = 9000, 24, 300
    rng = np.random.default_rng(400 + metric)
    rows = clustered(17, n, d, 40)
    cents = clustered(18, nlist, d, 40)
    owner = rng.integers(0, nlist, n)
    lists = [np.flatnonzero(owner == l).astype(np.uint32) for l in range(nlist)]
    inv = oracle.l2_modules(rows) if metric == 2 else None
    cinv = oracle.l2_modules(cents) if metric == 2 else None
    with rxgpu.VectorIndex(metric, d, n) as ix, rxgpu.VectorIndex(metric, d, nlist) as cx:
        ix.upload_rows(0, rows, inv)
        cx.upload_rows(0, cents, cinv)
        ix.set_lists(lists)
        for qi in range(6):
            q = clustered(500 + qi, 1, d, 40)[0]
            if metric == 2:
                q, _ = oracle.normalize_copy(q)
            for nprobe, k in ((3, 10), (64, 10), (65, 20), (100, 7), (128, 128), (129, 10), (250, 50), (300, 10), (1000, 10)):
                np_eff = min(nprobe, nlist)
                _, crow, ccnt = cx.search_knn(q, np_eff)
                probed = crow[0, :int(ccnt[0])]
                want_rows = np.sort(np.concatenate([lists[int(l)] for l in probed]))
                gd, gr, scanned = ix.search_knn_lists(cx, q, nprobe, k)
                assert scanned == want_rows.size, (nprobe, scanned, want_rows.size)
                wd, wr, wc = ix.search_knn_subset(q, k, want_rows)
                c = int(wc[0])
                assert np.array_equal(gr, wr[0, :c]) and np.array_equal(bits(gd), bits(wd[0, :c])), (metric, qi, nprobe, k)
                # range: a radius that takes the k best of the probed rows
                if c >= 2 and wd[0, c - 1] > wd[0, 0]:
                    radius = float(wd[0, c - 1])
                    rd, rr, rscanned = ix.search_range_lists(cx, q, nprobe, radius, inclusive=False, cap=4)   # cap 4: the overflow protocol too
                    sd, sr = ix.search_range_subset(q, radius, want_rows, inclusive=False)
                    assert rscanned == want_rows.size
                    assert np.array_equal(rr, sr) and np.array_equal(bits(rd), bits(sd)), (metric, qi, nprobe)
                    assert rr.size >= 1 and np.all(rd < radius)
                    rd2, rr2, _ = ix.search_range_lists(cx, q, nprobe, radius, inclusive=True)
                    assert rr2.size > rr.size


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_ivf_batch_and_wide_probe_searches_equal_single_searches(hostapi, oracle, metric):
    """GpuIvfFlat::SearchBatch (queries side by side on several streams) == the same queries one by one; nprobe beyond 64 / 128 lists
    and RangeSearch go through the device lists and still equal the exact search over ProbedRows."""
    n, d, nlist = 8000, 32, 200
    rows = clustered(27, n, d, 25)
    ivf = hostapi.GpuIvfFlat(metric, d, nlist)
    ivf.add_with_ids(rows, np.arange(n, dtype=np.int64) * 3 + 1)
    ivf.train()
    qs = clustered(28, 24, d, 25)
    for nprobe in (4, 100, 180):
        bd, bl = ivf.search_batch(qs, 10, nprobe=nprobe)
        for i in range(qs.shape[0]):
            sd, sl = ivf.search(qs[i], 10, nprobe=nprobe)
            assert np.array_equal(bl[i], sl) and np.array_equal(bits(bd[i]), bits(sd)), (metric, nprobe, i)
        # the range search over the same probe equals a filter of a wide KNN over the probed rows
        q = qs[0]
        cand = ivf.probed_rows(q, nprobe)
        kd, kl = ivf.search(q, 40, nprobe=nprobe)
        valid = kl >= 0
        kd, kl = kd[valid], kl[valid]
        if kd.size >= 12 and kd[10] != kd[11]:
            radius = float(kd[11])
            rd, rl = ivf.range_search(q, radius, nprobe=nprobe)
            inside = kd < radius if metric == 0 else kd > radius
            assert set(rl.tolist()) == set(kl[inside].tolist()), (metric, nprobe, rd.size, int(inside.sum()))
        assert cand.size > 0
    ivf.close()
