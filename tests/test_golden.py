"""Golden vectors generated from the real reference engines (tests/golden/make_golden.py).
CPU part: pins the plain-C oracle.  GPU part (-m gpu): pins the HIP kernels through the C-ABI."""
from pathlib import Path

import numpy as np
import pytest

from .conftest import apply_swap_deletes

G = Path(__file__).resolve().parent / "golden"
DIMS = [1, 7, 16, 33, 64, 100, 128, 200, 512, 768, 1000]


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def gdist():
    return np.load(G / "distances.npz")


@pytest.fixture(scope="module")
def gbf():
    return np.load(G / "bruteforce.npz")


def victims_labels(g, name):
    return [g[f"{name}_labels"][v] for v in g[f"{name}_victims"]]


def live_state(g, name):
    rows, labels = g[f"{name}_rows"], g[f"{name}_labels"]
    cnt_rows, cnt_labels = rows.copy(), labels.copy()
    cnt = rows.shape[0]
    for lab in victims_labels(g, name):
        pos = int(np.nonzero(cnt_labels[:cnt] == lab)[0][0])
        if pos + 1 != cnt:
            cnt_rows[pos] = cnt_rows[cnt - 1]
            cnt_labels[pos] = cnt_labels[cnt - 1]
        cnt -= 1
    return cnt_rows[:cnt].copy(), cnt_labels[:cnt].copy()


# ------------------------------------------------------------------------------------------- CPU: oracle
@pytest.mark.parametrize("d", DIMS)
def test_oracle_distances_match_golden(oracle, gdist, d):
    rows, q = gdist[f"rows_{d}"], gdist[f"q_{d}"]
    assert np.array_equal(bits(oracle.dist_many(0, q, rows)), bits(gdist[f"l2_{d}"]))
    assert np.array_equal(bits(-oracle.dist_many(1, q, rows)), bits(gdist[f"ip_{d}"]))
    got = oracle.l2_modules(gdist[f"norm_in_{d}"])
    assert np.array_equal(bits(got), bits(gdist[f"norm_{d}"]))


@pytest.mark.parametrize("name", ["gauss", "ties"])
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_oracle_bruteforce_matches_golden(oracle, gbf, name, metric):
    rows, labels = live_state(gbf, name)
    n0 = gbf[f"{name}_rows"].shape[0]
    inv = oracle.l2_modules(rows) if metric == 2 else None
    queries = gbf[f"{name}_queries"]
    for qi in range(queries.shape[0]):
        q = queries[qi]
        if metric == 2:
            q, _ = oracle.normalize_copy(q)
        for k in (1, 10, 64, 100, n0):
            gd, gl = oracle.bf_search_knn(metric, rows, labels, inv, q, k)
            assert np.array_equal(gl, gbf[f"{name}_m{metric}_q{qi}_k{k}_label"]), (qi, k)
            assert np.array_equal(bits(gd), bits(gbf[f"{name}_m{metric}_q{qi}_k{k}_dist"]))
        radius = float(gbf[f"{name}_m{metric}_q{qi}_radius"][0])
        gd, gl = oracle.bf_search_range(metric, rows, labels, inv, q, radius)
        assert np.array_equal(gl, gbf[f"{name}_m{metric}_q{qi}_range_label"])
        assert np.array_equal(bits(gd), bits(gbf[f"{name}_m{metric}_q{qi}_range_dist"]))


# ------------------------------------------------------------------------------------------- GPU: HIP kernels
@pytest.mark.gpu
@pytest.mark.parametrize("d", DIMS)
def test_gpu_distances_match_golden(rxgpu, gdist, d):
    rows, q = gdist[f"rows_{d}"], gdist[f"q_{d}"]
    ids = np.arange(rows.shape[0], dtype=np.uint32)
    with rxgpu.VectorIndex("l2", d, rows.shape[0]) as ix:
        ix.upload_rows(0, rows)
        assert np.array_equal(bits(ix.distances(q, ids)), bits(gdist[f"l2_{d}"]))
    with rxgpu.VectorIndex("ip", d, rows.shape[0]) as ix:
        ix.upload_rows(0, rows)
        assert np.array_equal(bits(-ix.distances(q, ids)), bits(gdist[f"ip_{d}"]))


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_gpu_bruteforce_matches_golden_gauss(rxgpu, oracle, gbf, metric):
    """No distance ties in the gaussian case => (dist,row) order == (dist,label) order: rows map 1:1 to labels."""
    rows, labels = live_state(gbf, "gauss")
    n0 = gbf["gauss_rows"].shape[0]
    inv = oracle.l2_modules(rows) if metric == 2 else None
    queries = gbf["gauss_queries"]
    with rxgpu.VectorIndex(metric, rows.shape[1], rows.shape[0]) as ix:
        ix.upload_rows(0, rows, inv)
        for qi in range(queries.shape[0]):
            q = queries[qi]
            if metric == 2:
                q, _ = oracle.normalize_copy(q)
            for k in (1, 10, 64, 100, n0):
                dist, row, cnt = ix.search_knn(q, k)
                c = int(cnt[0])
                assert c == min(k, rows.shape[0])
                assert np.array_equal(labels[row[0, :c]], gbf[f"gauss_m{metric}_q{qi}_k{k}_label"]), (qi, k)
                assert np.array_equal(bits(dist[0, :c]), bits(gbf[f"gauss_m{metric}_q{qi}_k{k}_dist"]))
            radius = float(gbf[f"gauss_m{metric}_q{qi}_radius"][0])
            rd, rr = ix.search_range(q, radius)
            assert np.array_equal(labels[rr], gbf[f"gauss_m{metric}_q{qi}_range_label"])
            assert np.array_equal(bits(rd), bits(gbf[f"gauss_m{metric}_q{qi}_range_dist"]))


# ------------------------------------------------------------------------------------------- HNSW (golden graph + results)
def golden_hnsw_graph(oracle, phase):
    z = np.load(G / "hnsw.npz")
    n, d = z["rows"].shape
    deleted = np.zeros(n, np.uint8)
    if phase:
        deleted[z["victims"]] = 1
    g = dict(metric=int(z["metric"]), n=n, dim=d, M=int(z["M"]), maxM0=2 * int(z["M"]), maxlevel=int(z["maxlevel"]), entry=int(z["entry"]),
             num_deleted=int(deleted.sum()), links0=z["links0"], upper_off=z["upper_off"], upper=z["upper"], levels=z["levels"],
             labels=z["labels"], deleted=deleted, vectors=np.ascontiguousarray(z["rows"]))
    return z, g


@pytest.mark.parametrize("phase", [0, 1])
def test_oracle_hnsw_search_matches_golden(oracle, phase):
    from oracle.pyoracle import oracle_hnsw_search_knn
    z, g = golden_hnsw_graph(oracle, phase)
    inv = oracle.l2_modules(g["vectors"])
    for qi in range(z["queries"].shape[0]):
        qn, _ = oracle.normalize_copy(z["queries"][qi])
        for k, ef in ((10, 128), (10, 10), (1, 0), (40, 64)):
            gd, gl = oracle_hnsw_search_knn(oracle, g, qn, k, ef, inv)
            assert np.array_equal(gl, z[f"p{phase}_q{qi}_k{k}_ef{ef}_label"]), (phase, qi, k, ef)
            assert np.array_equal(bits(gd), bits(z[f"p{phase}_q{qi}_k{k}_ef{ef}_dist"]))


@pytest.mark.parametrize("phase", [0, 1])
def test_oracle_hnsw_streaming_matches_golden(oracle, phase):
    """Streaming sessions recorded from the REAL engine (tests/golden/make_golden.py): the restatement must hand out the same batches and
    flip `exhausted` at the same call — pins oracle_hnsw.c's streaming half where oracle/_ref is absent."""
    from oracle.pyoracle import OracleHnswStream
    plans = [(0, [10, 10, 10]), (16, [5, 40, 1, 300]), (3, [1, 1, 2, 2000])]
    z, g = golden_hnsw_graph(oracle, phase)
    inv = oracle.l2_modules(g["vectors"])
    for qi in range(4):
        qn, _ = oracle.normalize_copy(z["queries"][qi])
        for si, (sef, plan) in enumerate(plans):
            s = OracleHnswStream(oracle, g, qn, sef, inv)
            for bi, b in enumerate(plan):
                dd, ll, ex = s.next(b)
                o = np.lexsort((ll, dd))
                assert np.array_equal(ll[o], z[f"s{phase}_q{qi}_p{si}_b{bi}_label"]), (phase, qi, si, bi)
                assert np.array_equal(bits(dd[o]), bits(z[f"s{phase}_q{qi}_p{si}_b{bi}_dist"]))
                assert ex == bool(z[f"s{phase}_q{qi}_p{si}_b{bi}_exhausted"])
            s.close()
