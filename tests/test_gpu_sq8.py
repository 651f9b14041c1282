"""-m gpu: the SQ8 (uint8) HNSW search on the device — batch_distances_sq8 in hnsw_search.hip behind rxgpu_hnsw_attach_sq8 /
rxgpu_hnsw_search_knn_sq8 and behind GpuHnswMap::Quantize.  Bar: the labels and the distance BITS of the reference's quantised engine
(HierarchicalNSWImpl<uint8_t>): tests/golden/sq8.npz holds graphs, codes and SearchKnn results exported from the real engine; the C
restatement (oracle/oracle_sq8.c + oracle_hnsw.c, pinned against that engine in tests/test_sq8_oracle.py) covers everything else."""
from pathlib import Path

import numpy as np
import pytest

from .conftest import make_corpus

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def as_sorted_pairs(dist, ids):
    order = np.lexsort((ids, dist))
    return dist[order], ids[order]


@pytest.fixture(scope="module")
def sq8(oracle):
    from oracle.pyoracle import Sq8Oracle
    return Sq8Oracle(oracle)


@pytest.mark.parametrize("metric", [0, 2])
def test_c_abi_sq8_search_matches_golden_engine_results(rxgpu, oracle, metric):
    z = np.load(G / "sq8.npz")
    key = f"hq_m{metric}"
    n, dim, M, maxM0, maxlevel, entry, num_deleted = (int(x) for x in z[key + "_meta"])
    g = dict(metric=metric, n=n, dim=dim, M=M, maxM0=maxM0, maxlevel=maxlevel, entry=entry, num_deleted=num_deleted,
             links0=z[key + "_links0"], upper_off=z[key + "_upper_off"], upper=z[key + "_upper"], deleted=z[key + "_deleted"])
    min_q, max_q, alpha, alpha_2, delta = (float(x) for x in z[key + "_params"])
    rows, labels = z[key + "_rows"], z[key + "_labels"]
    inv = oracle.l2_modules(rows) if metric == 2 else None
    from reindexer_amd.hostapi import sq8_quantize
    with rxgpu.VectorIndex(metric, dim, n) as ix:
        ix.upload_rows(0, rows, inv)
        ix.hnsw_attach_graph(g)
        ix.hnsw_attach_sq8(z[key + "_codes"], z[key + "_corr"], alpha_2)
        queries = z[key + "_queries"]
        qc, qo, qn = [], [], []
        for i, q in enumerate(queries):
            coef = np.float32(1.0) / np.float32(z[key + "_qnorms"][i]) if metric == 2 else np.float32(1.0)
            c, o, _ = sq8_quantize(metric, min_q, max_q, q, float(np.float32(1.0) / coef))   # prepareData: norm = 1.f / normCoef
            qc.append(c), qo.append(o), qn.append(coef)
        dist, row, cnt = ix.hnsw_search_knn_sq8(np.stack(qc), np.array(qo, np.float32), np.array(qn, np.float32), 10, 32)
        for i in range(len(queries)):
            c = int(cnt[i])
            gd, gl = as_sorted_pairs(dist[i, :c], labels[row[i, :c]])
            wl, wd = z[key + "_res_label"][i], z[key + "_res_dist"][i]
            wd, wl = as_sorted_pairs(wd[:len(gl)], wl[:len(gl)])
            assert np.array_equal(gl, wl), (metric, i)
            assert np.array_equal(bits(gd), bits(wd)), (metric, i)


def _clique(n, M=16):
    """A level-0 clique of n <= 2M + 1 nodes: a search with ef = k = n returns every node with its distance."""
    maxM0 = 2 * M
    links0 = np.zeros((n, 1 + maxM0), np.uint32)
    for i in range(n):
        others = [j for j in range(n) if j != i]
        links0[i, 0] = len(others)
        links0[i, 1:1 + len(others)] = others
    return dict(links0=links0, upper_off=np.zeros(n + 1, np.uint64), upper=np.zeros(0, np.uint32), deleted=np.zeros(n, np.uint8), M=M, maxM0=maxM0,
                maxlevel=0, entry=0, num_deleted=0)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("d", [1, 2, 3, 31, 63, 64, 65, 100, 127, 128, 130, 257, 384, 512, 768, 1000, 1024, 1536, 4096])
def test_sq8_distance_bits_all_dims(rxgpu, oracle, sq8, metric, d):
    """Every distance of the device equals DistCalculator<uint8_t>::operator()(query, row, id) bit for bit — ragged dims (tails, rows that
    do not start on a word boundary), saturated codes (sums beyond 2^24, where the float reduction order shows) and narrow ranges."""
    n = 33
    rng = np.random.default_rng(1000 * metric + d)
    g = _clique(n)
    for style in ("uniform", "extreme", "narrow"):
        if style == "uniform":
            codes, qcodes = rng.integers(0, 256, (n, d)), rng.integers(0, 256, (4, d))
        elif style == "extreme":
            codes, qcodes = rng.choice([0, 255, 254], (n, d)), rng.choice([0, 255], (4, d))
        else:
            codes, qcodes = rng.integers(100, 140, (n, d)), rng.integers(100, 140, (4, d))
        codes, qcodes = codes.astype(np.uint8), qcodes.astype(np.uint8)
        corr = rng.normal(0, 3, n).astype(np.float32)
        qcorr = rng.normal(0, 3, 4).astype(np.float32)
        qnorm = rng.uniform(0.5, 2, 4).astype(np.float32) if metric == 2 else np.ones(4, np.float32)
        inv = rng.uniform(0.2, 3, n).astype(np.float32) if metric == 2 else None
        alpha_2 = float(np.float32(0.0123) ** 2)
        with rxgpu.VectorIndex(metric, d, n) as ix:
            ix.upload_rows(0, np.zeros((n, d), np.float32), inv)
            ix.hnsw_attach_graph(g)
            ix.hnsw_attach_sq8(codes, corr, alpha_2)
            dist, row, cnt = ix.hnsw_search_knn_sq8(qcodes, qcorr, qnorm, n, n)
            for qi in range(4):
                assert cnt[qi] == n
                got = np.empty(n, np.float32)
                got[row[qi]] = dist[qi]
                want = sq8.dist_query_many(metric, dict(alpha_2=alpha_2), qcodes[qi], qcorr[qi], codes, corr, inv)
                want = (qnorm[qi] * want).astype(np.float32)
                assert np.array_equal(bits(got), bits(want)), (metric, d, style, qi)


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_quantised_map_equals_restated_quantised_engine(rxgpu, oracle, sq8, metric):
    """GpuHnswMap::Quantize: labels and distance bits of the restated HierarchicalNSWImpl<uint8_t> (itself equal to the real quantised engine,
    tests/test_sq8_oracle.py) on the Map's own graph — with deletes, then with points added after the quantisation."""
    from oracle.pyoracle import oracle_hnsw_search_knn_sq8
    from reindexer_amd import hostapi
    n, d, extra = 6000, 96, 300
    rows = make_corpus(71, n + extra, d)
    labels = (np.arange(n + extra, dtype=np.uint64) << np.uint64(32)) | np.uint64(5)
    m = hostapi.GpuHnswMap(metric, d, n + extra, M=16, ef_construction=100)
    m.add(rows[:n], labels[:n])
    for lab in labels[np.random.default_rng(3).choice(n, 150, replace=False)]:
        m.mark_delete(lab)
    min_q, max_q = float(np.quantile(rows[:n], 0.005)), float(np.quantile(rows[:n], 0.995))
    assert not m.is_quantized
    m.quantize(min_q, max_q)
    assert m.is_quantized
    p = sq8.params(min_q, max_q, d)
    for phase in range(2):
        cnt = n if phase == 0 else n + extra
        if phase:
            m.add(rows[n:], labels[n:])
        g = m.export_graph(with_views=True)   # internal-id order: added points recycle the slots of deleted ones
        vecs = np.array(g["vectors"])
        assert vecs.shape[0] == g["n"] <= cnt
        stored = [sq8.quantize(metric, p, x) for x in vecs]
        sq = dict(min_q=p["min_q"], alpha=p["alpha"], alpha_2=p["alpha_2"], delta=p["delta"], codes=np.stack([c for c, _ in stored]),
                  corr=np.array([o for _, o in stored], np.float32))
        inv = oracle.l2_modules(vecs) if metric == 2 else None
        for qi in range(25):
            q = make_corpus(1700 + qi, 1, d)[0]
            norm = None
            if metric == 2:
                q, k_ = oracle.normalize_copy(q)
                norm = float(np.float32(1.0) / np.float32(k_))
            for k, ef in ((10, 64), (1, 10), (50, 0)):
                wd, wl = oracle_hnsw_search_knn_sq8(oracle, g, sq, q, k, ef, inv, norm)
                gd, gl = m.search_knn_norm(q, k, ef, norm)
                assert np.array_equal(gl, wl), (metric, phase, qi, k, ef)
                assert np.array_equal(bits(gd), bits(wd))
    if metric == 2:
        with pytest.raises(RuntimeError, match="Norm is required"):
            m.search_knn_norm(rows[0], 5, 10, None)
    m.close()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_quantised_map_range_and_streaming_equal_the_reference_quantised_engine(rxgpu, ref, oracle, metric):
    """After Quantize() the Map answers EVERYTHING the reference's quantised engine does (HierarchicalNSWImpl<uint8_t>): SearchRange — the
    ef-search and the closure over codes, both on the device — and whole streaming sessions over codes, compared with the REAL engine built
    from the same inserts (the host builder is link-for-link the reference's; the quantisation range is the one the reference sampled)."""
    from oracle.pyoracle import RefHnsw, RefHnswQ
    from reindexer_amd import hostapi
    n, d = 4000, 64
    rows = make_corpus(83 + metric, n, d)
    labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(2)
    rf = RefHnsw(ref, metric, d, n, M=12, ef_construction=80)
    rf.add(rows, labels)
    m = hostapi.GpuHnswMap(metric, d, n, M=12, ef_construction=80)
    m.add(rows, labels)
    dead = labels[np.random.default_rng(4).choice(n, 120, replace=False)]
    for lab in dead:
        rf.mark_delete(lab)
        m.mark_delete(lab)
    rq = RefHnswQ(rf, sample_size=n)
    prm = rq.export()
    m.quantize(float(prm["min_q"]), float(prm["max_q"]))

    def pairs(dist, lab):
        o = np.lexsort((lab, dist))
        return dist[o], lab[o]

    for qi in range(12):
        q = make_corpus(2300 + qi, 1, d)[0]
        norm = None
        if metric == 2:
            q, k_ = oracle.normalize_copy(q)
            norm = float(np.float32(1.0) / np.float32(k_))
        wd, wl = rq.search_knn(q, 40, 64, norm)
        gd, gl = m.search_knn_norm(q, 40, 64, norm)
        assert np.array_equal(gl, wl) and np.array_equal(bits(gd), bits(wd)), (metric, qi, "knn")      # same graph, same codes
        # radii around the 10th / 40th hit: a handful .. a few hundred results, expansion several levels deep
        for radius in (float(wd[9]), float(wd[39]), float(wd[39]) + abs(float(wd[39])) * 0.05 + 0.02):
            for ef in (16, 64):
                rd, rl = rq.search_range(q, radius, ef, norm)
                sd, sl = m.search_range(q, radius, ef, norm=norm)
                a, b = pairs(sd, sl), pairs(rd, rl)
                assert np.array_equal(a[1], b[1]) and np.array_equal(bits(a[0]), bits(b[0])), (metric, qi, radius, ef, len(sd), len(rd))
        # streaming sessions over the codes: batch for batch, exhausted at the same call
        for ef, plan in ((0, [10] * 5), (32, [7, 60, 1, 200])):
            gs, ws = m.stream(q, ef, norm=norm), rq.stream(q, ef, norm)
            for bsz in plan:
                g1, w1 = gs.next(bsz), ws.next(bsz)
                assert g1[2] == w1[2] and len(g1[0]) == len(w1[0]), (metric, qi, ef, bsz)
                a, b = pairs(g1[0], g1[1]), pairs(w1[0], w1[1])
                assert np.array_equal(a[1], b[1]) and np.array_equal(bits(a[0]), bits(b[0])), (metric, qi, ef, bsz)
            gs.close()
            ws.close()
    rq.close()
    rf.close()
    m.close()


def _srand(seed):
    import ctypes
    ctypes.CDLL(None).srand(int(seed))


@pytest.mark.parametrize("metric", [0, 2])
def test_quantize_through_the_map_with_the_reference_config(rxgpu, ref, oracle, metric):
    """HnswIndexBase::Quantize() / SwitchMapOnQuantized() end to end through the Map (what patch 0001 calls): GpuHnswMap::Quantize(config)
    samples its OWN rows the way QuantizingParams does and equals the reference's parameters bit for bit (same srand() state); until the
    switch the Map answers over the float rows; after it, like the reference's quantised engine; points added LATER are quantised with the
    same parameters and only their codes travel (rxgpu_hnsw_upload_sq8_rows); the ANN cache carries the parameters and a load with
    LoadWithQuantizer brings the Map back quantised."""
    from oracle.pyoracle import RefHnsw, RefHnswQ
    from reindexer_amd import hostapi
    n0, n1, d = 3000, 3400, 48
    rows = make_corpus(301 + metric, n1, d)
    labels = (np.arange(n1, dtype=np.uint64) << np.uint64(32)) | np.uint64(1)
    rf = RefHnsw(ref, metric, d, n1, M=10, ef_construction=60)
    rf.add(rows[:n0], labels[:n0])
    m = hostapi.GpuHnswMap(metric, d, n1, M=10, ef_construction=60)
    m.add(rows[:n0], labels[:n0])
    queries = make_corpus(302, 10, d)

    def ask(engine, q, quantised):
        norm = None
        if metric == 2:
            q, k_ = oracle.normalize_copy(q)
            norm = float(np.float32(1.0) / np.float32(k_)) if quantised else None
        return engine.search_knn(q, 20, 64, norm) if isinstance(engine, RefHnswQ) else engine.search_knn_norm(q, 20, 64, norm)

    float_answers = [ask(m, q, False) for q in queries]
    _srand(777)
    rq = RefHnswQ(rf, sample_size=1000)          # the reservoir is active: 3000 rows, 1000 sampled
    want = rq.export()
    _srand(777)
    m.quantize_config(sample_size=1000, switch=False)
    assert not m.is_quantized                     # pending: readers stay on the float rows (hnsw.cc:108-114)
    for q, fa in zip(queries, float_answers):
        ga = ask(m, q, False)
        assert np.array_equal(ga[1], fa[1]) and np.array_equal(bits(ga[0]), bits(fa[0]))
    m.switch_on_quantized()
    assert m.is_quantized
    got = m.quantizing_params
    wantp = np.array([want["min_q"], want["max_q"], want["alpha"], want["alpha_2"], want["delta"]], np.float32)
    assert np.array_equal(bits(got), bits(wantp)), (got, wantp)
    for q in queries:
        wd, wl = ask(rq, q, True)
        gd, gl = ask(m, q, True)
        assert np.array_equal(gl, wl) and np.array_equal(bits(gd), bits(wd))
    # points added to the quantised graph (addPoint with a quantizer): the reference's quantised engine takes them too
    rq.close()
    rf.add(rows[n0:], labels[n0:])
    m.add(rows[n0:], labels[n0:])
    _srand(778)
    rq2 = RefHnswQ(rf, sample_size=n1)            # a reference engine over all rows, quantised with the Map's (earlier) parameters' range
    rq2.close()
    # (the reference cannot add to its quantised copy through this shim: check the Map against itself re-quantised from scratch instead)
    fresh = hostapi.GpuHnswMap(metric, d, n1, M=10, ef_construction=60)
    fresh.add(rows[:n0], labels[:n0])
    fresh.add(rows[n0:], labels[n0:])
    fresh.quantize(float(got[0]), float(got[1]))  # whole table built in one go with the same range
    for q in queries:
        a, b = ask(m, q, True), ask(fresh, q, True)
        assert np.array_equal(a[1], b[1]) and np.array_equal(bits(a[0]), bits(b[0]))
    assert any((a >> np.uint64(32)) >= n0 for q in queries for a in ask(m, q, True)[1]), "the added rows must be reachable"
    # the ANN cache of a quantised Map: flag 1 + QuantizingParams; loaded with LoadWithQuantizer it is quantised again, without it a float graph
    blob = m.save_index()
    back = hostapi.GpuHnswMap(metric, d, n1, M=10, ef_construction=60)
    back.load_index(blob, labels, rows, with_quantizer=True)
    assert back.is_quantized and np.array_equal(bits(back.quantizing_params), bits(got))
    for q in queries:
        a, b = ask(m, q, True), ask(back, q, True)
        assert np.array_equal(a[1], b[1]) and np.array_equal(bits(a[0]), bits(b[0]))
    plain = hostapi.GpuHnswMap(metric, d, n1, M=10, ef_construction=60)
    plain.load_index(blob, labels, rows, with_quantizer=False)
    assert not plain.is_quantized
    for x in (m, fresh, back, plain):
        x.close()
    rf.close()
