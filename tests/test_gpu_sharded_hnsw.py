"""-m gpu: HNSW over a DEVICE LIST (SURVEY §8(e) "HNSW": per-shard independent graphs + the same all-gather merge as brute force).

C-ABI: every shard of an rxgpu_index_create_sharded handle holds the graph of ITS rows (rows / graph attached through the
rxgpu_index_shard(h, s) handles); rxgpu_hnsw_search_knn on the sharded handle runs every shard's search at once, the per-shard results stay in
HBM and meet in one ncclAllGather + the (dist, global row) merge kernel.  Map: GpuHnswMap over a device list — the shape the reference's
factory reaches with RX_GPU_VECTOR_INDEXES=0-7 (rx_seam.h).  The 1-GPU test box lists device 0 several times: the code path is the multi-GPU one.

Bar: the merged result = the k best of the union of the per-shard ENGINE results — each shard pinned to the restated engine and, where
oracle/_ref travels, to the reference's own HierarchicalNSW built over the shard's points — labels and distance bits; recall vs exact brute
force >= the single graph's at equal ef."""
import numpy as np
import pytest

from .conftest import make_corpus

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def hostapi(rxgpu):
    from reindexer_amd import hostapi as h
    h.lib()
    return h


def expected_merge(per_shard, k):
    """per_shard: [(dist, global_row)] arrays of every shard -> the k best under (dist, global row)."""
    d = np.concatenate([p[0] for p in per_shard])
    r = np.concatenate([p[1] for p in per_shard])
    order = np.lexsort((r, d))[:k]
    return d[order], r[order]


@pytest.mark.parametrize("mode", ["rccl", "host"])
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_c_abi_sharded_hnsw_is_the_merge_of_the_per_shard_engines(rxgpu, hostapi, oracle, monkeypatch, mode, metric):
    from oracle.pyoracle import oracle_hnsw_search_knn
    if mode == "host":
        monkeypatch.setenv("RXGPU_SHARD_MERGE", "host")
    else:
        monkeypatch.delenv("RXGPU_SHARD_MERGE", raising=False)
    n, d = 5000, 48
    rows = make_corpus(70 + metric, n, d)
    # (4, 1400): shard 1 partial, shards 2 and 3 EMPTY; (4, -30): shard 1 holds 30 rows — fewer than k, so its lists come back with a
    # stride of 30 (the shard's search clamps k to its count) and the batched host merge must read them that way
    for shards, fill in ((2, n), (3, n), (8, n), (4, 1400), (4, -30)):
        with rxgpu.ShardedVectorIndex(metric, d, n, [0] * shards) as sx:
            assert sx.merge_mode == mode
            sr = sx.shard_rows
            if fill < 0:
                fill = sr - fill
            graphs = []
            for s in range(shards):
                lo, hi = s * sr, min(fill, (s + 1) * sr)
                if lo >= hi:
                    graphs.append(None)
                    continue
                part = rows[lo:hi]
                m = hostapi.GpuHnswMap(metric, d, hi - lo, M=8, ef_construction=80)
                m.add(part, np.arange(lo, hi, dtype=np.uint64))   # label = GLOBAL row
                g = m.export_graph()
                g["vectors"] = part
                m.close()
                inv = oracle.l2_modules(part) if metric == 2 else None
                view = sx.shard(s)
                view.upload_rows(0, part, inv)
                view.hnsw_attach_graph(g)
                graphs.append((g, inv))
            q = make_corpus(400 + metric, 7, d)
            if metric == 2:
                q = np.stack([oracle.normalize_copy(v)[0] for v in q])
            before = sx.collectives
            calls = 0
            for k, ef in ((10, 64), (1, 0), (40, 40), (64, 100)):
                for qs in (q[:1], q):
                    gd, gr, gc = sx.hnsw_search_knn(qs, k, ef)
                    calls += 1
                    for qi in range(qs.shape[0]):
                        per = []
                        for g in graphs:
                            if g is None:
                                continue
                            wd, wl = oracle_hnsw_search_knn(oracle, g[0], qs[qi], k, ef, g[1])
                            per.append((wd, wl.astype(np.uint32)))
                        wd, wr = expected_merge(per, k)
                        c = int(gc[qi])
                        assert c == wd.shape[0], (metric, shards, fill, k, ef, qi)
                        assert np.array_equal(gr[qi, :c], wr) and np.array_equal(bits(gd[qi, :c]), bits(wd)), (metric, shards, fill, k, ef, qi)
            assert sx.collectives - before == (calls if mode == "rccl" else 0)
            # k > 64: host merge in either mode
            gd, gr, gc = sx.hnsw_search_knn(q[:2], 100, 128)
            for qi in range(2):
                per = [(lambda r: (r[0], r[1].astype(np.uint32)))(oracle_hnsw_search_knn(oracle, g[0], q[qi], 100, 128, g[1])) for g in graphs if g]
                wd, wr = expected_merge(per, 100)
                c = int(gc[qi])
                assert c == wd.shape[0] and np.array_equal(gr[qi, :c], wr) and np.array_equal(bits(gd[qi, :c]), bits(wd))
            evals, hops = sx.hnsw_read_stats()
            assert evals > 0 and hops > 0


def test_sharded_handle_refuses_what_is_per_graph(rxgpu):
    from reindexer_amd import capi
    with rxgpu.ShardedVectorIndex(1, 16, 256, [0, 0]) as sx:
        g = dict(links0=np.zeros((1, 17), np.uint32), upper_off=np.zeros(2, np.uint64), upper=np.zeros(0, np.uint32), deleted=np.zeros(1, np.uint8),
                 M=8, maxM0=16, maxlevel=0, entry=0, num_deleted=0)
        with pytest.raises(capi.RxGpuError, match="one graph per shard"):
            sx.hnsw_attach_graph(g)
        d, r, c = sx.hnsw_search_knn(np.zeros((2, 16), np.float32), 5, 10)    # every shard empty: no hits, no error
        assert c.tolist() == [0, 0]


@pytest.mark.parametrize("metric", [0, 2])
def test_map_over_a_device_list_equals_the_per_shard_engines(hostapi, oracle, metric):
    """GpuHnswMap(devices=[0, 0, 0]): points fill the shards in insertion order; every shard's graph is what the engine builds over its points
    (restated engine on the exported shard graph); SearchKnn / SearchRange = the merge of the shard results; deletes and re-inserts route to the
    shard that holds the label; recall >= the single graph's."""
    from oracle.pyoracle import oracle_hnsw_search_knn
    n, d, k = 4500, 40, 10
    rows = make_corpus(81 + metric, n, d)
    rng = np.random.default_rng(metric)
    labels = (rng.permutation(n).astype(np.uint64) << np.uint64(32)) | np.uint64(5)
    many = hostapi.GpuHnswMap(metric, d, n, M=8, ef_construction=80, devices=[0, 0, 0])
    one = hostapi.GpuHnswMap(metric, d, n, M=8, ef_construction=80)
    assert many.shard_count == 3 and many.shard_rows == 1504
    many.add(rows, labels)
    one.add(rows, labels)
    assert many.count == n and [many.shard(s).count for s in range(3)] == [1504, 1504, n - 3008]
    inv = oracle.l2_modules(rows) if metric == 2 else None
    for phase in range(3):
        if phase == 1:
            for lab in labels[rng.choice(n, 300, replace=False)]:
                many.mark_delete(lab)
                one.mark_delete(lab)
            assert many.deleted_count == 300
        if phase == 2:   # updates in place: the label stays in its shard
            upd = rng.choice(n, 50, replace=False)
            rows[upd] = make_corpus(999, 50, d)
            many.add(rows[upd], labels[upd])
            one.add(rows[upd], labels[upd])
            inv = oracle.l2_modules(rows) if metric == 2 else None
            assert many.count == n
        graphs = []
        for s in range(3):
            g = many.shard(s).export_graph(with_views=True)
            graphs.append(g)
        hits_many = hits_one = total = 0
        for qi in range(20):
            q = make_corpus(700 + qi, 1, d)[0]
            if metric == 2:
                q, _ = oracle.normalize_copy(q)
            for kk, ef in ((k, 64), (3, 0)):
                per = [oracle_hnsw_search_knn(oracle, g, q, kk, ef, g["inv_norms"]) for g in graphs]
                # the Map merges under (dist, global row); labels of one shard keep the shard's internal order only through the row — compare as sets
                # of (dist bits, label), ties at the kk-th place excluded by the generic data
                wd = np.concatenate([p[0] for p in per])
                wl = np.concatenate([p[1] for p in per])
                order = np.lexsort((wl, wd))[:kk]
                gd, gl = many.search_knn(q, kk, ef)
                o2 = np.lexsort((gl, gd))
                assert np.array_equal(gl[o2], wl[order]) and np.array_equal(bits(gd[o2]), bits(wd[order])), (metric, phase, qi, kk, ef)
            alld = oracle.dist_many(metric, q, rows, inv)
            live = np.ones(n, bool)
            for g in graphs:
                dead = g["labels"][g["deleted"] != 0]
                live[np.isin(labels, dead)] = False
            alld[~live] = np.inf
            truth = set(labels[np.argsort(alld, kind="stable")[:k]].tolist())
            hits_many += len(truth & set(many.search_knn(q, k, 64)[1].tolist()))
            hits_one += len(truth & set(one.search_knn(q, k, 64)[1].tolist()))
            total += k
            radius = float(np.sort(alld)[25])
            gd, gl = many.search_range(q, radius, 32)
            per = [many.shard(s).search_range(q, radius, 32) for s in range(3)]
            wl = np.concatenate([p[1] for p in per])
            assert sorted(gl.tolist()) == sorted(wl.tolist())
        assert hits_many >= hits_one, (hits_many, hits_one, total)   # recall >= the single graph's at equal ef
    # growth: every range grows, the device mirror is re-created, results stay
    q = make_corpus(5, 1, d)[0]
    if metric == 2:
        q, _ = oracle.normalize_copy(q)
    before = many.search_knn(q, k, 64)
    many.resize(2 * n)
    assert many.shard_rows == 3008
    after = many.search_knn(q, k, 64)
    assert np.array_equal(before[1], after[1]) and np.array_equal(bits(before[0]), bits(after[0]))
    extra = make_corpus(6, 200, d)
    live_before, s0_before = many.count - many.deleted_count, many.shard(0).count - many.shard(0).deleted_count
    many.add(extra, (np.arange(n, n + 200, dtype=np.uint64) << np.uint64(32)) | np.uint64(5))
    # the first range has room again: all 200 go there (into its delete-marked slots first, like addPoint with replace_deleted, then new rows)
    assert many.count - many.deleted_count == live_before + 200
    assert many.shard(0).count - many.shard(0).deleted_count == s0_before + 200
    gd, gl = many.search_knn(extra[17], 1, 32)
    assert gl[0] == ((n + 17) << 32 | 5)
    # copy-on-write clone
    c = many.clone(2 * n + 10)
    cd, cl = c.search_knn(q, k, 64)
    md, ml = many.search_knn(q, k, 64)
    assert np.array_equal(cl, ml) and np.array_equal(bits(cd), bits(md))
    c.close()
    many.close()
    one.close()


def test_clear_then_partial_refill_searches_only_what_is_there(hostapi, oracle):
    """Map::Clear over a device list, then inserts that land in shard 0 only: the other shards' DEVICE indexes must be empty too (they get
    no mirror call while they hold no points) — no stale rows of the first life of the Map may come back, under any label."""
    n, d, k = 3000, 32, 10
    rows = make_corpus(91, n, d)
    labels = np.arange(n, dtype=np.uint64) + np.uint64(1000)
    many = hostapi.GpuHnswMap(0, d, n, M=8, ef_construction=80, devices=[0, 0, 0])
    many.add(rows, labels)
    q = make_corpus(92, 5, d)
    for qi in range(5):
        gd, gl = many.search_knn(q[qi], k, 64)
        assert len(gl) == k
    many.clear()
    assert many.count == 0 and [many.shard(s).count for s in range(3)] == [0, 0, 0]
    few = 40   # all in shard 0
    new_rows = make_corpus(93, few, d)
    new_labels = np.arange(few, dtype=np.uint64) + np.uint64(50_000)
    many.add(new_rows, new_labels)
    assert [many.shard(s).count for s in range(3)] == [few, 0, 0]
    one = hostapi.GpuHnswMap(0, d, n, M=8, ef_construction=80)
    one.add(new_rows, new_labels)
    for qi in range(5):
        gd, gl = many.search_knn(q[qi], k, 64)
        wd, wl = one.search_knn(q[qi], k, 64)
        assert set(gl.tolist()) <= set(new_labels.tolist()), (qi, gl)
        assert np.array_equal(np.sort(gl), np.sort(wl)) and np.array_equal(np.sort(bits(gd)), np.sort(bits(wd)))
    # a label deleted and inserted again goes wherever there is room, not back to a slot that is gone
    many.mark_delete(int(new_labels[3]))
    many.add(new_rows[3:4], new_labels[3:4])
    gd, gl = many.search_knn(new_rows[3], 1, 32)
    assert gl.tolist() == [int(new_labels[3])]
    one.close()
    many.close()


def test_map_shards_equal_the_reference_engine_built_over_the_same_points(hostapi, ref, oracle):
    """Each shard pinned to the reference's own HierarchicalNSW (oracle/_ref) over the shard's points in insertion order: the sharded Map's
    answer = the (dist, label)-sorted union of what those engines return, cut at k."""
    from oracle.pyoracle import RefHnsw
    n, d, metric = 3000, 64, 1
    rows = make_corpus(91, n, d)
    labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(2)
    many = hostapi.GpuHnswMap(metric, d, n, devices=[0, 0])
    many.add(rows, labels)
    sr = many.shard_rows
    engines = []
    for s in range(2):
        lo, hi = s * sr, min(n, (s + 1) * sr)
        r = RefHnsw(ref, metric, d, hi - lo)
        r.add(rows[lo:hi], labels[lo:hi])
        engines.append(r)
    for qi in range(20):
        q = make_corpus(300 + qi, 1, d)[0]
        for k, ef in ((10, 128), (5, 5)):
            per = [e.search_knn(q, k, ef) for e in engines]
            wd = np.concatenate([p[0] for p in per])
            wl = np.concatenate([p[1] for p in per])
            order = np.lexsort((wl, wd))[:k]
            gd, gl = many.search_knn(q, k, ef)
            o2 = np.lexsort((gl, gd))
            assert np.array_equal(gl[o2], wl[order]) and np.array_equal(bits(gd[o2]), bits(wd[order])), (qi, k, ef)
    for e in engines:
        e.close()
    many.close()


def test_in_tree_hnsw_shape_takes_its_device_list_from_the_environment(hostapi, monkeypatch):
    """rx_seam.h GpuHnswMapT: `Map(IsArray, metric, dim, maxElements, M, efConstruction)` (hnsw_index.cc:47-58) with the device list from
    RX_GPU_VECTOR_INDEXES."""
    monkeypatch.setenv("RX_GPU_VECTOR_INDEXES", "0,0")
    m = hostapi.GpuHnswMap(1, 16, 500, devices=[])
    assert m.shard_count == 2
    m.close()
    monkeypatch.setenv("RX_GPU_VECTOR_INDEXES", "0")
    m = hostapi.GpuHnswMap(1, 16, 500, devices=[])
    assert m.shard_count == 0
    m.close()


def _knn_equal(a, b, q, k, ef, norm=None):
    ad, al = a.search_knn_norm(q, k, ef, norm)
    bd, bl = b.search_knn_norm(q, k, ef, norm)
    oa, ob = np.lexsort((al, ad)), np.lexsort((bl, bd))
    return np.array_equal(al[oa], bl[ob]) and np.array_equal(bits(ad[oa]), bits(bd[ob]))


@pytest.mark.parametrize("metric", [0, 2])
def test_ann_cache_of_a_map_over_a_device_list_round_trips(hostapi, oracle, metric):
    """WriteIndexCache / LoadIndexCache (hnsw_index.cc:388-507) for the Map over a device list: a graph per shard behind a header no
    single-graph reader accepts.  Loaded into an empty Map over as many shards: same counts, same shards, the same answers (labels and
    distance bits), the same bytes written back, and the Map keeps growing like the one that built the graphs."""
    n, d, k = 2600, 48, 10
    rows = make_corpus(61 + metric, n, d)
    labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(5)
    built = hostapi.GpuHnswMap(metric, d, 3000, M=8, ef_construction=80, devices=[0, 0, 0])
    built.add(rows, labels)
    for lab in labels[::41]:
        built.mark_delete(int(lab))
    cache = built.save_index()
    loaded = hostapi.GpuHnswMap(metric, d, 3000, M=8, ef_construction=80, devices=[0, 0, 0])
    loaded.load_index(cache, labels, rows)
    assert loaded.count == built.count and loaded.deleted_count == built.deleted_count and loaded.shard_rows == built.shard_rows
    for s in range(3):
        assert loaded.shard(s).count == built.shard(s).count
        assert loaded.shard(s).save_index() == built.shard(s).save_index()
    for qi in range(16):
        q = make_corpus(500 + qi, 1, d)[0]
        norm = None
        if metric == 2:
            q, inv = oracle.normalize_copy(q)
            norm = 1.0 / inv
        assert _knn_equal(built, loaded, q, k, 64, norm), qi
        assert _knn_equal(built, loaded, q, 3, 8, norm), qi
    assert loaded.save_index() == cache
    # routing came back with the labels: a delete finds its shard, a new label goes where the builder's Map puts it, an old one is updated in place
    extra = make_corpus(62, 40, d)
    extra_labels = (np.arange(n, n + 40, dtype=np.uint64) << np.uint64(32)) | np.uint64(5)
    for m in (built, loaded):
        m.mark_delete(int(labels[7]))
        m.add(extra, extra_labels)
        m.add(rows[100:101] * np.float32(0.5), labels[100:101])
    assert loaded.count == built.count and loaded.deleted_count == built.deleted_count
    for s in range(3):   # the new labels went to the shards the builder's Map sent them to
        assert loaded.shard(s).count == built.shard(s).count and loaded.shard(s).deleted_count == built.shard(s).deleted_count, s
    # (the graphs themselves part ways from here: a loaded graph draws its levels from a freshly seeded generator, as the reference's reader
    # constructor does — hnswalg.h:297-409; tests/test_ann_cache.py::test_loaded_graph_keeps_building_like_the_reference)
    for m in (built, loaded):
        for i in (0, 13, 39):
            q, norm = extra[i], None
            if metric == 2:
                q, inv = oracle.normalize_copy(q)
                norm = 1.0 / inv
            assert m.search_knn_norm(q, 1, 64, norm)[1].tolist() == [int(extra_labels[i])]
        q, norm = rows[7], None
        if metric == 2:
            q, inv = oracle.normalize_copy(q)
            norm = 1.0 / inv
        assert int(labels[7]) not in m.search_knn_norm(q, 5, 64, norm)[1].tolist()
    with pytest.raises(hostapi.HostError, match="not empty"):
        loaded.load_index(cache, labels, rows)   # refused, and cleared as clearMap() does
    assert loaded.count == 0
    loaded.load_index(cache, labels, rows)       # ... an empty Map again: the cache loads
    assert loaded.count == n
    built.close()
    loaded.close()


def test_ann_cache_of_a_device_list_is_a_miss_for_every_other_reader(hostapi):
    """The sharded stream into a single-graph reader — the reference's engine, the host graph, a single-device Map — and a single graph's
    stream (or one written over another number of shards) into a Map over a device list: each refused with an error, the Map left empty
    (what HnswIndexBase::LoadIndexCache turns into "drop the cache, rebuild")."""
    from oracle import pyoracle
    n, d = 900, 16
    rows = make_corpus(71, n, d)
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    two = hostapi.GpuHnswMap(0, d, n, M=8, ef_construction=60, devices=[0, 0])
    two.add(rows, labels)
    sharded_cache = two.save_index()
    one = hostapi.GpuHnswMap(0, d, n, M=8, ef_construction=60)
    one.add(rows, labels)
    single_cache = one.save_index()
    one.close()
    fresh = hostapi.GpuHnswMap(0, d, n, M=8, ef_construction=60)
    with pytest.raises(hostapi.HostError, match="larger than max elements"):
        fresh.load_index(sharded_cache, labels, rows)
    assert fresh.count == 0
    fresh.load_index(single_cache, labels, rows)   # still usable
    assert fresh.count == n
    fresh.close()
    g = hostapi.HnswGraph(0, d, n, M=8, ef_construction=60)
    with pytest.raises(hostapi.HostError, match="larger than max elements"):
        g.load_index(sharded_cache, labels, rows)
    g.close()
    engine = pyoracle.ref_or_none()
    if engine is not None:   # the reference's own reader (hnswalg.h:297-306), where oracle/_ref travels
        with pytest.raises(RuntimeError, match="larger than max elements"):
            pyoracle.RefHnsw.load_index(engine, sharded_cache, 0, d, labels, rows)
    three = hostapi.GpuHnswMap(0, d, n, M=8, ef_construction=60, devices=[0, 0, 0])
    with pytest.raises(hostapi.HostError, match="written over 2 shards"):
        three.load_index(sharded_cache, labels, rows)
    assert three.count == 0
    with pytest.raises(hostapi.HostError, match="holds one graph"):
        three.load_index(single_cache, labels, rows)
    assert three.count == 0
    three.add(rows[:50], labels[:50])   # ... and the Map works on
    assert three.search_knn(rows[7], 1, 16)[1].tolist() == [int(labels[7])]
    three.close()
    # a Map defined with a smaller capacity than the cache's: every range grows to the writer's shard size first
    small = hostapi.GpuHnswMap(0, d, 64, M=8, ef_construction=60, devices=[0, 0])
    with pytest.raises(hostapi.HostError):
        small.load_index(sharded_cache[: len(sharded_cache) * 3 // 4], labels, rows)   # truncated inside the second shard's graph
    assert small.count == 0
    with pytest.raises(hostapi.HostError, match="no row with the stored key"):
        small.load_index(sharded_cache, labels[:-1], rows[:-1])   # a key the namespace does not hold any more
    assert small.count == 0
    small.load_index(sharded_cache, labels, rows)
    assert small.count == n and small.shard_rows == two.shard_rows
    for qi in range(8):
        assert _knn_equal(two, small, make_corpus(700 + qi, 1, d)[0], 5, 32)
    small.close()
    two.close()


def test_quantised_ann_cache_of_a_device_list(hostapi):
    """A quantised Map over a device list writes its QuantizingParams in front (hnsw.cc:56-62); LoadWithQuantizer brings it back quantised with
    those parameters on every shard, without it the float graphs come back (hnsw.cc:47-53)."""
    n, d, k = 2400, 64, 10
    rows = make_corpus(81, n, d)
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    flt = hostapi.GpuHnswMap(0, d, n, M=8, ef_construction=80, devices=[0, 0])
    flt.add(rows, labels)
    sq = hostapi.GpuHnswMap(0, d, n, M=8, ef_construction=80, devices=[0, 0])
    sq.add(rows, labels)
    params = sq.quantize_config(sample_size=2000)
    cache = sq.save_index()
    back = hostapi.GpuHnswMap(0, d, n, M=8, ef_construction=80, devices=[0, 0])
    back.load_index(cache, labels, rows, with_quantizer=True)
    assert back.is_quantized and np.array_equal(back.quantizing_params, params)
    for s in range(2):
        assert back.shard(s).is_quantized and np.array_equal(back.shard(s).quantizing_params, params)
    plain = hostapi.GpuHnswMap(0, d, n, M=8, ef_construction=80, devices=[0, 0])
    plain.load_index(cache, labels, rows)
    assert not plain.is_quantized and not plain.shard(0).is_quantized
    for qi in range(12):
        q = make_corpus(600 + qi, 1, d)[0]
        assert _knn_equal(sq, back, q, k, 64), qi
        assert _knn_equal(flt, plain, q, k, 64), qi
    assert back.save_index() == cache
    for m in (flt, sq, back, plain):
        m.close()


@pytest.mark.parametrize("metric", [0, 2])
def test_sq8_over_a_device_list_is_the_merge_of_the_quantised_shards(hostapi, oracle, metric):
    """Quantize on a Map over a device list: ONE quantiser (sampled over all points, numbered shard after shard = arrival order, like the
    internal ids of a single graph), a code table per shard; SearchKnn = the k best of the union of the quantised shards' results
    (each shard is a quantised single-device Map, pinned to HierarchicalNSWImpl<uint8_t> in test_gpu_sq8.py)."""
    n, d, k = 3000, 128, 10
    rows = make_corpus(55 + metric, n, d)
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    many = hostapi.GpuHnswMap(metric, d, n, M=8, ef_construction=80, devices=[0, 0, 0])
    one = hostapi.GpuHnswMap(metric, d, n, M=8, ef_construction=80)
    many.add(rows, labels)
    one.add(rows, labels)
    p_many, p_one = many.quantize_config(sample_size=2000), one.quantize_config(sample_size=2000)
    # (the sampler draws from std::rand like the reference's: two runs see two samples — the ranges agree to a few per cent, not to the bit)
    assert np.allclose(p_many[:2], p_one[:2], rtol=0.05) and many.is_quantized
    for s in range(3):   # one quantiser for the Map: every shard codes its rows with the Map's parameters
        assert many.shard(s).is_quantized and np.array_equal(many.shard(s).quantizing_params, p_many)
    for qi in range(12):
        q = make_corpus(300 + qi, 1, d)[0]
        norm = None
        if metric == 2:
            q, inv = oracle.normalize_copy(q)
            norm = 1.0 / inv
        for ef in (64, 16):
            gd, gl = many.search_knn_norm(q, k, ef, norm)
            parts = [many.shard(s).search_knn_norm(q, k, ef, norm) for s in range(3)]
            ad, al = np.concatenate([x[0] for x in parts]), np.concatenate([x[1] for x in parts])
            order = np.lexsort((al, ad))[:k]
            assert np.array_equal(np.sort(gl), np.sort(al[order])), (qi, ef)
            assert np.array_equal(np.sort(gd).view(np.uint32), np.sort(ad[order]).view(np.uint32))
        rd, rl = many.search_range(q, float(np.sort(ad)[5]), 32, norm=norm)
        want = [many.shard(s).search_range(q, float(np.sort(ad)[5]), 32, norm=norm) for s in range(3)]
        assert np.array_equal(np.sort(rl), np.sort(np.concatenate([w[1] for w in want])))
    many.close()
    one.close()


@pytest.mark.parametrize("quantised", [False, True])
def test_streaming_session_over_a_device_list_merges_the_shard_sessions(hostapi, oracle, quantised):
    """BeginStreamingSearch / ContinueStreamingSearch over a device list: a session per shard (the reference's stream over that shard's graph,
    hnswalg.h:1865-1975), merged batch by batch: every shard is asked for as many results as the batch may take from it, the batch takes the
    nearest of everything delivered so far under (dist, label).  Replayed here from the shards' own sessions with the same rule."""
    n, d = 2400, 128
    rows = make_corpus(91, n, d)
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    many = hostapi.GpuHnswMap(0, d, n, M=8, ef_construction=80, devices=[0, 0, 0])
    many.add(rows, labels)
    if quantised:
        many.quantize(float(np.quantile(rows, 0.005)), float(np.quantile(rows, 0.995)))
    for qi, plan in enumerate(([5, 5, 5, 20], [1, 2, 50, 3], [64] * 4, [7] * 9)):
        q = make_corpus(400 + qi, 1, d)[0]
        sess = many.stream(q, 32)
        parts = [dict(s=many.shard(i).stream(q, 32), held=[], done=False) for i in range(3)]
        for b in plan:
            gd, gl, gex = sess.next(b)
            for part in parts:
                if not part["done"] and len(part["held"]) < b:
                    d_, l_, ex = part["s"].next(b - len(part["held"]))
                    part["held"] = sorted(part["held"] + list(zip(d_.view(np.uint32).tolist(), d_.tolist(), l_.tolist())), key=lambda e: (e[1], e[2]))
                    part["done"] = ex
            pool = sorted([(e[1], e[2], e[0], pi) for pi, part in enumerate(parts) for e in part["held"]])[:b]
            for e in pool:
                parts[e[3]]["held"].remove((e[2], e[0], e[1]))
            assert sorted(gl.tolist()) == sorted(e[1] for e in pool), (qi, b)
            assert sorted(gd.view(np.uint32).tolist()) == sorted(e[2] for e in pool)
            assert gex == all(part["done"] and not part["held"] for part in parts)
        sess.close()
        for part in parts:
            part["s"].close()
    many.close()


def test_planner_threads_over_a_device_list_are_served_in_batches(hostapi):
    """T planner threads, one query per SearchKnn call each, on a Map over a device list: calls that arrive while the devices are busy leave as
    ONE fan-out + ONE all-gather (the Map's coalescer in front of the sharded handle) — transparent: every call returns what it returns alone."""
    import threading
    n, d, k = 3000, 64, 10
    rows = make_corpus(95, n, d)
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    many = hostapi.GpuHnswMap(1, d, n, M=8, ef_construction=80, devices=[0, 0, 0])
    many.add(rows, labels)
    queries = make_corpus(96, 96, d)
    alone = [many.search_knn(queries[i], k if i % 3 else 4, 64 if i % 2 else 24) for i in range(96)]
    out, errs = [None] * 96, []

    def work(t):
        try:
            for i in range(t, 96, 12):
                out[i] = many.search_knn(queries[i], k if i % 3 else 4, 64 if i % 2 else 24)
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    for _ in range(3):
        ths = [threading.Thread(target=work, args=(t,)) for t in range(12)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        assert not errs, errs
        for i in range(96):
            a, b = alone[i], out[i]
            oa, ob = np.lexsort((a[1], a[0])), np.lexsort((b[1], b[0]))
            assert np.array_equal(a[1][oa], b[1][ob]) and np.array_equal(bits(a[0][oa]), bits(b[0][ob])), i
    many.close()
