"""The reference's ANN disk cache (hnswalg.h:1213-1263 SaveIndex, :297-409 reader constructor, :1264-1281 initTree; flag of hnsw.cc:41-70)
for the product's HNSW graph / Map: a cache written by either engine must load into the other and give the same graph, and the two
writers must produce the same bytes for the same graph.  The streams go through in-memory IWriter / IReader stand-ins on both sides
(8 bytes per var-int, u64 length + bytes per string, 8-byte label per primary key), so what is compared is the ORDER and CONTENT of the
writer calls, which is the format.  CPU tests use the real engine from oracle/_ref; the GPU test loads a cache into a GpuHnswMap."""
import numpy as np
import pytest

from .conftest import make_corpus
from .test_hnsw_builder import graphs_equal as same_graph


def graphs_equal(a, b):
    """A deleted element has no primary key in the cache (its vector is stored instead), so its label does not survive a load."""
    a, b = dict(a), dict(b)
    for g in (a, b):
        g["labels"] = np.where(g["deleted"] != 0, np.uint64(0), g["labels"])
    same_graph(a, b)


def build_pair(ref, metric, n, d, M, efc, deletes=0, seed=5):
    from oracle.pyoracle import RefHnsw
    from reindexer_amd import hostapi
    rows = make_corpus(seed + n, n, d)
    labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(7)
    r = RefHnsw(ref, metric, d, n, M=M, ef_construction=efc)
    g = hostapi.HnswGraph(metric, d, n, M=M, ef_construction=efc)
    if n:
        r.add(rows, labels)
        g.add(rows, labels)
    if deletes:
        for lab in labels[np.random.default_rng(seed).choice(n, deletes, replace=False)]:
            r.mark_delete(lab)
            g.mark_delete(lab)
    return rows, labels, r, g


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("shape", [(1500, 24, 16, 100, 0), (900, 64, 8, 40, 60), (300, 128, 16, 200, 299), (1, 8, 4, 10, 0)])
def test_writers_agree_and_caches_cross_load(ref, metric, shape):
    from oracle.pyoracle import RefHnsw
    from reindexer_amd import hostapi
    n, d, M, efc, deletes = shape
    rows, labels, r, g = build_pair(ref, metric, n, d, M, efc, deletes)
    theirs, ours = r.save_index(), g.save_index()
    assert len(ours) == len(theirs)
    # the reference's cache into the product's graph.  The reference writes its upper-level blocks raw, stale slots past each list's
    # count included; the product's writer zeroes those, so the byte comparison is made on the re-saved stream
    g2 = hostapi.HnswGraph(metric, d, n, M=M, ef_construction=efc)
    g2.load_index(theirs, labels, rows)
    graphs_equal(g.export(), g2.export())
    vec, inv = g2.vector_views(n)
    src, src_inv = g.vector_views(n)
    assert np.array_equal(vec, src)
    if metric == 2:
        assert np.array_equal(inv, src_inv)
    assert g2.save_index() == ours
    # the product's cache into the reference's engine
    r2 = RefHnsw.load_index(ref, ours, metric, d, labels, rows)
    graphs_equal(r.export(with_vectors=False), r2.export(with_vectors=False))
    q = make_corpus(99, 4, d)
    for i in range(4):
        a, b = r.search_knn(q[i], 10, 64), r2.search_knn(q[i], 10, 64)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    for x in (r, r2, g, g2):
        x.close()


def test_loader_reads_the_golden_reference_cache():
    """No reference needed: a cache written by the real engine is committed (tests/golden/ann_cache.npz, make_golden.py) with the graph the
    engine holds; the product's loader must rebuild that graph from the stream, write a stream of the same length back, and read its own
    stream to the same graph again."""
    from pathlib import Path
    from reindexer_amd import hostapi
    z = np.load(Path(__file__).resolve().parent / "golden" / "ann_cache.npz")
    for metric in (0, 2):
        meta = z[f"m{metric}_meta"]
        _, n, d, M, efc, maxlevel, entry, ndel = (int(x) for x in meta)
        cache, rows, labels = z[f"m{metric}_cache"].tobytes(), z[f"m{metric}_rows"], z[f"m{metric}_labels"]
        g = hostapi.HnswGraph(metric, d, n, M=M, ef_construction=efc)
        g.load_index(cache, labels, rows)
        e = g.export()
        assert (e["n"], e["maxlevel"], e["entry"], e["num_deleted"]) == (n, maxlevel, entry, ndel)
        for key in ("links0", "levels", "deleted", "upper_off"):
            assert np.array_equal(e[key], z[f"m{metric}_{key}"]), key
        blocks = int(e["upper_off"][-1])
        count = e["upper"][:blocks, 0].astype(np.int64)
        live = np.arange(e["upper"].shape[1] - 1)[None, :] < count[:, None]   # the slots below each list's count: the rest is the writer's stale data
        assert np.array_equal(e["upper"][:blocks, 0], z[f"m{metric}_upper"][:, 0])
        assert np.array_equal(np.where(live, e["upper"][:blocks, 1:], 0), np.where(live, z[f"m{metric}_upper"][:, 1:], 0))
        alive = e["deleted"] == 0
        assert np.array_equal(e["labels"][alive], labels[alive])
        vec, _ = g.vector_views(n)
        assert np.array_equal(vec[alive], rows[alive])
        mine = g.save_index()
        assert len(mine) == len(cache)
        g2 = hostapi.HnswGraph(metric, d, n, M=M, ef_construction=efc)
        g2.load_index(mine, labels, rows)
        assert g2.save_index() == mine
        g.close()
        g2.close()


def test_loaded_graph_keeps_building_like_the_reference(ref):
    """After LoadIndex both engines continue from the same state: the level generator restarts from its seed in the reader constructor
    (hnswalg.h:301-303) and the slots of deleted elements are reusable (allow_replace_deleted).  The reference sizes the loaded graph
    from the stream, so the continuation stays inside the stored max_elements: the deleted slots."""
    from oracle.pyoracle import RefHnsw
    from reindexer_amd import hostapi
    metric, n, d, M, efc, deletes = 0, 800, 32, 16, 100, 40
    rows, labels, r, g = build_pair(ref, metric, n, d, M, efc, deletes=deletes)
    cache = r.save_index()
    more = make_corpus(1234, deletes, d)
    more_labels = (np.arange(n, n + deletes, dtype=np.uint64) << np.uint64(32)) | np.uint64(7)
    r2 = RefHnsw.load_index(ref, cache, metric, d, labels, rows)
    g2 = hostapi.HnswGraph(metric, d, n, M=M, ef_construction=efc)
    g2.load_index(cache, labels, rows)
    for x in (r, g, r2, g2):
        x.add(more, more_labels)
    graphs_equal(r2.export(with_vectors=False), g2.export())
    graphs_equal(r.export(with_vectors=False), g.export())
    all_rows, all_labels = np.concatenate([rows, more]), np.concatenate([labels, more_labels])
    g3 = hostapi.HnswGraph(metric, d, n, M=M, ef_construction=efc)
    g3.load_index(r2.save_index(), all_labels, all_rows)
    assert g3.save_index() == g2.save_index()
    for x in (r, g, r2, g2, g3):
        x.close()


def test_load_errors(ref):
    from reindexer_amd import hostapi
    metric, n, d, M, efc = 1, 400, 16, 8, 40
    rows, labels, r, g = build_pair(ref, metric, n, d, M, efc, deletes=10)
    cache = g.save_index()
    # construction constants differ -> refused, graph left empty (hnswalg.h:322-331)
    other = hostapi.HnswGraph(metric, d, n, M=16, ef_construction=efc)
    with pytest.raises(hostapi.HostError, match="M"):
        other.load_index(cache, labels, rows)
    assert other.export()["n"] == 0
    other.close()
    # into a non-empty graph
    with pytest.raises(hostapi.HostError, match="empty"):
        g.load_index(cache, labels, rows)
    g.clear()
    assert g.export()["n"] == 0
    g.load_index(cache, labels, rows)
    assert g.save_index() == cache
    # truncated streams and a missing row
    fresh = hostapi.HnswGraph(metric, d, n, M=M, ef_construction=efc)
    for cut in (0, 8, 40, len(cache) // 2, len(cache) - 1):
        with pytest.raises(hostapi.HostError):
            fresh.load_index(cache[:cut], labels, rows)
        assert fresh.export()["n"] == 0
    with pytest.raises(hostapi.HostError, match="unparsed"):
        fresh.load_index(cache + b"\0" * 8, labels, rows)
    fresh.clear()
    alive = np.flatnonzero(g.export()["deleted"] == 0)
    drop = alive[len(alive) // 2]
    keep = np.ones(n, bool)
    keep[drop] = False
    with pytest.raises(hostapi.HostError, match="no row"):
        fresh.load_index(cache, labels[keep], rows[keep])
    assert fresh.export()["n"] == 0
    # an upper-level link that leads to an element WITHOUT that level (a corrupt cache): refused at load time — search, insert and the device
    # export would index that element's upper lists out of bounds
    ex = g.export()
    tall = int(np.flatnonzero(ex["levels"] >= 1)[0])
    flat = int(np.flatnonzero(ex["levels"] == 0)[0])
    blk = np.asarray(ex["upper"], np.uint32)[int(ex["upper_off"][tall])].copy()   # the element's level-1 block: count + M links
    assert blk[0] >= 1
    at = cache.find(blk.tobytes())
    assert at > 0 and cache.find(blk.tobytes(), at + 1) < 0
    bad = blk.copy()
    bad[1] = flat
    fresh.clear()
    with pytest.raises(hostapi.HostError, match="without that level"):
        fresh.load_index(cache[:at] + bad.tobytes() + cache[at + blk.nbytes:], labels, rows)
    assert fresh.export()["n"] == 0
    # too small a graph for the cache
    small = hostapi.HnswGraph(metric, d, n // 2, M=M, ef_construction=efc)
    small.load_index(cache, labels, rows)   # the stored max_elements wins, as in the reader constructor
    assert small.export()["n"] == n
    for x in (r, g, fresh, small):
        x.close()


def test_the_header_of_a_sharded_cache_is_refused_by_every_single_graph_reader(ref):
    """The ANN cache of a Map over a device list (gpu_hnsw_map.cc, SaveIndex) opens — behind the quantisation flag — with capacity 0 and
    count = the number of shards, where one graph states its capacity and its element count: the reference's reader constructor
    (hnswalg.h:297-306) and the product's graph loader stop right there with the same error, which HnswIndexBase::LoadIndexCache turns into
    "drop the cache and rebuild".  (The stream itself is written and read back on the GPU: tests/test_gpu_sharded_hnsw.py.)"""
    import struct
    from oracle.pyoracle import RefHnsw
    from reindexer_amd import hostapi
    d = 16
    rows, labels = make_corpus(5, 10, d), np.arange(10, dtype=np.uint64)
    for shards in (2, 8):
        head = struct.pack("<QQQQQQ", 0, 0, shards, 1, 1000, 512)   # float graph; capacity 0, count = shards; version, Map capacity, shard rows
        for stream in (head, head + b"\0" * 256):
            g = hostapi.HnswGraph(1, d, 100, M=8, ef_construction=40)
            with pytest.raises(hostapi.HostError, match="Current elements count is larger than max elements count"):
                g.load_index(stream, labels, rows)
            assert g.export()["n"] == 0
            g.close()
            with pytest.raises(RuntimeError, match="Current elements count is larger than max elements count"):
                RefHnsw.load_index(ref, stream, 1, d, labels, rows)


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_gpu_map_serves_a_reference_cache(metric):
    """A cache written by the reference's engine, loaded into the GPU Map: searches equal the ones of a Map that built the graph itself,
    and the Map writes the same cache back."""
    from reindexer_amd import hostapi
    n, d, M, efc = 4000, 48, 16, 100
    rows = make_corpus(77, n, d)
    labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(3)
    built = hostapi.GpuHnswMap(metric, d, n, M=M, ef_construction=efc)
    built.add(rows, labels)
    for lab in labels[::37]:
        built.mark_delete(lab)
    cache = built.save_index()
    from oracle import pyoracle
    engine = pyoracle.ref_or_none()
    if engine is not None and engine.simd_level == 3:   # on the GPU box: the prebuilt oracle/_ref, when the host has AVX-512
        r = pyoracle.RefHnsw(engine, metric, d, n, M=M, ef_construction=efc)
        r.add(rows, labels)
        for lab in labels[::37]:
            r.mark_delete(lab)
        theirs = r.save_index()   # raw upper-level blocks: compared after a pass through the product's loader, which zeroes the stale slots
        r.close()
        assert len(theirs) == len(cache)
        via = hostapi.HnswGraph(metric, d, n, M=M, ef_construction=efc)
        via.load_index(theirs, labels, rows)
        assert via.save_index() == cache
        via.close()
    loaded = hostapi.GpuHnswMap(metric, d, n, M=M, ef_construction=efc)
    loaded.load_index(cache, labels, rows)
    assert loaded.count == built.count and loaded.deleted_count == built.deleted_count
    q = make_corpus(78, 32, d)
    for i in range(32):
        a, b = built.search_knn(q[i], 10, 64), loaded.search_knn(q[i], 10, 64)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
        radius = float(a[0][5])
        a, b = built.search_range(q[i], radius, 64), loaded.search_range(q[i], radius, 64)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    assert loaded.save_index() == cache
    # an inserted point after the load lands where the builder's own Map puts it
    extra = make_corpus(79, 1, d)[0]
    built.add(extra[None], np.array([1 << 40], np.uint64))
    loaded.add(extra[None], np.array([1 << 40], np.uint64))
    for i in range(8):
        a, b = built.search_knn(q[i], 10, 64), loaded.search_knn(q[i], 10, 64)
        assert np.array_equal(a[1], b[1])
    with pytest.raises(hostapi.HostError):
        loaded.load_index(cache, labels, rows)   # not empty any more -> refused, and cleared like clearMap()
    assert loaded.count == 0
    built.close()
    loaded.close()
