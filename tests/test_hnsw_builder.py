"""CPU: the product's host-side HNSW builder (reindexer_amd/host/hnsw_graph.cc) must build, for sequential inserts,
LINK FOR LINK the graph the real reference engine builds (same level RNG stream, same distance bits, same heap tie
mechanics) — compared against the engine itself where oracle/_ref exists, and against tests/golden/hnsw.npz elsewhere."""
from pathlib import Path

import numpy as np
import pytest

from .conftest import make_corpus

G = Path(__file__).resolve().parent / "golden"


def graphs_equal(a, b):
    for key in ("n", "M", "maxM0", "maxlevel", "entry", "num_deleted"):
        assert a[key] == b[key], key
    for key in ("levels", "labels", "deleted", "links0", "upper_off"):
        assert np.array_equal(a[key], b[key]), key
    blocks = int(a["upper_off"][-1])
    assert np.array_equal(a["upper"][:blocks], b["upper"][:blocks]), "upper"


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("shape", [(2500, 32, 16, 200), (1200, 100, 8, 40), (600, 128, 16, 200)])
def test_builder_equals_reference_graph(ref, metric, shape):
    from oracle.pyoracle import RefHnsw
    from reindexer_amd import hostapi
    n, d, M, efc = shape
    rows = make_corpus(31 + n, n, d)
    labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(2)
    r = RefHnsw(ref, metric, d, n, M=M, ef_construction=efc)
    r.add(rows, labels)
    g = hostapi.HnswGraph(metric, d, n, M=M, ef_construction=efc)
    g.add(rows, labels)
    dele = labels[np.random.default_rng(1).choice(n, 50, replace=False)]
    for lab in dele:
        r.mark_delete(lab)
        g.mark_delete(lab)
    graphs_equal(r.export(with_vectors=False), g.export())
    r.close()
    g.close()


def test_builder_matches_golden_graph():
    """The same check on machines without the reference tree: graph exported from the real engine, committed."""
    from reindexer_amd import hostapi
    z = np.load(G / "hnsw.npz")
    rows, labels = z["rows"], z["labels"]
    n, d = rows.shape
    g = hostapi.HnswGraph(int(z["metric"]), d, n, M=int(z["M"]), ef_construction=int(z["efc"]))
    g.add(rows, labels)
    e = g.export()
    assert e["maxlevel"] == int(z["maxlevel"]) and e["entry"] == int(z["entry"])
    assert np.array_equal(e["links0"], z["links0"]) and np.array_equal(e["levels"], z["levels"])
    blocks = int(e["upper_off"][-1])
    assert np.array_equal(e["upper"][:blocks], z["upper"][:blocks])
    g.close()


def test_builder_errors():
    from reindexer_amd import hostapi
    g = hostapi.HnswGraph(0, 8, 2, M=4, ef_construction=10)
    rows = make_corpus(1, 3, 8)
    g.add(rows[:2], np.array([1, 2], np.uint64))
    with pytest.raises(hostapi.HostError, match="exceeds the specified limit"):
        g.add(rows[2:3], np.array([3], np.uint64))
    with pytest.raises(hostapi.HostError, match="Label not found"):
        g.mark_delete(77)
    g.mark_delete(1)
    with pytest.raises(hostapi.HostError, match="Label not found"):
        g.mark_delete(1)   # label removed from the lookup on delete (allow_replace_deleted semantics)
    g.close()


# ------------------------------------------------------------------------------------------------ concurrent construction
def exact_topk(rows, q, k, metric):
    if metric == 0:
        d = ((rows - q) ** 2).sum(axis=1)
    else:
        d = -(rows @ q)
        if metric == 2:
            d = d / np.linalg.norm(rows, axis=1)
    return np.argsort(d, kind="stable")[:k]


def graph_invariants(e, n, M):
    links0, levels = e["links0"], e["levels"]
    assert e["n"] == n and links0.shape == (n, 1 + 2 * M)
    cnt = links0[:, 0]
    assert cnt.max() <= 2 * M
    for i in range(n):
        ll = links0[i, 1:1 + cnt[i]]
        assert ll.size == np.unique(ll).size and i not in ll and (ll < n).all(), i
    off, upper = e["upper_off"], e["upper"].reshape(-1, 1 + M)
    for i in np.flatnonzero(levels > 0):
        for lv in range(1, levels[i] + 1):
            blk = upper[off[i] + lv - 1]
            ll = blk[1:1 + blk[0]]
            assert blk[0] <= M and ll.size == np.unique(ll).size and i not in ll, (i, lv)
            assert (levels[ll] >= lv).all(), (i, lv)
    assert levels[e["entry"]] == e["maxlevel"] == levels.max()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_concurrent_path_from_one_thread_equals_sequential_graph(metric):
    """AddPointConcurrent (addPoint<RegularLocker>) driven from ONE thread takes every lock of the multithreaded build and must still produce the
    sequential graph link for link — the concurrent code is the same algorithm."""
    from reindexer_amd import hostapi
    n, d, M, efc = 3000, 48, 12, 100
    rows = make_corpus(77, n, d)
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    a = hostapi.HnswGraph(metric, d, n, M=M, ef_construction=efc)
    a.add(rows, labels)
    b = hostapi.HnswGraph(metric, d, n, M=M, ef_construction=efc)
    b.add(rows, labels, threads=1)
    graphs_equal(a.export(), b.export())
    a.close()
    b.close()


@pytest.mark.parametrize("metric", [0, 2])
def test_concurrent_build_is_a_valid_graph_with_the_sequential_recall(oracle, metric):
    """8 inserting threads (the reference's HierarchicalNSWMT build): the graph depends on timing, so the checks are structural (degree bounds,
    no self / duplicate / dangling links, links only to nodes that own the level, entry point on the top level, the same multiset of levels —
    the level RNG stream is shared) plus recall@10 of the restated search, which must match the sequential graph's within 0.02."""
    from oracle.pyoracle import oracle_hnsw_search_knn
    from reindexer_amd import hostapi
    n, d, M, efc = 12000, 32, 16, 200
    rows = make_corpus(5, n, d)
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    inv = oracle.l2_modules(rows) if metric == 2 else None
    graphs = {}
    for name, threads in (("seq", 0), ("mt", 8)):
        g = hostapi.HnswGraph(metric, d, n, M=M, ef_construction=efc)
        g.add(rows, labels, threads=threads)
        graphs[name] = g.export()
        graphs[name]["vectors"] = rows
        g.close()
    graph_invariants(graphs["mt"], n, M)
    assert np.array_equal(np.sort(graphs["mt"]["levels"]), np.sort(graphs["seq"]["levels"]))
    assert np.array_equal(np.sort(graphs["mt"]["labels"][:n]), labels)
    queries = make_corpus(6, 60, d)
    recall = {}
    for name, g in graphs.items():
        # internal ids differ between the two builds (insertion order), labels do not: compare through labels
        lab2row = {int(l): i for i, l in enumerate(labels)}
        vec = np.stack([rows[lab2row[int(l)]] for l in g["labels"][:n]])
        g["vectors"] = vec
        ginv = oracle.l2_modules(vec) if metric == 2 else None
        hit = 0
        for q in queries:
            qq = oracle.normalize_copy(q)[0] if metric == 2 else q
            _, got = oracle_hnsw_search_knn(oracle, g, qq, 10, 64, ginv)
            want = labels[exact_topk(rows, qq, 10, metric)]
            hit += len(set(int(x) for x in got) & set(int(x) for x in want))
        recall[name] = hit / (10 * len(queries))
    assert recall["mt"] >= 0.9 and abs(recall["mt"] - recall["seq"]) <= 0.02, recall


def test_concurrent_build_errors_propagate():
    from reindexer_amd import hostapi
    g = hostapi.HnswGraph(0, 8, 10, M=4, ef_construction=10)
    rows = make_corpus(1, 12, 8)
    with pytest.raises(hostapi.HostError, match="exceeds the specified limit"):
        g.add(rows, np.arange(12, dtype=np.uint64), threads=4)
    g.close()


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("shape", [(900, 24, 8, 60), (500, 64, 16, 200), (1500, 16, 4, 30)])
def test_builder_delete_upsert_cycles_equal_reference_graph(ref, metric, shape):
    """updatePoint + deleted-slot reuse (hnswalg.h:1401-1587, 1589-1680): after rounds of deletes and inserts — the new points take the
    vacated slots in the order of the reference's hash set (*deleted_elements.begin()), their one- / two-hop neighbourhoods are re-selected
    and the element is re-linked — the graph equals the real engine's link for link, and the element count does not grow."""
    from oracle.pyoracle import RefHnsw
    from reindexer_amd import hostapi
    n, d, M, efc = shape
    rng = np.random.default_rng(n + metric)
    rows = make_corpus(5 + n, n, d)
    labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(1)
    r = RefHnsw(ref, metric, d, n + 40, M=M, ef_construction=efc)
    g = hostapi.HnswGraph(metric, d, n + 40, M=M, ef_construction=efc)
    for x in (r, g):
        x.add(rows, labels)
    live = list(labels)
    next_label = n
    for rnd in range(6):
        # delete a batch (enough of them, in later rounds, to take the hash set through a rehash: > 15 elements)
        k = [3, 10, 25, 40, 7, 60][rnd]
        pick = rng.choice(len(live), k, replace=False)
        for idx in sorted(pick.tolist(), reverse=True):
            lab = live.pop(idx)
            r.mark_delete(lab)
            g.mark_delete(lab)
        graphs_equal(r.export(with_vectors=False), g.export())
        # insert fewer / as many / more points than were deleted: slots are recycled first, the rest is appended
        m = [3, 6, 30, 40, 2, 75][rnd]
        new_rows = make_corpus(1000 + rnd + n, m, d)
        if rnd == 2:
            new_rows[:5] = rows[:5]   # exact duplicates of live vectors: zero / tied distances in the re-selection
        new_labels = ((np.arange(next_label, next_label + m, dtype=np.uint64)) << np.uint64(32)) | np.uint64(2)
        next_label += m
        for i in range(m):
            r.add(new_rows[i:i + 1], new_labels[i:i + 1])
            g.add(new_rows[i:i + 1], new_labels[i:i + 1])
        live.extend(new_labels.tolist())
        e_ref, e_got = r.export(with_vectors=False), g.export()
        graphs_equal(e_ref, e_got)
    assert r.count == g.export()["n"] <= n + 40
    r.close()
    g.close()


def test_builder_update_existing_label_in_place(ref):
    """addPoint with a label that is already present updates the element in place (hnswalg.h:1709-1724) — same graph as the reference's."""
    from oracle.pyoracle import RefHnsw
    from reindexer_amd import hostapi
    n, d = 400, 20
    rows = make_corpus(77, n, d)
    labels = np.arange(n, dtype=np.uint64) + np.uint64(10)
    r = RefHnsw(ref, 0, d, n, M=8, ef_construction=40)
    g = hostapi.HnswGraph(0, d, n, M=8, ef_construction=40)
    for x in (r, g):
        x.add(rows, labels)
    upd = make_corpus(78, 12, d)
    for i in range(12):
        lab = labels[i * 17:i * 17 + 1]
        r.add(upd[i:i + 1], lab)
        g.add(upd[i:i + 1], lab)
    graphs_equal(r.export(with_vectors=False), g.export())
    assert g.export()["n"] == n
    r.close()
    g.close()
