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
