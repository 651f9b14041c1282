"""CPU: reindexer_amd/host/sorted_union.h — the row list of an IVF query (union of the probed inverted lists' rows, ascending).  Both strategies
and the cost-model pick must equal numpy's sorted concatenation."""
import ctypes as C

import numpy as np
import pytest


def union(runs, universe, strategy):
    from reindexer_amd import hostapi
    L = hostapi.lib()
    L.rxhost_sorted_union.restype = C.c_long
    L.rxhost_sorted_union.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
    flat = np.ascontiguousarray(np.concatenate(runs) if runs else np.empty(0, np.uint32), np.uint32)
    off = np.zeros(len(runs) + 1, np.uint64)
    off[1:] = np.cumsum([len(r) for r in runs])
    out = np.empty(max(flat.size, 1), np.uint32)
    n = L.rxhost_sorted_union(flat.ctypes.data if flat.size else None, off.ctypes.data, len(runs), universe, strategy, out.ctypes.data, out.size)
    assert n >= 0
    return out[:n].copy()


@pytest.mark.parametrize("universe,nlists,probe", [(1000, 7, 3), (200_000, 256, 1), (200_000, 256, 40), (200_000, 256, 256), (5_000_000, 64, 2)])
def test_sorted_union_matches_numpy(universe, nlists, probe):
    rng = np.random.default_rng(universe + probe)
    n = min(universe, 300_000)
    rows = rng.choice(universe, n, replace=False).astype(np.uint32)
    owner = rng.integers(0, nlists, n)
    lists = [np.sort(rows[owner == i]) for i in range(nlists)]
    lists[0] = np.empty(0, np.uint32)   # an empty list among the probed ones
    pick = [lists[i] for i in rng.choice(nlists, probe, replace=False)] + [lists[0]]
    want = np.sort(np.concatenate(pick)).astype(np.uint32)
    for strategy in (0, 1, 2):
        assert np.array_equal(union(pick, universe, strategy), want), strategy
    assert union([], universe, 0).size == 0
