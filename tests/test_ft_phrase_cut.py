"""CPU: reindexer_amd/csrc/ft_phrase_cut.h — PhraseMerger's admission cut (phrasemerger.h:341, phrasemergerimpl.h:181-183, 209-215) settled
between document-range shards.  The plain rule: the candidates of a phrase's first term in (sub-term row, document) order, the first
mergeLimit of them stay.  The sharded layer only sees, per shard and row, how many candidates that shard admitted under its LOCAL bound
min(mergeLimit, postings of its fragment) — possibly truncated — and must hand every shard the length of the prefix of its own slots that
survives.  Replayed here against the plain rule over random corpora (the GPU suite checks the merges themselves: tests/test_gpu_ft_sharded.py)."""
import ctypes as C

import numpy as np
import pytest


def cut(counts, limit):
    from reindexer_amd import hostapi
    L = hostapi.lib()
    L.rxhost_ft_shard_phrase_cut.restype = None
    L.rxhost_ft_shard_phrase_cut.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_void_p]
    counts = np.ascontiguousarray(counts, np.uint32)
    keep = np.zeros(counts.shape[0], np.uint64)
    L.rxhost_ft_shard_phrase_cut(counts.ctypes.data, counts.shape[0], counts.shape[1], limit, keep.ctypes.data)
    return keep


@pytest.mark.parametrize("seed", range(12))
def test_cut_equals_the_plain_rule(seed):
    rng = np.random.default_rng(seed)
    shards, rows = int(rng.integers(2, 9)), int(rng.integers(1, 7))
    docs_per_shard = int(rng.integers(50, 400))
    # candidate[r][d]: document d is added by row r (its first posting of the first term that is preselected with a non-zero rank); postings[r][d]:
    # row r holds a posting of d at all (candidates are a subset: the local bound counts postings, the admission counts candidates)
    total = shards * docs_per_shard
    postings = rng.random((rows, total)) < rng.uniform(0.05, 0.6)
    candidate = postings & (rng.random((rows, total)) < 0.7)
    seen = np.zeros(total, bool)
    for r in range(rows):   # a document is a candidate of the FIRST row that could add it
        candidate[r] &= ~seen
        seen |= candidate[r]
    shard_of = np.arange(total) // docs_per_shard
    for limit in sorted({1, 7, int(candidate.sum() // 3) + 1, int(candidate.sum()), int(candidate.sum()) + 5, int(postings.sum()) + 1}):
        # the plain rule over the whole index
        order = [(r, d) for r in range(rows) for d in np.flatnonzero(candidate[r])]
        want = np.zeros(shards, np.uint64)
        for r, d in order[:limit]:
            want[shard_of[d]] += 1
        # what every shard reports: its candidates in ITS (row, document) order, cut at its local bound
        counts = np.zeros((shards, rows), np.uint32)
        for s in range(shards):
            mine = shard_of == s
            bound = min(limit, int(postings[:, mine].sum()))
            left = bound
            for r in range(rows):
                c = min(int(candidate[r, mine].sum()), left)
                counts[s, r] = c
                left -= c
        got = cut(counts, limit)
        assert np.array_equal(got, want), (seed, limit, got, want)
        assert got.sum() == min(limit, int(candidate.sum()))
        assert np.all(got <= counts.sum(axis=1))   # a prefix of what the shard admitted itself


def test_a_shard_that_admitted_nothing_and_an_empty_phrase():
    counts = np.array([[3, 2], [0xFFFFFFFF, 0], [1, 4]], np.uint32)   # the middle shard holds no posting of the first term
    assert cut(counts, 100).tolist() == [5, 0, 5]
    assert cut(counts, 5).tolist() == [4, 0, 1]    # row 0: 3 + 1, then one of shard 0's row-1 candidates
    assert cut(counts, 4).tolist() == [3, 0, 1]
    assert cut(counts, 0).tolist() == [0, 0, 0]
    assert cut(np.zeros((4, 3), np.uint32), 10).tolist() == [0, 0, 0, 0]
