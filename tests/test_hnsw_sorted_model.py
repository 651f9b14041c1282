"""CPU: the rule set of the sorted-list HNSW search (HnswSortedList in reindexer_amd/csrc/hnsw_search.hip), restated in Python and run
against the C restatement of the reference's two-heap search (oracle/oracle_hnsw.c) on graphs FULL of equal distances.

The device kernel keeps top_candidates + candidate_set as one sorted list and claims: unless it raises its tie flag, pops, evictions and
lowerBound depend on keys alone, so the result is the reference's whatever its heaps did among equal keys.  That claim is what this file
attacks: small integer-grid corpora (most distances repeat), small ef, thousands of searches.  Every search the model does NOT flag must
return exactly the oracle's (distance bits, label) set and take the oracle's number of hops; flagged searches are counted — the flag
must stay the exception on tie-free data and must not swallow everything on tie-heavy data (else the test proves nothing)."""
import math

import numpy as np
import pytest

from .conftest import make_corpus


class SortedListSearch:
    """Line by line the layer-0 loop of hnsw_search_kernel<..., kSorted>: same insertion position among equal keys (in front of
    them), same bookkeeping of `outside` / `pend`, same places where the tie flag is raised."""

    def __init__(self, g, dist, ef, k):
        self.g, self.dist, self.ef, self.k = g, dist, ef, k
        self.keys, self.ids, self.done = [], [], []
        self.lower = np.float32(3.402823466e+38)
        self.outside = math.nan
        self.pend, self.pending, self.tie = 0.0, False, False
        self.hops = 0
        self.pop_ties = self.equal_inserts = 0   # how often the rules were actually exercised

    @property
    def n(self):
        return len(self.keys)

    def insert(self, nd, nid):
        if not (nd < math.inf):
            self.tie = True
        if self.n == self.ef:
            self.outside = self.lower
        pos = sum(1 for x in self.keys if x < nd)
        self.equal_inserts += int(any(x == nd for x in self.keys))
        self.keys.insert(pos, nd)
        self.ids.insert(pos, nid)
        self.done.insert(pos, False)
        if self.n > self.ef:
            self.keys.pop()
            self.ids.pop()
            self.done.pop()
        self.lower = self.keys[-1]

    def first_open(self):
        for e, d in enumerate(self.done):
            if not d:
                return e
        return -1

    def settle(self):
        if self.pending and self.n == self.ef and not (self.lower > self.pend):
            self.tie = True
        self.pending = False

    def pop(self):
        e = self.first_open()
        if e < 0:
            self.settle()
            return None
        dist, node = self.keys[e], self.ids[e]
        if self.pending and dist > self.pend:
            self.settle()
        self.done[e] = True
        nxt = self.first_open()
        if dist == self.outside:
            self.tie = True
        if nxt >= 0 and self.keys[nxt] == dist:
            self.pend = max(self.pend, dist) if self.pending else dist
            self.pending = True
            self.pop_ties += 1
        return node, dist

    def entry_point(self):
        """getLayer0EntryPoint: greedy descent through the upper levels, first improvement in list order wins the step."""
        g, dist = self.g, self.dist
        cur = int(g["entry"])
        curdist = dist[cur]
        M = g["M"]
        for level in range(int(g["maxlevel"]), 0, -1):
            changed = True
            while changed:
                changed = False
                block = g["upper"][int(g["upper_off"][cur]) + level - 1]
                for nb in block[1:1 + int(block[0])]:
                    if dist[nb] < curdist:
                        curdist, cur, changed = dist[nb], int(nb), True
        assert M == g["M"]
        return cur, curdist

    def run(self):
        g, dist, ef = self.g, self.dist, self.ef
        cur, curdist = self.entry_point()
        self.insert(curdist, cur)
        visited = {cur}
        while True:
            got = self.pop()
            if got is None:
                if self.n == ef and self.lower == self.outside:
                    self.tie = True
                break
            node, cdist = got
            if self.tie or cdist > self.lower:
                break
            self.hops += 1
            row = g["links0"][node]
            fresh = []
            for nb in row[1:1 + int(row[0])]:
                nb = int(nb)
                if nb not in visited:
                    visited.add(nb)
                    fresh.append(nb)
            for base in range(0, len(fresh), 64):   # one wave-wide chunk of neighbours at a time, like the kernel
                chunk = fresh[base:base + 64]
                admitted = [self.n < ef or self.lower > dist[nb] for nb in chunk]
                if self.n == ef and any(dist[nb] == self.lower for nb in chunk):
                    self.outside = self.lower
                for ok, nb in zip(admitted, chunk):
                    if not ok:
                        continue
                    nd = dist[nb]
                    if self.n < ef or self.lower > nd:
                        self.insert(nd, nb)
                    elif nd == self.lower:
                        self.outside = nd
        keep = min(self.n, self.k)
        if not self.tie and self.n > keep and self.keys[keep - 1] == self.keys[keep]:
            self.tie = True
        return [(self.keys[i], self.ids[i]) for i in range(keep)]


FLT_MAX = float(np.float32(3.402823466e+38))


class SortedListSearchDel:
    """Non-bare form: deleted nodes are candidates only."""
    def __init__(self, g, dist, ef, k, cap):
        self.g, self.dist, self.ef, self.k, self.cap = g, dist, ef, k, cap
        self.keys, self.ids, self.done, self.isdel = [], [], [], []
        self.live = 0
        self.lower = FLT_MAX
        self.outside = math.nan
        self.pend, self.pending, self.tie = 0.0, False, False
        self.hops = 0
        self.pop_ties = 0
    @property
    def n(self): return len(self.keys)
    def last_live(self):
        for e in range(self.n - 1, -1, -1):
            if not self.isdel[e]: return e
        return -1
    def truncate_after_last_live(self):
        L = self.last_live()
        if L + 1 < self.n:
            self.outside = self.keys[L + 1]      # the smallest key that leaves
            del self.keys[L+1:], self.ids[L+1:], self.done[L+1:], self.isdel[L+1:]
    def insert(self, nd, nid, isdel):
        if not (nd < math.inf): self.tie = True
        full = self.live == self.ef
        if self.n == self.cap and not (full and not isdel):
            self.tie = True     # no room: the heaps take over
            return
        pos = sum(1 for x in self.keys if x < nd)
        self.keys.insert(pos, nd); self.ids.insert(pos, nid); self.done.insert(pos, False); self.isdel.insert(pos, isdel)
        if not isdel:
            if not full:
                self.live += 1
            else:
                L = self.last_live()             # the largest live entry leaves top_candidates
                self.outside = self.keys[L]
                del self.keys[L], self.ids[L], self.done[L], self.isdel[L]
            if self.live == self.ef:
                self.truncate_after_last_live()
        if self.live > 0:
            self.lower = self.keys[self.last_live()]
    def first_open(self):
        for e, d in enumerate(self.done):
            if not d: return e
        return -1
    def settle(self):
        if self.pending and self.live == self.ef and not (self.lower > self.pend): self.tie = True
        self.pending = False
    def pop(self):
        e = self.first_open()
        if e < 0:
            self.settle(); return None
        dist, node = self.keys[e], self.ids[e]
        if self.pending and dist > self.pend: self.settle()
        self.done[e] = True
        nxt = self.first_open()
        if dist == self.outside: self.tie = True
        if nxt >= 0 and self.keys[nxt] == dist:
            self.pend = max(self.pend, dist) if self.pending else dist
            self.pending = True; self.pop_ties += 1
        return node, dist
    def entry_point(self):
        g, dist = self.g, self.dist
        cur = int(g["entry"]); curdist = dist[cur]
        for level in range(int(g["maxlevel"]), 0, -1):
            changed = True
            while changed:
                changed = False
                block = g["upper"][int(g["upper_off"][cur]) + level - 1]
                for nb in block[1:1 + int(block[0])]:
                    if dist[nb] < curdist:
                        curdist, cur, changed = dist[nb], int(nb), True
        return cur, curdist
    def run(self):
        g, dist, ef = self.g, self.dist, self.ef
        deleted = g["deleted"]
        cur, curdist = self.entry_point()
        if not deleted[cur]:
            self.insert(curdist, cur, False)
        else:
            self.insert(FLT_MAX, cur, True)
        visited = {cur}
        while True:
            got = self.pop()
            if got is None:
                if self.live == ef and self.lower == self.outside: self.tie = True
                break
            node, cdist = got
            if self.tie: break
            if cdist > self.lower and self.live == ef: break
            self.hops += 1
            row = g["links0"][node]
            fresh = []
            for nb in row[1:1 + int(row[0])]:
                nb = int(nb)
                if nb not in visited:
                    visited.add(nb); fresh.append(nb)
            for base in range(0, len(fresh), 64):
                chunk = fresh[base:base + 64]
                admitted = [self.live < ef or self.lower > dist[nb] for nb in chunk]
                if self.live == ef and any(dist[nb] == self.lower for nb in chunk):
                    self.outside = self.lower
                for ok, nb in zip(admitted, chunk):
                    if not ok: continue
                    nd = dist[nb]
                    if self.live < ef or self.lower > nd:
                        self.insert(nd, nb, bool(deleted[nb]))
                        if self.tie: break
                    elif nd == self.lower:
                        self.outside = nd
                if self.tie: break
        lk = [(self.keys[i], self.ids[i]) for i in range(self.n) if not self.isdel[i]]
        keep = min(len(lk), self.k)
        if not self.tie and len(lk) > keep and keep > 0 and lk[keep - 1][0] == lk[keep][0]: self.tie = True
        return lk[:keep]


def build_graph(metric, rows, M, efc):
    from reindexer_amd import hostapi
    n, d = rows.shape
    labels = np.arange(n, dtype=np.uint64) + np.uint64(1000)
    g = hostapi.HnswGraph(metric, d, n, M=M, ef_construction=efc)
    g.add(rows, labels)
    e = g.export()
    e["vectors"] = rows
    g.close()
    return e


def run_model_against_oracle(oracle, metric, rows, queries, M, efc, plans):
    from oracle.pyoracle import oracle_hnsw_search_knn
    g = build_graph(metric, rows, M, efc)
    inv = oracle.l2_modules(rows) if metric == 2 else None
    flagged = total = 0
    cleared = 0   # unflagged searches that met at least one pop tie: the lowerBound argument let them through
    for q in queries:
        if metric == 2:
            q, _ = oracle.normalize_copy(q)
        dist = [float(x) for x in oracle.dist_many(metric, q, rows, inv)]
        for k, ef in plans:
            eff = ef if ef else max(k * 3 // 2, 1)
            wd, wl, _, hops = oracle_hnsw_search_knn(oracle, g, q, k, ef, inv, with_stats=True)
            s = SortedListSearch(g, dist, eff, min(k, g["n"]))
            got = s.run()
            total += 1
            if s.tie:
                flagged += 1
                continue
            mine = sorted((np.float32(d_).view(np.uint32).item(), int(g["labels"][i])) for d_, i in got)
            theirs = sorted((np.float32(d_).view(np.uint32).item(), int(l_)) for d_, l_ in zip(wd, wl))
            assert mine == theirs, (metric, k, ef)
            assert s.hops == hops, (metric, k, ef, s.hops, hops)
            cleared += int(s.pop_ties > 0)
    run_model_against_oracle.cleared = cleared
    return flagged, total


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("shape", [(300, 5, 4, 3), (500, 8, 6, 2), (600, 6, 6, 1), (2000, 10, 8, 4)])
def test_unflagged_searches_on_tie_heavy_grids_equal_the_reference(oracle, metric, shape):
    n, d, M, span = shape
    rng = np.random.default_rng(17 * n + d + metric)
    rows = rng.integers(-span, span + 1, size=(n, d)).astype(np.float32)
    rows[np.all(rows == 0, axis=1)] = 1.0
    queries = [rng.integers(-span, span + 1, size=d).astype(np.float32) for _ in range(60)]
    queries = [q if np.any(q) else np.ones(d, np.float32) for q in queries]
    queries += [rows[i].copy() for i in rng.integers(n, size=20)]
    plans = ((3, 6), (5, 0), (10, 10), (4, 24), (16, 40), (1, 1), (30, 64))
    flagged, total = run_model_against_oracle(oracle, metric, rows, queries, M, 40, plans)
    assert total == 80 * len(plans)
    assert 0 < flagged < total, (flagged, total)   # ties are met AND some searches get through them


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_moderately_repeating_distances(oracle, metric):
    """Coordinates on a quarter grid in 12 dimensions: a search meets a few equal keys, rarely at a place that matters — the regime of
    real float32 corpora, compressed.  Many searches must get through, a good part of them past a pop tie (measured: 112 of the 218
    unflagged L2 searches, 92 of 191 for the inner product, 37 of 598 for cosine)."""
    rng = np.random.default_rng(23 + metric)
    n, d = 1500, 12
    rows = (np.round(rng.standard_normal((n, d)) * 4) / 4).astype(np.float32)
    rows[np.all(rows == 0, axis=1)] = 0.25
    queries = [(np.round(rng.standard_normal(d) * 4) / 4).astype(np.float32) for _ in range(120)]
    queries = [q if np.any(q) else np.full(d, 0.25, np.float32) for q in queries]
    flagged, total = run_model_against_oracle(oracle, metric, rows, queries, 8, 60, ((10, 16), (10, 48), (5, 0), (32, 32), (20, 128)))
    assert flagged < total * 3 // 4, (flagged, total)
    assert run_model_against_oracle.cleared >= 10, run_model_against_oracle.cleared


@pytest.mark.parametrize("metric", [0, 2])
def test_tie_free_data_is_almost_never_flagged(oracle, metric):
    n, d = 1500, 24
    rows = make_corpus(91, n, d)
    queries = list(make_corpus(92, 60, d))
    flagged, total = run_model_against_oracle(oracle, metric, rows, queries, 8, 60, ((10, 32), (5, 0), (40, 64)))
    assert flagged <= total // 20, (flagged, total)


def test_duplicated_rows_and_tiny_graphs(oracle):
    """Some rows twice (equal keys with DIFFERENT labels), every row five times (then every search is flagged: a corpus of exact
    duplicates belongs to the heap kernel), and graphs smaller than ef."""
    rng = np.random.default_rng(5)
    base = rng.standard_normal((240, 6)).astype(np.float32)
    rows = np.ascontiguousarray(np.concatenate([base, base[:60]])[rng.permutation(300)])
    queries = [rng.standard_normal(6).astype(np.float32) for _ in range(40)] + [rows[i].copy() for i in range(10)]
    for metric in (0, 1):
        flagged, total = run_model_against_oracle(oracle, metric, rows, queries, 6, 30, ((5, 8), (10, 30), (20, 0), (300, 256)))
        assert 0 < flagged < total, (flagged, total)
    fives = np.ascontiguousarray(np.repeat(base[:60], 5, axis=0)[rng.permutation(300)])
    flagged, total = run_model_against_oracle(oracle, 0, fives, queries, 6, 30, ((5, 8), (10, 30)))
    assert flagged > total // 2
    for n in (1, 2, 7):
        tiny = rng.integers(-1, 2, size=(n, 4)).astype(np.float32) + np.float32(0.5)
        run_model_against_oracle(oracle, 0, tiny, queries[:10], 4, 10, ((3, 16), (1, 1), (n, 64)))


def run_deleted_model_against_oracle(oracle, metric, rows, queries, M, efc, plans, del_frac, seed):
    """The same comparison for HnswSortedListDel: a random subset of the nodes marked deleted (the entry point among them now and then)."""
    from oracle.pyoracle import oracle_hnsw_search_knn
    g = build_graph(metric, rows, M, efc)
    rng = np.random.default_rng(seed)
    n = g["n"]
    dele = rng.random(n) < del_frac
    if seed % 2:
        dele[int(g["entry"])] = True
    if not dele.any():
        dele[0] = True
    g["deleted"] = dele.astype(np.uint8)
    g["num_deleted"] = int(dele.sum())
    inv = oracle.l2_modules(rows) if metric == 2 else None
    flagged = total = 0
    for q in queries:
        if metric == 2:
            q, _ = oracle.normalize_copy(q)
        dist = [float(x) for x in oracle.dist_many(metric, q, rows, inv)]
        for k, ef in plans:
            eff = ef if ef else max(k * 3 // 2, 1)
            wd, wl, _, hops = oracle_hnsw_search_knn(oracle, g, q, k, ef, inv, with_stats=True)
            cap = 128 if eff <= 96 else 192 if eff <= 160 else 256   # the kernel's 2 / 3 / 4 entries a lane
            s = SortedListSearchDel(g, dist, eff, min(k, n), cap)
            got = s.run()
            total += 1
            if s.tie:
                flagged += 1
                continue
            mine = sorted((np.float32(d_).view(np.uint32).item(), int(g["labels"][i])) for d_, i in got)
            theirs = sorted((np.float32(d_).view(np.uint32).item(), int(l_)) for d_, l_ in zip(wd, wl))
            assert mine == theirs, (metric, k, ef, del_frac)
            assert s.hops == hops, (metric, k, ef, del_frac, s.hops, hops)
    return flagged, total


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("del_frac", [0.02, 0.3, 0.8])
def test_graphs_with_deleted_nodes(oracle, metric, del_frac):
    """Deleted nodes are candidates but never results; lowerBound follows the live entries; a deleted entry point enters at FLT_MAX.
    Gaussian rows (few equal keys: most searches must get through) and a quarter grid (many)."""
    rng = np.random.default_rng(41 + metric)
    n, d = 900, 10
    plans = ((10, 16), (5, 0), (10, 96), (20, 128), (30, 200), (1, 1))
    rows = rng.standard_normal((n, d)).astype(np.float32)
    queries = [rng.standard_normal(d).astype(np.float32) for _ in range(40)]
    flagged, total = run_deleted_model_against_oracle(oracle, metric, rows, queries, 8, 60, plans, del_frac, 7 + metric)
    # the list has room for 32 .. 64 deleted candidates beside its ef live entries: with 30 % of the nodes deleted it runs out for the
    # plans with ef next to a capacity step (a flag like any other: the heaps take over), with 2 % it never does
    limit = {0.02: total // 10, 0.3: total * 6 // 10, 0.8: total - 1}[del_frac]
    assert flagged <= limit, (flagged, total)
    grid = (np.round(rng.standard_normal((n, d)) * 4) / 4).astype(np.float32)
    grid[np.all(grid == 0, axis=1)] = 0.25
    gq = [(np.round(rng.standard_normal(d) * 4) / 4).astype(np.float32) for _ in range(40)]
    gq = [q if np.any(q) else np.full(d, 0.25, np.float32) for q in gq]
    flagged, total = run_deleted_model_against_oracle(oracle, metric, grid, gq, 8, 60, plans, del_frac, 8 + metric)
    assert flagged < total
