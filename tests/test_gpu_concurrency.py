"""-m gpu: the reference's threading contract for reads — any number of query threads search one index concurrently under the namespace's
shared lock (SURVEY §8b "Threading"; the reference test helper runs 4 threads x 20 queries, gtests/tests/unit/float_vector_index.cc:258-294).
Every engine entry must be re-entrant: results of concurrent calls equal the sequential ones, bit for bit."""
import threading

import numpy as np
import pytest

from .conftest import make_corpus

pytestmark = pytest.mark.gpu


def _run_threads(n_threads, fn):
    errs, out = [], [None] * n_threads

    def work(t):
        try:
            out[t] = fn(t)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ths = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    return out


@pytest.mark.parametrize("pruned", [False, True])
def test_concurrent_bruteforce_searches(rxgpu, oracle, monkeypatch, pruned):
    if pruned:
        monkeypatch.setenv("RXGPU_SCAN_BF16", "1")
    n, d, k = 60_000, 128, 10
    rows = make_corpus(51, n, d)
    queries = make_corpus(52, 64, d)
    with rxgpu.VectorIndex("l2", d, n) as ix:
        ix.upload_rows(0, rows)
        want = [ix.search_knn(queries[i:i + 1], k) for i in range(64)]                 # sequential, batch 1
        want_b = ix.search_knn(queries, k)                                             # sequential, one batch

        def fn(t):
            res = []
            for j in range(20):
                i = (t * 7 + j) % 64
                res.append((i, ix.search_knn(queries[i:i + 1], k)))
            res.append(("batch", ix.search_knn(queries, k)))                           # batched path from several threads at once
            return res

        for res in _run_threads(8, fn):
            for i, (dist, row, cnt) in res:
                wd, wr, wc = want_b if i == "batch" else want[i]
                assert np.array_equal(row, wr) and np.array_equal(dist.view(np.uint32), wd.view(np.uint32)) and np.array_equal(cnt, wc)


def test_concurrent_map_hnsw_and_ft(rxgpu, oracle):
    from reindexer_amd import hostapi
    from .test_bm25_oracle import _multi_case
    n, d = 4000, 64
    rows = make_corpus(53, n, d)
    labels = np.arange(n, dtype=np.uint64) << np.uint64(32)
    queries = make_corpus(54, 32, d)
    bf = hostapi.GpuBruteforceMap(0, d, n)
    bf.add(rows, labels)
    hn = hostapi.GpuHnswMap(0, d, n, M=8, ef_construction=100)
    hn.add(rows, labels)
    _, words, avg, removed, excluded, terms, store = _multi_case(3, 3, 3000, 20000, (2, 1), False, None)
    ft = hostapi.GpuFtMerger(3)
    ft.set_docs(words, avg, removed)
    for s in store:
        ft.set_word_fpos(s["word"], s)
    cfg = hostapi.default_ft_config(3)
    gterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    want_bf = [bf.search_knn(q, 10) for q in queries]
    want_hn = [hn.search_knn(q, 10, 64) for q in queries]
    want_ft = ft.merge_query(cfg, gterms, excluded, sort_by_rank=False)

    def fn(t):
        ok = True
        for j in range(12):
            i = (t * 5 + j) % 32
            a, b = bf.search_knn(queries[i], 10), hn.search_knn(queries[i], 10, 64)
            ok &= all(np.array_equal(x, y) for x, y in zip(a, want_bf[i])) and all(np.array_equal(x, y) for x, y in zip(b, want_hn[i]))
            if j % 4 == 0:
                f = ft.merge_query(cfg, gterms, excluded, sort_by_rank=False)
                ok &= all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(f[:4], want_ft[:4]))
            if j % 6 == 0:   # a streaming session of its own per thread
                s = hn.stream(queries[i], 16)
                got = [s.next(5)[1] for _ in range(3)]
                s.close()
                ok &= len(np.unique(np.concatenate(got))) == sum(len(g) for g in got)
        return ok

    assert all(_run_threads(6, fn))
    bf.close()
    hn.close()
    ft.close()


def test_map_query_coalescing_is_transparent(rxgpu, oracle):
    """T threads call GpuBruteforceMap::SearchKnn at once with different k: calls arriving while the device is busy are served by ONE
    batched search; every caller still gets exactly its batch-1 result (ties at the k-th distance included)."""
    from reindexer_amd import hostapi
    rng = np.random.default_rng(8)
    n, d = 30_000, 64
    rows = rng.integers(-2, 3, (n, d)).astype(np.float32)                 # quantised: plenty of exact ties across the k-th boundary
    labels = (rng.permutation(n).astype(np.uint64) << np.uint64(32))
    queries = rng.integers(-2, 3, (48, d)).astype(np.float32)
    m = hostapi.GpuBruteforceMap(0, d, n)
    m.add(rows, labels)
    m.enable_coalescing(False)
    ks = [1, 5, 10, 33]
    want = {(i, k): m.search_knn(queries[i], k) for i in range(48) for k in ks}
    m.enable_coalescing(True)
    b0, q0 = m.coalescing_stats()

    def fn(t):
        ok = True
        for j in range(24):
            i, k = (t * 11 + j) % 48, ks[(t + j) % 4]
            got = m.search_knn(queries[i], k)
            ok &= all(np.array_equal(x, y) for x, y in zip(got, want[(i, k)]))
        return ok

    assert all(_run_threads(12, fn))
    b1, q1 = m.coalescing_stats()
    assert q1 - q0 == 12 * 24 and b1 - b0 < q1 - q0        # fewer device round trips than queries: batches did form
    m.close()


def test_concurrent_hybrid_queries_and_plain_merges_on_one_text_index(rxgpu, oracle):
    """Several planner threads run HYBRID queries (resident FT merge -> prepare -> resident KNN search -> fusion: four C-ABI calls each)
    on one text index + one vector index while other threads run ordinary FT merges on the same text index.  A resident merge is a session
    of its thread until that thread's fusion (ordinary merges take the handle's other lanes, another thread's resident merge waits), and every
    thread has its own resident KNN buffers: each fused list must be the one the same query gives when it runs alone."""
    from reindexer_amd import hostapi
    from .test_bm25_oracle import _multi_case
    n_docs, d = 6000, 48
    total = n_docs + 1
    _, words, avg, removed, excluded, terms_all, store = _multi_case(91, 1, total, 20000, (1, 1, 1, 1), False, None, sizes=(300, 2000), nsub_range=(1, 3))
    ftm = hostapi.GpuFtMerger(1)
    ftm.set_docs(words, avg, None)
    for s in store:
        ftm.set_word_fpos(s["word"], s)
    rows = make_corpus(92, total, d)
    vm = hostapi.GpuBruteforceMap(2, d, total)
    vm.add(rows, np.arange(total, dtype=np.uint64) << np.uint64(32))
    keys = make_corpus(93, 12, d)
    cfg = hostapi.default_ft_config(1)
    gterms = [dict(op=1, opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms_all]
    queries = [[gterms[i % 4]] if i % 3 == 0 else [gterms[i % 4], gterms[(i + 1) % 4]] for i in range(12)]

    def hybrid(i):
        return hostapi.hybrid_query_resident(vm, ftm, cfg, queries[i], keys[i], 20, kind="rrf", params=[60.0], union=True, desc=True)

    want_h = [hybrid(i) for i in range(12)]
    want_m = [ftm.merge_query(cfg, queries[i], None, sort_by_rank=False) for i in range(12)]
    assert all(len(w[0]) > 20 for w in want_h)

    def fn(t):
        ok = True
        for j in range(10):
            i = (t * 5 + j) % 12
            if t < 4:   # hybrid callers
                got = hybrid(i)
                ok &= np.array_equal(got[0], want_h[i][0]) and np.array_equal(got[1].view(np.uint32), want_h[i][1].view(np.uint32))
            else:       # plain merges on the same text index meanwhile
                got = ftm.merge_query(cfg, queries[i], None, sort_by_rank=False)
                ok &= all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(got[:4], want_m[i][:4]))
        return ok

    assert all(_run_threads(7, fn))
    vm.close()
    ftm.close()


def test_resident_contexts_follow_the_live_threads(rxgpu):
    """rxgpu_search_knn_resident keeps a stream + result buffers per calling thread.  Planner threads are short-lived: a thread that ends
    hands its context back to the index's pool, so 40 threads that searched one after another leave no context behind (and the pool did
    not grow to 40 either: each newcomer takes over what its predecessor returned); threads that are alive at once each hold their own."""
    import torch
    n, d, k = 20_000, 64, 5
    rows = make_corpus(91, n, d)
    queries = make_corpus(92, 48, d)
    lib = rxgpu.lib()
    with rxgpu.VectorIndex("l2", d, n) as ix:
        ix.upload_rows(0, rows)
        want = [ix.search_knn(queries[i:i + 1], k) for i in range(48)]

        def one(i):
            dd, dr, dc, st, cnt = ix.search_knn_resident(queries[i], k)
            torch.cuda.synchronize()   # the search was only enqueued
            return cnt

        for i in range(40):   # one after another, each in a thread of its own
            assert _run_threads(1, lambda t, i=i: one(i))[0] == k
            assert lib.rxgpu_index_resident_contexts(ix._h) == 0
        assert one(40) == k and lib.rxgpu_index_resident_contexts(ix._h) == 1   # the calling thread stays alive and keeps its context
        gate = threading.Barrier(8)

        def alive(t):
            one(t)
            gate.wait()
            held = lib.rxgpu_index_resident_contexts(ix._h)
            gate.wait()
            return held

        assert all(h == 9 for h in _run_threads(8, alive))   # eight live threads + the main one
        assert lib.rxgpu_index_resident_contexts(ix._h) == 1
        # and the plain searches still give the sequential results out of the recycled contexts
        for i in (0, 17, 47):
            dist, row, cnt = ix.search_knn(queries[i:i + 1], k)
            assert np.array_equal(row, want[i][1]) and np.array_equal(dist.view(np.uint32), want[i][0].view(np.uint32))
