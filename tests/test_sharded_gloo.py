"""CPU, world_size = 2, gloo: the N > 1 path (per-shard top-k -> all-gather -> merge) must return exactly what one
index over the whole corpus returns.  The local scan is the CPU oracle here; on GPUs it is the rxgpu shard."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from .conftest import lex_topk, make_corpus


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, d, kk, nq, ties, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pyoracle import Oracle
    from reindexer_amd.sharded import ShardedBruteforce
    orc = Oracle()
    rows = _corpus(n, d, ties)
    shard = n // world
    mine = rows[rank * shard:(rank + 1) * shard]

    def local_search(queries, k):
        ds, rs = [], []
        for q in queries.numpy():
            alld = orc.dist_many(1, q, mine)
            dd, rr = lex_topk(alld, k)
            pad = k - dd.shape[0]
            ds.append(np.concatenate([dd, np.full(pad, np.inf, np.float32)]))
            rs.append(np.concatenate([rr.astype(np.int64), np.full(pad, 0xFFFFFFFF, np.int64)]))
        return torch.from_numpy(np.stack(ds)), torch.from_numpy(np.stack(rs))

    sb = ShardedBruteforce(local_search, shard)
    queries = torch.from_numpy(_queries(nq, d, ties))
    dd, rr = sb.search(queries, kk)
    out_q.put((rank, dd.numpy().copy(), rr.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def _corpus(n, d, ties):
    if ties:
        return np.random.default_rng(1).integers(-1, 2, (n, d)).astype(np.float32)
    return make_corpus(1, n, d)


def _queries(nq, d, ties):
    if ties:
        return np.random.default_rng(2).integers(-1, 2, (nq, d)).astype(np.float32)
    return make_corpus(2, nq, d)


@pytest.mark.parametrize("ties", [False, True])
def test_two_rank_merge_equals_single_index(oracle, ties):
    world, n, d, kk, nq = 2, 4000, (8 if ties else 64), 11, 6
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, d, kk, nq, ties, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out_q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows, queries = _corpus(n, d, ties), _queries(nq, d, ties)
    for rank, dd, rr in results:
        for qi in range(nq):
            wd, wr = lex_topk(oracle.dist_many(1, queries[qi], rows), kk)
            assert np.array_equal(rr[qi], wr.astype(np.int64)), (rank, qi)
            assert np.array_equal(dd[qi].view(np.uint32), wd.view(np.uint32))


def test_merge_handles_short_shards():
    """A shard with fewer than kk rows pads with the invalid row; padding must sort last and come back as -1."""
    from reindexer_amd.sharded import merge_shard_topk, pack_topk
    d0 = torch.tensor([[0.5, 2.0, float("inf")]])
    r0 = torch.tensor([[3, 1, 0xFFFFFFFF]])
    d1 = torch.tensor([[-1.0, float("inf"), float("inf")]])
    r1 = torch.tensor([[0, 0xFFFFFFFF, 0xFFFFFFFF]])
    g = torch.stack([pack_topk(d0, r0), pack_topk(d1, r1)])  # [world=2, nq=1, kk=3]
    m = merge_shard_topk(g, 4, shard_rows=10)
    assert m[0, :, 1].tolist() == [10, 3, 1, -1]
    assert m[0, :3, 0].to(torch.int32).view(torch.float32).tolist() == [-1.0, 0.5, 2.0]


def _hnsw_worker(rank, world, port, n, d, kk, nq, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pyoracle import Oracle, oracle_hnsw_search_knn
    from reindexer_amd import hostapi
    from reindexer_amd.sharded import ShardedBruteforce
    orc = Oracle()
    rows = make_corpus(5, n, d)
    shard = n // world
    mine = rows[rank * shard:(rank + 1) * shard]
    graph = hostapi.HnswGraph(0, d, shard, M=8, ef_construction=100)        # per-shard independent graph
    graph.add(mine, np.arange(shard, dtype=np.uint64))
    g = graph.export()
    g["vectors"] = mine

    def local_search(queries, k):
        ds, rs = [], []
        for q in queries.numpy():
            dd, ll = oracle_hnsw_search_knn(orc, g, q, k, 64)               # best first, label == shard-local row here
            pad = k - dd.shape[0]
            ds.append(np.concatenate([dd, np.full(pad, np.inf, np.float32)]))
            rs.append(np.concatenate([ll.astype(np.int64), np.full(pad, 0xFFFFFFFF, np.int64)]))
        return torch.from_numpy(np.stack(ds)), torch.from_numpy(np.stack(rs))

    sb = ShardedBruteforce(local_search, shard)
    dd, rr = sb.search(torch.from_numpy(make_corpus(6, nq, d)), kk)
    out_q.put((rank, dd.numpy().copy(), rr.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_hnsw_shards_merge(oracle):
    """Per-shard HNSW graphs + the same exchange: both ranks hold the identical merged list, every hit carries its exact distance, and
    recall vs exact search over the whole corpus is at least what a graph search gives (>= 0.9 here at ef=64)."""
    world, n, d, kk, nq = 2, 3000, 32, 10, 12
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hnsw_worker, args=(r, world, port, n, d, kk, nq, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((out_q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(results[0][1].view(np.uint32), results[1][1].view(np.uint32)) and np.array_equal(results[0][2], results[1][2])
    rows, queries = make_corpus(5, n, d), make_corpus(6, nq, d)
    hits = 0
    for qi in range(nq):
        alld = oracle.dist_many(0, queries[qi], rows)
        rr, dd = results[0][2][qi], results[0][1][qi]
        assert np.array_equal(dd.view(np.uint32), alld[rr].view(np.uint32))           # global row = rank * shard_rows + local row
        assert np.all(np.diff(dd) >= 0)
        hits += len(set(rr.tolist()) & set(lex_topk(alld, kk)[1].tolist()))
    assert hits / (nq * kk) >= 0.9


def _subset_worker(rank, world, port, n, d, kk, nq, dens, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pyoracle import Oracle
    from reindexer_amd.sharded import ShardedBruteforce
    orc = Oracle()
    rows = make_corpus(1, n, d)
    shard = n // world
    mine = rows[rank * shard:(rank + 1) * shard]

    def local_subset(queries, k, local_rows):
        ds, rs = [], []
        for q in queries.numpy():
            alld = orc.dist_many(0, q, mine[local_rows]) if local_rows.size else np.empty(0, np.float32)
            dd, pos = lex_topk(alld, k)
            pad = k - dd.shape[0]
            ds.append(np.concatenate([dd, np.full(pad, np.inf, np.float32)]))
            rs.append(np.concatenate([local_rows[pos].astype(np.int64), np.full(pad, 0xFFFFFFFF, np.int64)]))
        return torch.from_numpy(np.stack(ds)), torch.from_numpy(np.stack(rs))

    allowed = np.flatnonzero(np.random.default_rng(3).random(n) < dens)
    sb = ShardedBruteforce(None, shard)
    dd, rr = sb.search_subset(torch.from_numpy(make_corpus(2, nq, d)), kk, allowed, local_subset)
    out_q.put((rank, dd.numpy().copy(), rr.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dens", [0.3, 0.002])
def test_two_rank_prefiltered_search_equals_single_index(oracle, dens):
    """The pre-filtered search over two shards == the pre-filtered search over one index: every rank scans only the allowed rows of its own
    shard (one shard may hold fewer than kk, or none), the exchange and the merge are the unfiltered ones."""
    world, n, d, kk, nq = 2, 4000, 32, 7, 5
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subset_worker, args=(r, world, port, n, d, kk, nq, dens, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out_q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows, queries = make_corpus(1, n, d), make_corpus(2, nq, d)
    allowed = np.flatnonzero(np.random.default_rng(3).random(n) < dens)
    for rank, dd, rr in results:
        for qi in range(nq):
            wd, pos = lex_topk(oracle.dist_many(0, queries[qi], rows[allowed]), kk)
            m = wd.shape[0]
            assert np.array_equal(rr[qi, :m], allowed[pos].astype(np.int64)), (rank, qi)
            assert np.array_equal(dd[qi, :m].view(np.uint32), wd.view(np.uint32))
            assert np.all(rr[qi, m:] == -1)


def _labels_worker(rank, world, port, n, d, k, nq, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pyoracle import Oracle
    from reindexer_amd.sharded import ShardedBruteforce
    orc = Oracle()
    rows, labels = _corpus(n, d, True), _perm_labels(n)
    shard = n // world
    mine, mine_labels = rows[rank * shard:(rank + 1) * shard], labels[rank * shard:(rank + 1) * shard]

    def local_search(queries, kk):
        ds, rs = [], []
        for q in queries.numpy():
            dd, rr = lex_topk(orc.dist_many(0, q, mine), kk)
            pad = kk - dd.shape[0]
            ds.append(np.concatenate([dd, np.full(pad, np.inf, np.float32)]))
            rs.append(np.concatenate([rr.astype(np.int64), np.full(pad, 0xFFFFFFFF, np.int64)]))
        return torch.from_numpy(np.stack(ds)), torch.from_numpy(np.stack(rs))

    def local_range(q, radius):
        alld = orc.dist_many(0, q.numpy(), mine)
        hit = np.flatnonzero(alld <= radius)
        return alld[hit], hit

    sb = ShardedBruteforce(local_search, shard)
    res = sb.search_knn_labels(torch.from_numpy(_queries(nq, d, True)), k, mine_labels, local_range)
    out_q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def _perm_labels(n):
    return (np.random.default_rng(9).permutation(n).astype(np.uint64) << np.uint64(32)) | np.uint64(1)


@pytest.mark.parametrize("k", [7, 40])
def test_two_rank_label_search_replays_cross_shard_ties(oracle, k):
    """The (dist, label) semantics of BruteforceSearch::SearchKnn over a 2-rank sharded corpus with massive exact ties and labels that are
    not in row order: the k-th boundary cuts through a group of equal distances spread over both shards; every rank must return what the
    restated engine (pinned to the real one) returns over the whole corpus."""
    world, n, d, nq = 2, 3000, 8, 8
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_labels_worker, args=(r, world, port, n, d, k, nq, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out_q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows, labels, queries = _corpus(n, d, True), _perm_labels(n), _queries(nq, d, True)
    ties_seen = 0
    for rank, res in results:
        for qi in range(nq):
            wd, wl = oracle.bf_search_knn(0, rows, labels, None, queries[qi], k)
            got_d = np.array([p[0] for p in res[qi]], np.float32)
            got_l = np.array([p[1] for p in res[qi]], np.uint64)
            order = np.lexsort((wl, wd))
            assert np.array_equal(got_l, wl[order]), (rank, qi)
            assert np.array_equal(got_d.view(np.uint32), wd[order].view(np.uint32))
            alld = np.sort(oracle.dist_many(0, queries[qi], rows))
            ties_seen += int(alld[k - 1] == alld[k])
    assert ties_seen > 0


# ---------------------------------------------------------------------------------------------- ft_fast over document-range shards
def _ft_case():
    from .test_bm25_oracle import _multi_case
    nf, total = 1, 40_000
    _, words, avg, removed, excluded, terms, store = _multi_case(6161, nf, total, 900, (1, 2, 1, 3), False, None, sizes=(3000, 14_000), nsub_range=(2, 4))
    return nf, total, words, avg, removed, terms


def _ft_worker(rank, world, port, limit, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pyoracle import FtOracle, Oracle
    from reindexer_amd.sharded import ShardedFtExchange
    ft = FtOracle(Oracle())
    nf, total, words, avg, removed, terms = _ft_case()
    cfg = ft.default_config(nf, merge_limit=limit)
    # this rank's documents: a contiguous run of 8192-document ranges, like rxgpu_ft_create_sharded cuts them
    n_ranges = (total + 8191) // 8192
    per = -(-n_ranges // world)
    lo, hi = min(rank * per * 8192, total), min((rank + 1) * per * 8192, total)
    # ---- local facts from the postings of MY documents (buildRestrictingBitmask, calcTermScores: every field has the same boost here)
    mask = np.ones(total, bool)
    score = np.zeros(total, np.int64)
    first_row = np.full(total, -1, np.int64)
    row, total_vids = 0, 0
    for t in terms:
        held = np.zeros(total, bool)
        scored = np.zeros(total, bool)
        for s in t["subs"]:
            total_vids += len(s["doc"])
            docs = s["doc"][(s["doc"] >= lo) & (s["doc"] < hi)]
            held[docs] = True
            if t["op"] != 3:
                new = docs[~scored[docs]]
                p16 = min(int(np.float32(s["proc"]) * np.float32(t["opts"]["field_boost"][0]) * np.float32(t["opts"]["boost"])) & 0xFFFF, 65535 // 4)
                score[new] = np.minimum(score[new] + p16, 65535)
                scored[new] = True
                fresh = docs[first_row[docs] < 0]
                first_row[fresh] = row
                row += 1
        if t["op"] == 2:
            mask &= held
        elif t["op"] == 3:
            mask &= ~held
    n_rows = row
    mine = np.zeros(total, bool)
    mine[lo:hi] = True
    mask &= mine
    elig = mask & (removed == 0)
    score[~elig] = 0
    hist = np.bincount(score[lo:hi][score[lo:hi] > 0], minlength=65536).astype(np.int64)
    max_merged = min(limit, total_vids)
    est_or = sum(sum(len(s["doc"]) for s in t["subs"]) for t in terms if t["op"] == 1)
    est_and = min([sum(len(s["doc"]) for s in t["subs"]) for t in terms if t["op"] == 2] or [1 << 62])
    host_gate = min(est_or, est_and, total) > limit and total > limit   # estimateNumDocsInMerge (merger.h:239-267) + mergerimpl.h:486-490
    x = ShardedFtExchange()
    on, min_score, quota = x.preselect(torch.from_numpy(hist), int(mask.sum()), limit, max_merged, host_gate)
    kept = elig & (first_row >= 0)
    if on:
        ties = np.flatnonzero(elig & (score == min_score))
        kept = elig & (score > min_score)
        kept[ties[:quota]] = True
        kept &= first_row >= 0
    counts = np.bincount(first_row[kept], minlength=n_rows).astype(np.int64)
    bases, total_docs = x.slot_bases(torch.from_numpy(counts))
    # ---- the local merge of the kept documents: the oracle over the whole index with everything else excluded (ranks use the global N / df)
    wd, wp, wf, wn, _ = ft.merge_query(ft.default_config(nf, merge_limit=1 << 30), terms, total, words, avg, removed, (~kept).astype(np.uint8), sort_by_rank=False)
    # ... arrives in (row, document) order: the k-th document of row r sits at slot bases[r] + k
    rows_of = first_row[wd.astype(np.int64)]
    slots = np.empty(len(wd), np.int64)
    for r in range(n_rows):
        sel = np.flatnonzero(rows_of == r)
        slots[sel] = int(bases[r]) + np.arange(len(sel))
    out_q.put((rank, slots, wd.astype(np.int64), wp.copy(), wn.copy(), bool(on), total_docs, x.collectives))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("limit", [20000, 900, 150])
def test_two_rank_ft_document_range_shards_equal_single_index(oracle, limit):
    """SURVEY 8e "BM25" as a multi-process path: two ranks hold the postings of their document ranges, exchange the folded pre-score
    histograms + mask popcounts and the first-met counts (reindexer_amd.sharded.ShardedFtExchange, two all_gathers), and write their documents
    at global merge slots — the union is the single index's merge, slot for slot: documents, raw ranks, uint8 ranks, preselect flag (with the
    ties at the threshold handed out in shard order: limit 900 / 150), the mergeLimit cut (limit 150)."""
    from oracle.pyoracle import FtOracle
    world = 2
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ft_worker, args=(r, world, port, limit, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out_q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ft = FtOracle(oracle)
    nf, total, words, avg, removed, terms = _ft_case()
    cfg = ft.default_config(nf, merge_limit=limit)
    wd, wp, wf, wn, wpre = ft.merge_query(cfg, terms, total, words, avg, removed, None, sort_by_rank=False)
    n = len(wd)
    docs = np.full(n, -1, np.int64)
    procs_ = np.zeros(n, np.float32)
    for rank, slots, d, pr, nm, on, total_docs, collectives in results:
        assert on == bool(wpre) and min(total_docs, limit) == n, (rank, on, wpre, total_docs, n)
        assert collectives in (1, 2) and (collectives == 2 or not on)   # one all_gather for the slots, one more when the host gate held
        keep = slots < n   # the mergeLimit cut: slots at or beyond maxMergedDocs are never added
        docs[slots[keep]] = d[keep]
        procs_[slots[keep]] = pr[keep]
    assert np.array_equal(docs, wd.astype(np.int64))
    assert np.array_equal(procs_.view(np.uint32), wp.view(np.uint32))


def _phrase_cut_case(seed, world):
    rng = np.random.default_rng(seed)
    rows, docs_per_rank = 5, 300
    total = world * docs_per_rank
    postings = rng.random((rows, total)) < 0.4
    candidate = postings & (rng.random((rows, total)) < 0.7)
    seen = np.zeros(total, bool)
    for r in range(rows):   # a document is added by the first row of the phrase's first term that can add it
        candidate[r] &= ~seen
        seen |= candidate[r]
    return rows, docs_per_rank, postings, candidate


def _phrase_cut_worker(rank, world, port, seed, limits, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from reindexer_amd.sharded import ShardedFtExchange
    rows, per, postings, candidate = _phrase_cut_case(seed, world)
    mine = slice(rank * per, (rank + 1) * per)
    x = ShardedFtExchange()
    got = []
    for limit in limits:
        left = min(limit, int(postings[:, mine].sum()))   # the rank's own admission stops at its local bound (ft_phrase_admit)
        counts = np.zeros(rows, np.int64)
        for r in range(rows):
            counts[r] = min(int(candidate[r, mine].sum()), left)
            left -= counts[r]
        keep, merged = x.phrase_cut(torch.from_numpy(counts), limit)
        got.append((keep, merged, int(counts.sum())))
    out_q.put((rank, got, x.collectives))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_phrase_admission_cut_over_ranks_equals_the_plain_rule(world):
    """PhraseMerger's admission cut (phrasemerger.h:341) as a multi-process exchange: every rank admits the candidates of its documents under its
    local bound, ONE all_gather of the per-row counts, every rank derives what it keeps — a prefix of its own slots.  Against the plain rule:
    the first mergeLimit candidates of the whole index in (row, document) order."""
    seed = 99 + world
    rows, per, postings, candidate = _phrase_cut_case(seed, world)
    n_cand = int(candidate.sum())
    limits = [1, 17, n_cand // 3, n_cand - 1, n_cand, n_cand + 9]
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_phrase_cut_worker, args=(r, world, port, seed, limits, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(out_q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    order = [(r, d) for r in range(rows) for d in np.flatnonzero(candidate[r])]
    for li, limit in enumerate(limits):
        want = np.zeros(world, np.int64)
        for r, d in order[:limit]:
            want[d // per] += 1
        for rank, got, collectives in results:
            keep, merged, admitted = got[li]
            assert keep == want[rank] and merged == min(limit, n_cand) and keep <= admitted, (limit, rank, keep, want[rank])
            assert collectives == len(limits)
