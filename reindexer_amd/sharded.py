"""Row-range sharding of a float_vector index over the GPUs of one node.

One process per GPU.  Every rank holds rows [rank*shard_rows, (rank+1)*shard_rows) of the corpus as its own
rxgpu index and receives the same query.  The only exchange is one all-gather per query batch of the per-shard top-kk
(kk * 8 bytes per rank per query: f32 distance + u32 shard-local row) over RCCL/xGMI — latency-bound, so everything for
one batch goes in ONE collective.  Each rank then merges with the same total order the single-index engine uses:
(dist, global row) ascending == the eviction order of the reference's (dist,label) max-heap
(hnswlib/bruteforce.cc:103-127, priority_queue.h), with global row = rank * shard_rows + local row.
The reference has no counterpart (it has no device notion); the merged result equals what one index over the whole
corpus returns, which is what the tests assert.

torch / torch.distributed are plumbing here: device memory, streams and the collective.
"""
from __future__ import annotations

import torch


def _orderable_i32(dist: torch.Tensor) -> torch.Tensor:
    """float32 -> int32 whose signed order equals the float order; -0.0 and +0.0 map to the same value."""
    bits = (dist + 0.0).contiguous().view(torch.int32)
    return bits ^ ((bits >> 31) & 0x7FFFFFFF)


def pack_topk(dist: torch.Tensor, row: torch.Tensor, row_base: int = 0) -> torch.Tensor:
    """Pack a shard-local top list for the wire: int64 [..., kk] = (dist bits << 32) | local row (row_base unused here,
    kept for symmetry; the rank index supplies the base after the gather)."""
    del row_base
    d = dist.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    r = row.to(torch.int64) & 0xFFFFFFFF
    return (d << 32) | r


def unpack_topk(packed: torch.Tensor):
    d = ((packed >> 32) & 0xFFFFFFFF).to(torch.int32).view(torch.float32)  # low 32 bits reinterpret
    r = packed & 0xFFFFFFFF
    return d, r


def merge_shard_topk(gathered: torch.Tensor, kk: int, shard_rows: int | None = None, invalid_row: int = 0xFFFFFFFF) -> torch.Tensor:
    """gathered: int64 [world, kk] (one query) or [world, nq, kk] of pack_topk() lists.
    Returns int64 [kk] / [nq, kk]: (dist bits << 32 | ...) is not order-preserving, so the result is re-packed as
    [..., kk, 2] -> (dist bits, global row) pairs in two int64 lanes."""
    single = gathered.dim() == 2
    g = gathered.unsqueeze(1) if single else gathered           # [world, nq, kk]
    world, nq, k_in = g.shape
    dist_bits = ((g >> 32) & 0xFFFFFFFF).to(torch.int32)
    local_row = g & 0xFFFFFFFF
    valid = local_row != invalid_row
    if shard_rows is None:
        shard_rows = 1 << 32
    base = torch.arange(world, device=g.device, dtype=torch.int64).view(world, 1, 1) * shard_rows
    global_row = local_row + base
    key_hi = _orderable_i32(dist_bits.view(torch.float32)).to(torch.int64)
    # sort key: (orderable dist, global row); invalid entries last
    key_hi = torch.where(valid, key_hi, torch.full_like(key_hi, 1 << 31))
    flat_hi = key_hi.permute(1, 0, 2).reshape(nq, world * k_in)
    flat_row = global_row.permute(1, 0, 2).reshape(nq, world * k_in)
    flat_bits = dist_bits.permute(1, 0, 2).reshape(nq, world * k_in).to(torch.int64) & 0xFFFFFFFF
    # two-pass stable sort = lexicographic (hi, row)
    o1 = torch.argsort(flat_row, dim=1, stable=True)
    hi1 = torch.gather(flat_hi, 1, o1)
    o2 = torch.argsort(hi1, dim=1, stable=True)
    order = torch.gather(o1, 1, o2)[:, :kk]
    out = torch.stack([torch.gather(flat_bits, 1, order), torch.gather(flat_row, 1, order)], dim=-1)  # [nq, kk, 2]
    bad = torch.gather(flat_hi, 1, order) == (1 << 31)
    out[..., 1] = torch.where(bad, torch.full_like(out[..., 1], -1), out[..., 1])
    return out[0] if single else out


class ShardedBruteforce:
    """Row-sharded KNN: local search on this rank's GPU (brute-force scan, or an HNSW graph over the shard's rows) + one all-gather +
    merge.

    local_search(queries[nq,dim] tensor, kk) -> (dist[nq,kk] f32 tensor, row[nq,kk] int tensor) is injected so the
    exchange/merge logic runs unchanged on CPU under gloo in the tests; on a GPU box it is the rxgpu index of this rank.
    """

    def __init__(self, local_search, shard_rows: int, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.local_search = local_search
        self.shard_rows = int(shard_rows)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def search(self, queries: torch.Tensor, kk: int):
        """-> (dist[nq,kk] f32, global_row[nq,kk] i64) identical on every rank."""
        d, r = self.local_search(queries, kk)
        return self._exchange(d, r, kk)

    def search_subset(self, queries: torch.Tensor, kk: int, global_rows, local_search_subset):
        """Pre-filtered form (`WHERE cond AND KNN(...)` over a sharded index): `global_rows` = sorted global row numbers allowed to compete,
        the same on every rank.  Each rank keeps the part that falls into its shard and scans only those rows
        (local_search_subset(queries, kk, local_rows) -> (dist, row) padded with +inf / 0xFFFFFFFF past the allowed rows); the exchange and the
        merge are the ones of search().  Equals the pre-filtered search over one index holding the whole corpus."""
        import numpy as np
        rows = np.ascontiguousarray(global_rows, dtype=np.int64)
        lo = self.rank * self.shard_rows
        a, b = np.searchsorted(rows, [lo, lo + self.shard_rows])
        d, r = local_search_subset(queries, kk, (rows[a:b] - lo).astype(np.uint32))
        return self._exchange(d, r, kk)

    def search_knn_labels(self, queries: torch.Tensor, k: int, local_labels, local_range):
        """SearchKnn with the reference's (dist, LABEL) semantics over the sharded corpus (bruteforce.cc:103-127): -> per query a list of
        (dist f32, label u64) best first, identical on every rank and identical to hnswlib::BruteforceSearch over the whole corpus.

        The merge order (dist, global row) equals the reference's result unless an exact distance tie straddles the k-th boundary: the
        reference then keeps the tied rows its heap keeps — admission is strict (`dist < worst`), eviction removes the largest (dist, label).
        Exactly as the single-device Map does (gpu_bruteforce_map.cc, replayTies) the rule is then replayed over every row with
        dist <= d_k: each rank fetches those rows of its shard (local_range(query, d_k) -> (dist, local rows), inclusive) with their labels
        (local_labels: this shard's labels, indexable by local row), the lists travel in one more all-gather, and every rank replays the scan
        in global row order.  local_labels / local_range are injected like local_search."""
        import numpy as np
        kk = k + 1
        d, r = self.search(queries, kk)
        d_np, r_np = d.cpu().numpy(), r.cpu().numpy()
        nq = d_np.shape[0]
        lo = self.rank * self.shard_rows
        need = [qi for qi in range(nq) if r_np[qi, k] >= 0 and not (d_np[qi, k - 1] < d_np[qi, k])] if k >= 1 else []
        # labels of the merged rows: every rank contributes the labels of the rows it owns (one small all-reduce-by-gather)
        lab = np.zeros((nq, kk), np.int64)
        own = (r_np >= lo) & (r_np < lo + self.shard_rows)
        lab[own] = np.asarray(local_labels)[(r_np[own] - lo).astype(np.int64)].astype(np.uint64).view(np.int64)
        lab_t = torch.from_numpy(lab).to(queries.device if queries.is_cuda else "cpu")
        if self.world > 1:
            self._dist.all_reduce(lab_t, op=self._dist.ReduceOp.SUM, group=self.group)   # exactly one rank owns each row: the sum is its label
        lab = lab_t.cpu().numpy().view(np.uint64)
        out = []
        replay = {}
        if need:
            # (dist, global row, label) of every row with dist <= d_k on this shard, for all queries that need it, in one padded exchange
            mine = []
            for qi in need:
                rd, rr = local_range(queries[qi], float(d_np[qi, k - 1]))
                rd = np.asarray(rd, np.float32)
                rr = np.asarray(rr, np.int64)
                ll = np.asarray(local_labels)[rr].astype(np.uint64).view(np.int64)
                mine.append(np.stack([rd.view(np.int32).astype(np.int64), rr + lo, ll], axis=1))
            # the exchange tensors live where the group's backend wants them: RCCL rejects CPU tensors, gloo takes either
            xdev = lab_t.device
            counts = torch.tensor([m.shape[0] for m in mine], dtype=torch.int64, device=xdev)
            all_counts = [torch.zeros_like(counts) for _ in range(self.world)]
            if self.world > 1:
                self._dist.all_gather(all_counts, counts, group=self.group)
            else:
                all_counts = [counts]
            width = int(max(int(c.max()) for c in all_counts)) if need else 0
            buf = torch.zeros((len(need), max(width, 1), 3), dtype=torch.int64)
            for j, m in enumerate(mine):
                buf[j, : m.shape[0]] = torch.from_numpy(m)
            buf = buf.to(xdev)
            bufs = [torch.zeros_like(buf) for _ in range(self.world)]
            if self.world > 1:
                self._dist.all_gather(bufs, buf, group=self.group)
            else:
                bufs = [buf]
            bufs = [b.cpu() for b in bufs]
            all_counts = [c.cpu() for c in all_counts]
            for j, qi in enumerate(need):
                rows = np.concatenate([bufs[w][j, : int(all_counts[w][j])].numpy() for w in range(self.world)], axis=0)
                rows = rows[np.argsort(rows[:, 1], kind="stable")]                       # scan order = global row order
                dk = d_np[qi, k - 1]
                better, ties = [], []                                                     # ties: labels, the largest is evicted first
                for bits, _, label in rows:
                    dist_v = np.array([bits], np.int64).astype(np.int32).view(np.float32)[0]
                    is_tie = not (dist_v < dk)
                    ulabel = int(np.array([label], np.int64).view(np.uint64)[0])
                    if len(better) + len(ties) < k:
                        (ties if is_tie else better).append(ulabel if is_tie else (dist_v, ulabel))
                    elif not is_tie:
                        ties.remove(max(ties))
                        better.append((dist_v, ulabel))
                res = better + [(dk, t) for t in ties]
                res.sort(key=lambda p: (p[0], p[1]))
                replay[qi] = res
        for qi in range(nq):
            if qi in replay:
                out.append(replay[qi])
            else:
                n = int((r_np[qi, :k] >= 0).sum())
                res = [(d_np[qi, j], int(lab[qi, j])) for j in range(n)]
                res.sort(key=lambda p: (p[0], p[1]))
                out.append(res)
        return out

    def _exchange(self, d: torch.Tensor, r: torch.Tensor, kk: int):
        packed = pack_topk(d, r)                                  # [nq, kk]
        if self.world > 1:
            gathered = torch.empty((self.world,) + tuple(packed.shape), dtype=torch.int64, device=packed.device)
            self._dist.all_gather_into_tensor(gathered.view(-1), packed.contiguous().view(-1), group=self.group)
        else:
            gathered = packed.unsqueeze(0)
        merged = merge_shard_topk(gathered, kk, self.shard_rows)  # [nq, kk, 2]
        dist_out = merged[..., 0].to(torch.int32).view(torch.float32)
        return dist_out, merged[..., 1]


def rxgpu_local_search(index, device):
    """Adapter: an rxgpu VectorIndex shard as the local_search of ShardedBruteforce (device tensors in/out)."""
    def run(queries: torch.Tensor, kk: int):
        q = queries.to(device=device, dtype=torch.float32).contiguous()
        nq = q.shape[0]
        d = torch.empty((nq, kk), dtype=torch.float32, device=device)
        r = torch.empty((nq, kk), dtype=torch.int32, device=device)
        stream = torch.cuda.current_stream(device)
        index.search_knn_device(q.data_ptr(), nq, kk, d.data_ptr(), r.data_ptr(), None, stream.cuda_stream)
        return d, r
    return run


def rxgpu_local_search_subset(index, device=None):
    """Adapter: the pre-filtered scan of an rxgpu VectorIndex shard (rxgpu_search_knn_subset) as the local_search_subset of
    ShardedBruteforce.search_subset; entries past the number of allowed rows come back as (+inf, invalid row)."""
    import numpy as np

    def run(queries: torch.Tensor, kk: int, local_rows):
        q = queries.detach().to("cpu", torch.float32).contiguous().numpy()
        dist, row, cnt = index.search_knn_subset(q, kk, local_rows)
        d = np.full((q.shape[0], kk), np.inf, np.float32)
        r = np.full((q.shape[0], kk), 0xFFFFFFFF, np.int64)
        for i in range(q.shape[0]):
            c = int(cnt[i])
            d[i, :c], r[i, :c] = dist[i, :c], row[i, :c]
        dt, rt = torch.from_numpy(d), torch.from_numpy(r)
        return (dt.to(device), rt.to(device)) if device is not None else (dt, rt)
    return run


def rxgpu_hnsw_local_search(index, device=None):
    """Adapter: an rxgpu VectorIndex shard with an attached HNSW graph (per-shard independent graphs + the same all-gather
    merge).  Rows are the shard's internal ids; queries must already be normalised for cosine.  ef is fixed per adapter call."""
    import numpy as np

    def run(queries: torch.Tensor, kk: int, ef: int = 0):
        q = queries.detach().to("cpu", torch.float32).contiguous().numpy()
        dist, row, cnt = index.hnsw_search_knn(q, kk, ef)
        d = np.full((q.shape[0], kk), np.inf, np.float32)
        r = np.full((q.shape[0], kk), 0xFFFFFFFF, np.int64)
        for i in range(q.shape[0]):
            c = int(cnt[i])
            order = np.lexsort((row[i, :c], dist[i, :c]))          # the engine hands back a heap; the wire format is best first
            d[i, :c], r[i, :c] = dist[i, :c][order], row[i, :c][order]
        dt, rt = torch.from_numpy(d), torch.from_numpy(r)
        return (dt.to(device), rt.to(device)) if device is not None else (dt, rt)
    return run


class ShardedBruteforceGpu:
    """The all-device form used on GPUs: local scan -> ONE all-gather of the raw [2][nq][kk] output words -> merge kernel.
    No torch arithmetic in the step: torch only owns the buffers and issues the RCCL collective."""

    def __init__(self, index, shard_rows: int, device, max_queries: int, kk: int, group=None):
        import torch.distributed as dist
        self._dist, self.index, self.shard_rows, self.device, self.kk, self.group = dist, index, int(shard_rows), device, kk, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.local = torch.empty((2, max_queries, kk), dtype=torch.int32, device=device)          # dist bits | rows
        self.gathered = torch.empty((self.world, 2, max_queries, kk), dtype=torch.int32, device=device)
        self.max_queries = max_queries

    def search_into(self, d_queries_ptr: int, nq: int, out_dist: torch.Tensor, out_row: torch.Tensor) -> None:
        """queries: device pointer to [nq][dim] f32; out_dist [nq][kk] f32 and out_row [nq][kk] i32 (global rows) device tensors."""
        from . import capi
        assert nq == self.max_queries, "buffers are sized for a fixed batch"
        stream = torch.cuda.current_stream(self.device).cuda_stream
        base = self.local.data_ptr()
        self.index.search_knn_device(d_queries_ptr, nq, self.kk, base, base + nq * self.kk * 4, None, stream)
        if self._dist.is_initialized():   # also with world == 1 (a self-copy), so that single-rank runs exercise the same path
            self._dist.all_gather_into_tensor(self.gathered.view(-1), self.local.view(-1), group=self.group)
            src = self.gathered.data_ptr()
        else:
            src = base
        capi.merge_shards_device(src, self.world, nq, self.kk, self.shard_rows, out_dist.data_ptr(), out_row.data_ptr(), None, stream)


class ShardedFtExchange:
    """ft_fast merge over DOCUMENT-RANGE shards, one rank per shard (SURVEY 8e "BM25"): what crosses the ranks, and what every rank derives
    from it.  The same two exchanges rxgpu_ft_create_sharded runs between the kernels of its launch train inside one process
    (rxgpu_ft_capi.hip, run_merge_sharded), here over torch.distributed — the one-process-per-GPU deployment; the local merger is injected
    (on a GPU box the rank's rxgpu shard, in tests/test_sharded_gloo.py the CPU oracle), so the exchange logic runs unchanged under gloo.

      1. every rank's pre-score histogram (65536 counters: documents of its range inside the restricting mask, not removed, by uint16
         pre-score) and the popcount of its mask words  ->  ONE all_gather.  From the sums every rank takes the same decisions the single
         index takes (mergerimpl.h:486-490, 433-446): does the preselect run, the threshold score, how many documents AT the threshold are
         kept — and, because ties are kept in document order = shard order, its own share of that quota: what is left after the shards in front.
      2. every rank's count of documents first met per sub-term row (addDoc order, merger.h:161-180)  ->  ONE all_gather.  The merge slot of
         a document is (documents first met in an earlier row, anywhere) + (same row, shards in front) + (same row, same shard, smaller id):
         every rank writes its documents at their GLOBAL slots, the union of the ranks' lists IS the single index's list."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.collectives = 0

    def _all_gather(self, t: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return t.unsqueeze(0)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(out, t, group=self.group)
        self.collectives += 1
        return torch.stack(out)

    def preselect(self, local_hist: torch.Tensor, local_popcount: int, merge_limit: int, max_merged: int, host_gate: bool):
        """local_hist: int64 [65536].  -> (on, min_score, quota): on = the preselect runs (the host half of the 2-phase gate held AND the summed
        popcount exceeds mergeLimit); a rank then keeps its documents with a pre-score above min_score and the first `quota` of its
        documents AT min_score, in document order."""
        if not host_gate:
            return False, 0, 0
        payload = torch.cat([local_hist.to(torch.int64), torch.tensor([int(local_popcount)], dtype=torch.int64)])
        g = self._all_gather(payload)
        if int(g[:, 65536].sum()) <= merge_limit:
            return False, 0, 0
        hist = g[:, :65536]
        total = hist.sum(dim=0)
        # mergerimpl.h:433-446: walk the scores downwards until maxMergedDocs documents are covered
        above = torch.flip(torch.cumsum(torch.flip(total, [0]), 0), [0]) - total   # documents strictly above every score
        visited = (above < max_merged)
        visited[0] = False   # the walk stops at score 1
        idx = torch.nonzero(visited)
        if idx.numel() == 0:
            return True, 65535, 0
        min_score = int(idx.min())
        min_docs = max_merged - int(above[min_score])
        used_in_front = int(hist[:self.rank, min_score].sum())
        return True, min_score, max(0, min_docs - used_in_front)

    def phrase_cut(self, local_row_admitted: torch.Tensor, merge_limit: int):
        """PhraseMerger's admission cut (phrasemerger.h:341: at most mergeLimit candidates of the phrase's first term, in (sub-term row, document)
        order) over the ranks.  local_row_admitted: int64 [n_rows] = the candidates this rank admitted per row under its local bound  ->  ONE
        all_gather; every rank derives the same cut (rows in order, inside a row the ranks in order: a rank's documents lie before the next
        rank's) and returns (keep: how many of ITS admitted candidates stay — a prefix of its own slots, NumDocsMerged() of the whole index).
        The same rule the library runs between its shards' admission passes (csrc/ft_phrase_cut.h)."""
        g = self._all_gather(local_row_admitted.to(torch.int64))   # [world, n_rows]
        flat = g.t().reshape(-1)                                   # (row, rank) order
        before = torch.cumsum(flat, 0) - flat
        take = torch.clamp(torch.minimum(flat, merge_limit - before), min=0)
        take = take.reshape(g.shape[1], g.shape[0])
        return int(take[:, self.rank].sum()), int(take.sum())

    def slot_bases(self, local_first_met: torch.Tensor):
        """local_first_met: int64 [n_rows] = this shard's documents first met per sub-term row.  -> (bases int64 [n_rows]: the merge slot of this
        shard's first document of every row, total documents the merge adds before the mergeLimit cut)"""
        g = self._all_gather(local_first_met.to(torch.int64))   # [world, n_rows]
        row_tot = g.sum(dim=0)
        rows_before = torch.cumsum(row_tot, 0) - row_tot
        shards_before = g[:self.rank].sum(dim=0) if self.rank else torch.zeros_like(row_tot)
        return rows_before + shards_before, int(row_tot.sum())
