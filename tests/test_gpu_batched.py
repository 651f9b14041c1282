"""-m gpu: batched queries (nq >= 2) take the MFMA candidate path (knn_batched.hip): fp32 matrix-core GEMM nominates
rows, the exact kernels decide.  The contract is unchanged: identical rows and distance bits to per-query exact search
(the reference has no batched API — a batch is B sequential SearchKnn calls)."""
import numpy as np
import pytest

from .conftest import lex_topk, make_corpus

pytestmark = pytest.mark.gpu


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def check_batch(ix, oracle, metric, rows, inv, queries, kk):
    dist, row, cnt = ix.search_knn(queries, kk)
    for qi in range(queries.shape[0]):
        want_all = oracle.dist_many(metric, queries[qi], rows, inv)
        c = min(kk, rows.shape[0])
        wd, wr = lex_topk(want_all, c)
        assert int(cnt[qi]) == c
        assert np.array_equal(row[qi, :c], wr), (metric, qi, kk)
        assert np.array_equal(bits(dist[qi, :c]), bits(wd))


@pytest.fixture(params=["bf16", "bf16_single_ring", "f32"])
def nomination(request, monkeypatch):
    """The nomination GEMMs: bf16 matrix cores over the shadow (default for every batch; the split-ring kernel, and the single-ring one it
    replaced: RXGPU_GEMM_SPLIT=0) and f32-input MFMA over the rows (what a GPU without room for the shadow falls back to)."""
    if request.param == "f32":
        monkeypatch.setenv("RXGPU_BATCH_BF16_MIN", "0")
    if request.param == "bf16_single_ring":
        monkeypatch.setenv("RXGPU_GEMM_SPLIT", "0")
    return request.param


@pytest.fixture(params=["split_ring", "single_ring", "rowmajor_shadow"])
def gemm_ring(request, monkeypatch):
    """Both nomination kernels over the tile-blocked bf16 shadow (default), and the split-ring one over the row-major shadow."""
    if request.param == "single_ring":
        monkeypatch.setenv("RXGPU_GEMM_SPLIT", "0")
    if request.param == "rowmajor_shadow":
        monkeypatch.setenv("RXGPU_SHADOW_BLOCKED", "0")
    return request.param


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("d", [24, 100, 128, 768])
def test_batched_equals_sequential(rxgpu, oracle, nomination, metric, d):
    n = 40_000 if d <= 128 else 12_000
    rows = make_corpus(d, n, d)
    inv = oracle.l2_modules(rows) if metric == 2 else None
    allq = make_corpus(1000 + d, 70, d)
    if metric == 2:
        allq = np.stack([oracle.normalize_copy(q)[0] for q in allq])
    with rxgpu.VectorIndex(metric, d, n) as ix:
        ix.upload_rows(0, rows, inv)
        for nq, kk in ((2, 11), (5, 1), (32, 11), (33, 64), (70, 11)):
            check_batch(ix, oracle, metric, rows, inv, allq[:nq], kk)


@pytest.mark.parametrize("metric", [0, 2])
@pytest.mark.parametrize("d", [64, 128])
def test_batched_many_tiles_per_block_at_small_ld(rxgpu, oracle, metric, d):
    """ld = 64 / 128 (2 / 4 stages per tile) with > 2 tiles per workgroup: the split-ring kernel's row loaders run 2-3 TILES ahead, so the
    per-row epilogue terms (|x|^2, 1/|x|) of several tiles are in LDS at once (a ring of two was overrun here).  Rows with
    very different norms make a swapped term visible: a tile scored with its neighbour's norms drops true neighbours."""
    n, nq, kk = 330_000, 256, 11
    rng = np.random.default_rng(64 + d + metric)
    rows = rng.normal(0, 0.25, (n, d)).astype(np.float32)
    rows *= rng.uniform(0.2, 3.0, (n, 1)).astype(np.float32)   # norms vary tile to tile and row to row
    inv = oracle.l2_modules(rows) if metric == 2 else None
    queries = make_corpus(9000 + d, nq, d)
    if metric == 2:
        queries = np.stack([oracle.normalize_copy(q)[0] for q in queries])
    with rxgpu.VectorIndex(metric, d, n) as ix:
        ix.upload_rows(0, rows, inv)
        dist, row, cnt = ix.search_knn(queries, kk)
        for qi in range(0, nq, 3):
            wd, wr = lex_topk(oracle.dist_many(metric, queries[qi], rows, inv), kk)
            assert int(cnt[qi]) == kk
            assert np.array_equal(row[qi, :kk], wr), (metric, d, qi)
            assert np.array_equal(bits(dist[qi, :kk]), bits(wd))


def test_batched_more_than_256_queries(rxgpu, oracle):
    n, d = 30_000, 64
    rows = make_corpus(1, n, d)
    queries = make_corpus(2, 300, d)
    with rxgpu.VectorIndex("ip", d, n) as ix:
        ix.upload_rows(0, rows)
        check_batch(ix, oracle, 1, rows, None, queries, 11)


def test_batched_ties_take_the_gated_fallback(rxgpu, oracle):
    """Quantised data: thousands of rows tie with the k-th distance, the nomination lists overflow and the device-side
    gate reruns those queries through the exact fused scan. Results must still be exact."""
    rng = np.random.default_rng(5)
    n, d = 60_000, 8
    rows = rng.integers(-1, 2, (n, d)).astype(np.float32)
    queries = rng.integers(-1, 2, (40, d)).astype(np.float32)
    for metric in (0, 1):
        with rxgpu.VectorIndex(metric, d, n) as ix:
            ix.upload_rows(0, rows)
            check_batch(ix, oracle, metric, rows, None, queries, 11)


def test_batched_small_index_and_mutation(rxgpu, oracle):
    """n < kk, n < sample size, and the cached row statistics are invalidated by uploads / moves / truncation."""
    d = 128
    rows = make_corpus(3, 5000, d) * np.float32(0.01)
    queries = make_corpus(4, 8, d)
    with rxgpu.VectorIndex("l2", d, 6000) as ix:
        ix.upload_rows(0, rows[:5])
        check_batch(ix, oracle, 0, rows[:5], None, queries, 11)
        ix.upload_rows(5, rows[5:])
        check_batch(ix, oracle, 0, rows, None, queries, 11)
        big = (make_corpus(5, 1000, d) * np.float32(50.0)).astype(np.float32)   # much larger norms: the bound must be refreshed
        ix.upload_rows(5000, big)
        allrows = np.concatenate([rows, big])
        check_batch(ix, oracle, 0, allrows, None, queries, 11)
        ix.move_row(5999, 0)
        ix.truncate(5999)
        allrows[0] = allrows[5999]
        check_batch(ix, oracle, 0, allrows[:5999], None, queries, 11)


def test_batched_device_api_matches_host_api(rxgpu, oracle):
    import torch
    n, d, nq, kk = 50_000, 256, 48, 11
    rows = make_corpus(7, n, d)
    queries = make_corpus(8, nq, d)
    t_rows = torch.from_numpy(rows).cuda()
    t_q = torch.from_numpy(queries).cuda()
    od = torch.empty((nq, kk), dtype=torch.float32, device="cuda")
    orow = torch.empty((nq, kk), dtype=torch.int32, device="cuda")
    ocnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
    with rxgpu.VectorIndex("cosine", d) as ix:
        inv = oracle.l2_modules(rows)
        t_inv = torch.from_numpy(inv).cuda()
        ix.adopt_device_rows(t_rows.data_ptr(), n, d, t_inv.data_ptr(), keepalive=(t_rows, t_inv))
        ix.search_knn_device(t_q.data_ptr(), nq, kk, od.data_ptr(), orow.data_ptr(), ocnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        hd, hr, hc = ix.search_knn(queries, kk)
        assert np.array_equal(orow.cpu().numpy().view(np.uint32), hr)
        assert np.array_equal(bits(od.cpu().numpy()), bits(hd))
        check_batch(ix, oracle, 2, rows, inv, queries[:6], kk)


# ---------------------------------------------------------------------------------------------- bf16 nomination path (batches > 64 queries)
@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("d,n", [(100, 30_000), (768, 20_000), (130, 9_000), (64, 300), (512, 6_000), (250, 5_000), (768, 33)])
def test_bf16_nomination_is_exact(rxgpu, oracle, gemm_ring, metric, d, n):
    """Batches of 65..256 queries are nominated by the bf16 matrix-core GEMM over the bf16 shadow of the rows (knn_batched_bf16.hip) under a
    rigorous rounding bound; the exact kernels re-score.  Rows and distance bits must equal per-query exact search."""
    rows = make_corpus(d + 7, n, d)
    inv = oracle.l2_modules(rows) if metric == 2 else None
    allq = make_corpus(2000 + d, 256, d)
    if metric == 2:
        allq = np.stack([oracle.normalize_copy(q)[0] for q in allq])
    with rxgpu.VectorIndex(metric, d, n) as ix:
        ix.upload_rows(0, rows, inv)
        for nq, kk in ((65, 11), (128, 3), (200, 11), (256, 33)):
            check_batch(ix, oracle, metric, rows, inv, allq[:nq], kk)


@pytest.mark.parametrize("metric", [0, 1])
def test_bf16_nomination_adversarial_magnitudes(rxgpu, oracle, metric):
    """Rows spanning 12 orders of magnitude, queries with huge and tiny components, exact ties between near-duplicate rows that differ below
    bf16 resolution: the bound must stay sound (no true neighbour lost), whatever it costs in nominated rows."""
    rng = np.random.default_rng(77)
    n, d = 20_000, 96
    rows = (rng.normal(0, 1, (n, d)) * np.exp(rng.uniform(-14, 14, (n, 1)))).astype(np.float32)
    rows[1000:1200] = rows[0] * (1 + rng.uniform(-1e-4, 1e-4, (200, 1))).astype(np.float32)      # near duplicates of row 0
    rows[2000:2050] = rows[0]                                                                    # exact duplicates
    queries = (rng.normal(0, 1, (100, d)) * np.exp(rng.uniform(-6, 6, (100, 1)))).astype(np.float32)
    queries[:10] = rows[0] * np.float32(1.0) + rng.normal(0, 1e-6, (10, d)).astype(np.float32)
    with rxgpu.VectorIndex(metric, d, n) as ix:
        ix.upload_rows(0, rows)
        check_batch(ix, oracle, metric, rows, None, queries, 11)


def test_bf16_shadow_follows_mutations(rxgpu, oracle, gemm_ring):
    rng = np.random.default_rng(9)
    n, d = 5000, 80
    rows = make_corpus(3, n, d)
    queries = make_corpus(4, 80, d)
    with rxgpu.VectorIndex("l2", d, n + 100) as ix:
        ix.upload_rows(0, rows)
        check_batch(ix, oracle, 0, rows, None, queries, 5)
        rows2 = rows.copy()
        rows2[17] = queries[3]                       # overwrite a row: the shadow must be rebuilt, the new row is now the nearest
        ix.upload_rows(17, rows2[17:18])
        check_batch(ix, oracle, 0, rows2, None, queries, 5)
        extra = make_corpus(5, 60, d)
        ix.upload_rows(n, extra)
        cur = np.concatenate([rows2, extra])
        check_batch(ix, oracle, 0, cur, None, queries, 5)
        # swap-with-last deletes (bruteforce.cc:70-86): the shadow row and |x|^2 travel with the moved row, the tail is cut
        for victim in (17, 0, 4000):
            last = cur.shape[0] - 1
            ix.move_row(last, victim)
            ix.truncate(last)
            cur[victim] = cur[last]
            cur = cur[:last]
            check_batch(ix, oracle, 0, cur, None, queries, 5)


def test_batched_thresholds_on_insertion_ordered_corpus(rxgpu, oracle):
    """Rows inserted cluster by cluster (the far clusters first): a prefix sample would give useless thresholds; the strided sample keeps the
    nomination lists short.  Either way the result must be exact."""
    rng = np.random.default_rng(31)
    d, per, nc = 48, 1500, 24
    centres = rng.normal(0, 4.0, (nc, d)).astype(np.float32)
    rows = np.concatenate([centres[c] + rng.normal(0, 0.3, (per, d)).astype(np.float32) for c in range(nc)])
    queries = (centres[nc - 1] + rng.normal(0, 0.3, (100, d))).astype(np.float32)       # all queries live in the LAST cluster
    for metric in (0, 1):
        with rxgpu.VectorIndex(metric, d, rows.shape[0]) as ix:
            ix.upload_rows(0, rows)
            check_batch(ix, oracle, metric, rows, None, queries, 11)
            check_batch(ix, oracle, metric, rows, None, queries[:40], 11)
