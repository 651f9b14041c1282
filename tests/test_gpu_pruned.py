"""-m gpu: the opt-in bf16-pruned scan (RXGPU_SCAN_BF16=1; knn_scan_bf16 + knn_filter_approx in knn_scan.hip): approximate distances from the
bf16 shadow prune the corpus under a rigorous bound, the exact kernels re-score the survivors.  The contract does not move: identical rows
and distance bits to the exact f32 scan / the reference."""
import numpy as np
import pytest

from .conftest import lex_topk, make_corpus
from .test_gpu_batched import bits, check_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["blocked_shadow", "rowmajor_shadow"])
def pruned(request, monkeypatch):
    """The scan reads the same bf16 shadow as the nomination GEMM: tile-blocked by default, row-major with RXGPU_SHADOW_BLOCKED=0."""
    monkeypatch.setenv("RXGPU_SCAN_BF16", "1")
    if request.param == "rowmajor_shadow":
        monkeypatch.setenv("RXGPU_SHADOW_BLOCKED", "0")


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("d,n", [(768, 30_000), (128, 50_000), (512, 4_000), (100, 20_000), (130, 3_000)])
def test_pruned_scan_is_exact(rxgpu, oracle, pruned, metric, d, n):
    rows = make_corpus(d + 11, n, d)
    inv = oracle.l2_modules(rows) if metric == 2 else None
    allq = make_corpus(3000 + d, 8, d)
    if metric == 2:
        allq = np.stack([oracle.normalize_copy(q)[0] for q in allq])
    with rxgpu.VectorIndex(metric, d, n) as ix:
        ix.upload_rows(0, rows, inv)
        for nq, kk in ((1, 11), (1, 1), (3, 11), (8, 64), (1, 33)):
            check_batch(ix, oracle, metric, rows, inv, allq[:nq], kk)


def test_pruned_scan_ties_and_small_indexes(rxgpu, oracle, pruned):
    rng = np.random.default_rng(5)
    n, d = 60_000, 128
    rows = rng.integers(-1, 2, (n, d)).astype(np.float32)          # quantised: thousands of exact ties at the k-th distance
    queries = rng.integers(-1, 2, (4, d)).astype(np.float32)
    for metric in (0, 1):
        with rxgpu.VectorIndex(metric, d, n) as ix:
            ix.upload_rows(0, rows)
            check_batch(ix, oracle, metric, rows, None, queries[:1], 11)
            check_batch(ix, oracle, metric, rows, None, queries, 11)
    small = make_corpus(1, 7, 128)
    with rxgpu.VectorIndex("l2", 128, 16) as ix:
        ix.upload_rows(0, small)
        check_batch(ix, oracle, 0, small, None, make_corpus(2, 2, 128), 11)   # kk > n


def test_pruned_scan_adversarial_magnitudes_and_mutations(rxgpu, oracle, pruned):
    rng = np.random.default_rng(78)
    n, d = 20_000, 128
    rows = (rng.normal(0, 1, (n, d)) * np.exp(rng.uniform(-14, 14, (n, 1)))).astype(np.float32)
    rows[1000:1200] = rows[0] * (1 + rng.uniform(-1e-4, 1e-4, (200, 1))).astype(np.float32)
    rows[2000:2050] = rows[0]
    queries = (rng.normal(0, 1, (6, d)) * np.exp(rng.uniform(-6, 6, (6, 1)))).astype(np.float32)
    queries[0] = rows[0]
    for metric in (0, 1):
        with rxgpu.VectorIndex(metric, d, n + 10) as ix:
            ix.upload_rows(0, rows)
            for qi in range(6):
                check_batch(ix, oracle, metric, rows, None, queries[qi:qi + 1], 11)
            cur = rows.copy()
            cur[5] = queries[3]
            ix.upload_rows(5, cur[5:6])
            check_batch(ix, oracle, metric, cur, None, queries[3:4], 5)
            last = cur.shape[0] - 1
            ix.move_row(last, 5)
            ix.truncate(last)
            cur[5] = cur[last]
            check_batch(ix, oracle, metric, cur[:last], None, queries[3:4], 5)
