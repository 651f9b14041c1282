"""pytest configuration: `-m gpu` tests need a real MI355X; everything else runs on CPU."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("RX_TARGET_INSTRUCTIONS", "avx512")  # pin the reference's SIMD level (cpucheck.cc:201-231)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_runtest_logreport(report):
    """RXGPU_TEST_TIMES=<file>: one line per finished test (outcome, seconds, node id), flushed at once — what a run that is cut short
    (a `timeout` around pytest on the GPU box) still leaves behind."""
    path = os.environ.get("RXGPU_TEST_TIMES")
    if path and report.when == "call":
        with open(path, "a") as f:
            f.write(f"{report.outcome} {report.duration:8.2f} {report.nodeid}\n")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The real reference engines (oracle/_ref). Built on demand where /root/reference exists; else skipped."""
    from oracle import pyoracle
    if not pyoracle.REF_SO.exists() and pyoracle.REFERENCE_TREE.exists():
        pyoracle.build_ref()
    r = pyoracle.ref_or_none()
    if r is None:
        pytest.skip("oracle/_ref/libref_oracle.so not available (no /root/reference here)")
    if r.simd_level != 3:
        pytest.skip("host CPU lacks AVX-512: the reference dispatches to a different summation order")
    return r


@pytest.fixture(scope="session")
def rxgpu():
    """The HIP library through its C-ABI. Fails loudly when it is not built; skips only when no GPU is visible."""
    from reindexer_amd import capi
    capi.lib()  # raises if librxgpu.so is missing — no fallback
    if not capi.gpu_available():
        pytest.skip("no HIP device visible")
    return capi


def make_corpus(seed, n, d, scale=0.25):
    """Reference test distribution: N(0, 0.25) per component (gtests/tools.h:121-129)."""
    rng = np.random.default_rng(seed)
    return rng.normal(0.0, scale, (n, d)).astype(np.float32)


def lex_topk(dist, kk):
    """Exact top-kk under the (dist,row) total order."""
    order = np.lexsort((np.arange(dist.shape[0]), dist))[:kk]
    return dist[order], order.astype(np.uint32)


def apply_swap_deletes(rows, labels, victims):
    """Replays BruteforceSearch::RemovePoint (bruteforce.cc:70-86): the last row is moved into the hole."""
    rows, labels = rows.copy(), labels.copy()
    cnt = rows.shape[0]
    for v in victims:
        lab = labels[v] if np.isscalar(v) or isinstance(v, (int, np.integer)) else v
        pos = int(np.nonzero(labels[:cnt] == lab)[0][0])
        if pos + 1 != cnt:
            rows[pos] = rows[cnt - 1]
            labels[pos] = labels[cnt - 1]
        cnt -= 1
    return rows[:cnt].copy(), labels[:cnt].copy()
